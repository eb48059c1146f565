"""On-device PPO update for the MAPPO networks over a rollout record (BASELINE configs[2]: "on-device PPO+GAE").

The maths is `MappoAgent.ppo_update` (src/agent/mappo_agent.py:247-293): returns by the reverse discounted sum
(`collector.device_returns`, one HIP launch), advantages = returns - values standardised with std + 1e-8, the critic's
MSE and the clipped surrogate (epsilon = 0.2, no entropy / value coefficients) — batched over [T, B, A] with every
agent's own action / log-probability / advantage, in minibatches of env-steps.

What makes it fast (round 2's plain-torch form spent 35.7 ms per update of 1.3 M transitions, 98 % of an iteration):
  * no host synchronisation anywhere in the loop (losses stay device tensors; the caller reads them when it wants);
  * first layers as matmuls on two [mb, N] one-hot / multi-hot matrices (all police actors in one matmul; a row-lookup
    form has a scatter-add backward with ~10^5 colliding rows per minibatch: 14 of round 2's 30 ms), second layers as
    ONE batched matmul -> logits [A, mb, N];
  * the new log-probabilities come from a log-sum-exp over the <= 16 AFFORDABLE entries of the agent's ELL row (gathered
    from the logits with the board table) — the masked, renormalised softmax of `select_action` (mappo_agent.py:112-134)
    without ever forming [mb, A, N] probability / mask tensors;
  * one fused Adam step per minibatch; optionally the whole minibatch step replayed as ONE HIP graph (`use_graph`).

`fused=True` (the default on a GPU) goes further: loss AND gradient of
a minibatch come from ONE HIP kernel, `sy_mappo_ppo_grad` (include/sy_env.h, csrc/sy_ppo.hip), which reads the packed
rollout record in place through the minibatch's row indices (no shuffled copy of the record), evaluates only the
affordable logits of every (row, agent) and accumulates each network's gradient in LDS (float64); the launch that sums
the blocks' tables also takes the Adam step, on parameters that stay RESIDENT in the layout the kernels read (one slab per
network; the module is refreshed from them at the end of the update and read into them at its start).  A minibatch is two
launches instead of ~120.  The torch form above stays as `fused=False` — the restatement the kernel is tested against
(tests/test_gpu_surface.py: every gradient, and the parameters after Adam steps).
"""
import ctypes as C
from typing import Dict, Optional

import torch

from .policies import MappoPolicy


class _OneHotLinear(torch.autograd.Function):
    """y = oh @ Wt for a {0, 1} matrix oh [R, N] without gradient.  The weight gradient oh^T g is a product with a tiny
    output (N x H) and a huge reduction dimension (R rows): the BLAS picks one workgroup column for it (0.81 ms at
    R = 262 144 on MI355X); cutting R into chunks — a batched product followed by a sum, i.e. split-K — runs 3x faster."""

    @staticmethod
    def forward(ctx, oh, Wt):
        ctx.save_for_backward(oh)
        return oh @ Wt

    @staticmethod
    def backward(ctx, g):
        (oh,) = ctx.saved_tensors
        R, N = oh.shape
        C = 1
        while C < 64 and R % (2 * C) == 0 and R // (2 * C) >= 4096:
            C *= 2
        if C == 1:
            return None, oh.t() @ g
        gW = torch.bmm(oh.view(C, R // C, N).transpose(1, 2), g.reshape(C, R // C, g.shape[1])).sum(0)
        return None, gW


class MappoUpdater:
    def __init__(self, net: MappoPolicy, ell: torch.Tensor, env_graph: torch.Tensor, lr: float = 3e-4, clip: float = 0.2,
                 minibatch: int = 32768, value_coef: float = 0.5, use_graph: bool = False, mrx_money: int = 1000,
                 fused: Optional[bool] = None, grad_sync=None):
        """ell int32 [G, N, 16] (the engine's board table: neighbour | weight << 16), env_graph int [B].
        fused: None = the HIP gradient kernel when it applies (GPU, hidden a multiple of 4 up to 128);
        True = require it (raises otherwise); False = the torch form.
        grad_sync: data-parallel training on the fused path — a callable that receives the minibatch's gradient slab
        [A + 1, S] (summed over THIS rank's rows) and averages it across the ranks in place, e.g.
        `collector.allreduce_slab`; the Adam step then runs as its own launch on the averaged gradient."""
        self.net, self.clip, self.minibatch, self.value_coef = net, float(clip), int(minibatch), float(value_coef)
        self.device = next(net.parameters()).device
        self.ell = ell.to(self.device)
        self.env_graph = env_graph.to(self.device).long()
        self.N, self.P, self.A = net.N, net.P, net.A
        self.mrx_money = int(mrx_money)
        on_gpu = self.device.type == "cuda"
        self.opt = torch.optim.Adam(net.parameters(), lr=lr, fused=on_gpu, capturable=on_gpu and use_graph)
        self.use_graph = bool(use_graph) and on_gpu
        self._graph, self._static = None, None
        self.last_losses = None
        self._fz = None
        self.grad_sync = grad_sync
        self._env_graph32 = self.env_graph.to(torch.int32).contiguous()
        H = net.actors[0][0].out_features
        # (mirror of sy_ppo.hip's limits: a gradient table that does not fit the LDS is cut into row ranges, one role each,
        # at most 64 roles = 2 tables x (A + 1) networks x ranges)
        room = 160 * 1024 - (H + ((max(self.N, H) + 3) & ~3) + 8) * 8
        rpp = min(self.N, room // (H * 8)) if H > 0 else 0
        ranges = -(-self.N // rpp) if rpp >= 1 else 1 << 30
        fits = on_gpu and H % 4 == 0 and H <= 128 and 2 * (self.A + 1) * ranges <= 64
        if fused and not fits:
            raise ValueError("the fused PPO gradient kernel needs a GPU, hidden % 4 == 0, hidden <= 128 and nodes x hidden small "
                             "enough for 64 table roles (e.g. 1024 nodes at hidden 64, 512 at 128 with 5 agents)")
        self.fused = fits if fused is None else bool(fused)
        self.H = H

    @staticmethod
    def _slab_floats(N, H):
        """floats per role of the kernel's gradient layout (mirror of sy_ppo_slab_floats; checked against it in `_fused_state`)"""
        return 2 * N * H + H + ((max(N, H) + 3) & ~3) + 8

    # ------------------------------------------------------------------ the HIP gradient kernel
    def _fused_state(self):
        """Persistent buffers of the fused path (allocated once: a captured graph replays on them).  Parameters, gradients
        and Adam's moments share ONE layout — a slab per network, include/sy_env.h — so a minibatch is two launches: the
        gradient kernel reads `theta`, the reduction applies the Adam step to it in place."""
        if self._fz is not None:
            return self._fz
        from . import _lib
        lib = _lib.load()
        A, N, H, dev = self.A, self.N, self.H, self.device
        S = int(lib.sy_ppo_slab_floats(N, H))
        assert S == self._slab_floats(N, H)
        f = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)    # noqa: E731
        DN = (max(N, H) + 3) & ~3
        self._fz = {"lib": lib, "_lib": _lib, "S": S, "theta": f(A + 1, S), "grads": f(A + 1, S), "m": f(A + 1, S), "v": f(A + 1, S),
                    "step": torch.zeros(1, dtype=torch.int32, device=dev), "scratch": f(int(lib.sy_ppo_scratch_floats(A, N, H))),
                    "actor_loss": f(), "critic_loss": f(), "oE": 2 * N * H + H + DN, "image": None}
        # The actors' parameters become views of four stacked tensors (same values, same Parameter objects): the module <->
        # slab copies are then one strided copy per kind instead of one per actor and kind.
        with torch.no_grad():
            acts = self.net.actors
            mir = {"w1": torch.stack([a[0].weight for a in acts]), "b1": torch.stack([a[0].bias for a in acts]),
                   "w2": torch.stack([a[2].weight for a in acts]), "b2": torch.stack([a[2].bias for a in acts])}
            for a, actor in enumerate(acts):
                actor[0].weight.data, actor[0].bias.data = mir["w1"][a], mir["b1"][a]
                actor[2].weight.data, actor[2].bias.data = mir["w2"][a], mir["b2"][a]
        self._fz["mirror"] = mir
        return self._fz

    def _mirrors(self):
        """The stacked actor parameters, re-stacked if somebody re-pointed a parameter's storage since (load_state_dict
        copies in place and keeps the views; `.data = ...` assignments do not)."""
        mir, acts = self._fz["mirror"], self.net.actors
        same = all(acts[a][0].weight.data_ptr() == mir["w1"][a].data_ptr() and acts[a][2].weight.data_ptr() == mir["w2"][a].data_ptr() and
                   acts[a][0].bias.data_ptr() == mir["b1"][a].data_ptr() and acts[a][2].bias.data_ptr() == mir["b2"][a].data_ptr()
                   for a in range(self.A))
        if not same:
            with torch.no_grad():
                mir = {"w1": torch.stack([a[0].weight for a in acts]), "b1": torch.stack([a[0].bias for a in acts]),
                       "w2": torch.stack([a[2].weight for a in acts]), "b2": torch.stack([a[2].bias for a in acts])}
                for a, actor in enumerate(acts):
                    actor[0].weight.data, actor[0].bias.data = mir["w1"][a], mir["b1"][a]
                    actor[2].weight.data, actor[2].bias.data = mir["w2"][a], mir["b2"][a]
            self._fz["mirror"] = mir
        return mir

    def _slab_views(self, t):
        """Named views of a [A + 1, S] slab tensor in torch's parameter shapes (transposed where the kernel's layout is)."""
        A, N, H = self.A, self.N, self.H
        NH, oE = N * H, self._fz["oE"]
        act, cri = t[:A], t[A]
        return {"w1t": act[:, :NH].view(A, N, H), "w2": act[:, NH:2 * NH].view(A, N, H), "b1": act[:, 2 * NH:2 * NH + H],
                "b2": act[:, 2 * NH + H:2 * NH + H + N], "c1m": cri[:NH].view(N, H), "c1p": cri[NH:2 * NH].view(N, H),
                "cb1": cri[2 * NH:2 * NH + H], "c2": cri[2 * NH + H:2 * NH + 2 * H], "cb2": cri[oE + 1:oE + 2]}

    @torch.no_grad()
    def load_from_module(self):
        """module parameters -> the resident slabs (start of every update: the module stays the source of truth between
        updates, so checkpoints / manual edits of it are honoured)."""
        z, net, H, A, N = self._fused_state(), self.net, self.H, self.A, self.N
        v = self._slab_views(z["theta"])
        mir = self._mirrors()
        v["w1t"].copy_(mir["w1"].transpose(1, 2))
        v["w2"].copy_(mir["w2"])
        v["b1"].copy_(mir["b1"])
        v["b2"].copy_(mir["b2"])
        c1 = net.critic[0].weight.view(H, A, N)
        v["c1m"].copy_(c1[:, 0].t())
        z["c1p_before"] = c1[:, 1:].sum(1).t().contiguous()
        v["c1p"].copy_(z["c1p_before"])
        v["cb1"].copy_(net.critic[0].bias)
        v["c2"].copy_(net.critic[2].weight.view(-1))
        v["cb2"].copy_(net.critic[2].bias)

    @torch.no_grad()
    def store_to_module(self):
        """the resident slabs -> module parameters (end of every update).  The critic's P police blocks all took the same
        steps (the police table is their sum, moved P steps): each block moves by the table's change / P."""
        z, net, H, A, N, P = self._fz, self.net, self.H, self.A, self.N, self.P
        v = self._slab_views(z["theta"])
        mir = z["mirror"]                                   # (the parameters are views of these: load_from_module checked)
        mir["w1"].copy_(v["w1t"].transpose(1, 2))
        mir["w2"].copy_(v["w2"])
        mir["b1"].copy_(v["b1"])
        mir["b2"].copy_(v["b2"])
        c1 = net.critic[0].weight.view(H, A, N)
        c1[:, 0].copy_(v["c1m"].t())
        c1[:, 1:].add_(((v["c1p"] - z["c1p_before"]) / P).t().unsqueeze(1))
        net.critic[0].bias.copy_(v["cb1"])
        net.critic[2].weight.view(-1).copy_(v["c2"])
        net.critic[2].bias.copy_(v["cb2"])

    def fused_gradients(self):
        """The last minibatch's gradient in torch's parameter shapes, keyed like `net.named_parameters()` (tests)."""
        g = self._slab_views(self._fz["grads"])
        out = {}
        for a in range(self.A):
            out["actors.%d.0.weight" % a] = g["w1t"][a].t()
            out["actors.%d.0.bias" % a] = g["b1"][a]
            out["actors.%d.2.weight" % a] = g["w2"][a]
            out["actors.%d.2.bias" % a] = g["b2"][a]
        c1g = torch.stack([g["c1m"].t()] + [g["c1p"].t()] * self.P, dim=1)          # [H, A, N]
        out["critic.0.weight"] = c1g.reshape(self.H, self.A * self.N)
        out["critic.0.bias"] = g["cb1"]
        out["critic.2.weight"] = g["c2"].view(1, self.H)
        out["critic.2.bias"] = g["cb2"]
        return out

    def _step_fused(self, z, num_rows, row0=0):
        """One minibatch (image rows row0 .. row0 + num_rows - 1; under a captured graph the device word z['row0'] decides):
        sy_mappo_ppo_grad = the gradient launch + the reduction that takes the Adam step on the resident parameters."""
        ptr = lambda t: C.c_void_p(t.data_ptr())                                   # noqa: E731
        b1, b2 = self.opt.param_groups[0]["betas"]
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        if self.grad_sync is not None:                  # gradient launch, the ranks' slabs averaged, then the Adam launch
            args = z["_lib"].PpoArgs(ptr(z["image"]), int(z["image_rows"]), int(row0), ptr(z["row0"]) if self.use_graph else None,
                                     int(num_rows), ptr(self.ell), self.P, self.N, self.H, ptr(z["theta"]), self.clip, self.value_coef,
                                     ptr(z["scratch"]), int(z["scratch"].numel()), ptr(z["grads"]), None, None, None,
                                     float(self.opt.param_groups[0]["lr"]), float(b1), float(b2), float(self.opt.param_groups[0]["eps"]))
            z["_lib"].check(z["lib"].sy_mappo_ppo_grad(C.byref(args), stream), "sy_mappo_ppo_grad")
            self.grad_sync(z["grads"])
            z["_lib"].check(z["lib"].sy_ppo_adam_step(ptr(z["theta"]), ptr(z["grads"]), ptr(z["m"]), ptr(z["v"]), ptr(z["step"]), self.P,
                                                      self.N, self.H, float(self.opt.param_groups[0]["lr"]), float(b1), float(b2),
                                                      float(self.opt.param_groups[0]["eps"]), stream), "sy_ppo_adam_step")
            return
        args = z["_lib"].PpoArgs(ptr(z["image"]), int(z["image_rows"]), int(row0), ptr(z["row0"]) if self.use_graph else None,
                                 int(num_rows), ptr(self.ell), self.P, self.N, self.H, ptr(z["theta"]), self.clip, self.value_coef,
                                 ptr(z["scratch"]), int(z["scratch"].numel()), ptr(z["grads"]), ptr(z["m"]), ptr(z["v"]), ptr(z["step"]),
                                 float(self.opt.param_groups[0]["lr"]), float(b1), float(b2), float(self.opt.param_groups[0]["eps"]))
        z["_lib"].check(z["lib"].sy_mappo_ppo_grad(C.byref(args), stream), "sy_mappo_ppo_grad")

    @staticmethod
    def _chunks(t, inner):
        """(rows per chunk, chunk stride in elements) of a [..., B, inner] view whose trailing [T, B, inner] block is
        contiguous: one chunk if the whole view is, else one chunk per index of the leading dim (the per-rank arenas of
        `TrajectoryExchange.gather`: [world, T, B, inner], contiguous inside a rank, `nbytes` apart between ranks)."""
        if t.is_contiguous():
            return 0, 0
        if t.dim() >= 3 and t[0].is_contiguous() and t.stride(0) >= t[0].numel():
            return t[0].numel() // inner, t.stride(0)
        raise ValueError("the fused update reads the record in place: it must be contiguous, or a stack of contiguous per-rank records")

    def _update_fused(self, rec, returns, values, generator):
        A = rec["action"].shape[-1]
        B = rec["action"].shape[-2]
        R = rec["action"].numel() // A                                           # rows = (ranks x) T x B
        mb = min(self.minibatch, R)
        nfull = R // mb                                                           # (a ragged tail is dropped, as minibatch PPO does)
        record, log_prob = rec["record"], rec["log_prob"]
        if record.dtype != torch.int32 or log_prob.dtype != torch.float32 or record.shape[:-1] != rec["action"].shape[:-1]:
            raise ValueError("the fused update reads the packed rollout record: rec['record'] int32 [..., T, B, RW], rec['log_prob'] float32")
        RW = int(record.shape[-1])
        rec_chunk, rec_stride = self._chunks(record, RW)
        lp_chunk, lp_stride = self._chunks(log_prob, A)
        if (rec_chunk == 0) != (lp_chunk == 0) or (rec_chunk and rec_chunk != lp_chunk):
            log_prob, lp_chunk, lp_stride = log_prob.contiguous(), 0, 0
            if rec_chunk:
                record, rec_chunk, rec_stride = record.contiguous(), 0, 0
        adv = returns if values is None else returns - values.unsqueeze(-1)
        adv = ((adv - adv.mean()) / (adv.std() + 1e-8)).float().contiguous()      # mappo_agent.py:256-258
        team_ret = returns.sum(-1).float().contiguous()
        if self.ell.dtype != torch.int32 or not self.ell.is_contiguous():
            self.ell = self.ell.to(torch.int32).contiguous()
        z = self._fused_state()
        lib, _lib = z["lib"], z["_lib"]
        rows = nfull * mb
        need = int(lib.sy_ppo_image_bytes(A, rows))
        if z["image"] is None or z["image"].numel() < need or z["image_rows"] != rows:
            z["image"] = torch.empty(need, dtype=torch.uint8, device=self.device)
            z["image_rows"] = rows
            z["row0"] = torch.zeros(1, dtype=torch.int32, device=self.device)
            z["starts"] = torch.arange(0, rows, mb, dtype=torch.int32, device=self.device)
            self._graph = None
        self.load_from_module()
        # ONE shuffle of the record per update, as a compact image the gradient launches stream (minibatch i = its rows
        # i * mb ...): pos / budget / action of the packed record, the log-probabilities, advantages and critic targets
        # (the shuffle is a keyed permutation computed inside the pack kernel — no sort; keyed by the generator's seed, or
        # torch's, and the number of updates so far)
        self._updates = getattr(self, "_updates", 0) + 1
        base = generator.initial_seed() if generator is not None else torch.initial_seed()
        seed = (int(base) * 0x9E3779B97F4A7C15 + self._updates * 0xD1B54A32D192ED03 + 0x85EBCA6B) & (2 ** 64 - 1)
        if generator is not None:
            seed = (int(base) * 0x9E3779B97F4A7C15 + 0x85EBCA6B) & (2 ** 64 - 1)   # an explicit generator decides alone (reproducible)
        ptr = lambda t: C.c_void_p(t.data_ptr())                                   # noqa: E731
        pargs = _lib.PpoPackArgs(ptr(record), RW, ptr(log_prob), ptr(adv), ptr(team_ret), None, 0, rows, B,
                                 ptr(self._env_graph32), self.P, ptr(z["image"]), int(z["image"].numel()), R, seed,
                                 int(rec_chunk), int(rec_stride), int(lp_stride))
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(lib.sy_ppo_pack(C.byref(pargs), stream), "sy_ppo_pack")
        if self.use_graph and self._graph is not None and self._graph_key != (z["image"].data_ptr(), mb):
            self._graph = None                                                    # other buffers: capture again
        for i in range(nfull):
            if not self.use_graph:
                self._step_fused(z, mb, i * mb)
                continue
            z["row0"].copy_(z["starts"][i:i + 1])
            if self._graph is None:
                self._step_fused(z, mb)                                           # first minibatch: eager (warm-up), then capture
                torch.cuda.synchronize(self.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.device(self.device), torch.cuda.graph(g):
                    self._step_fused(z, mb)
                self._graph, self._graph_key = g, (z["image"].data_ptr(), mb)
            else:
                self._graph.replay()
        oE = z["oE"]
        z["actor_loss"].copy_(z["grads"][:A, oE].sum())
        z["critic_loss"].copy_(z["grads"][A, oE])
        self.store_to_module()
        self.last_losses = (z["actor_loss"], z["critic_loss"])
        return self.last_losses

    # ------------------------------------------------------------------ one minibatch
    def _losses(self, pos, budget, act, old_lp, adv, team_ret, graph_of_row):
        """pos, budget, act int64 [mb, A]; old_lp, adv float [mb, A]; team_ret [mb]; graph_of_row int64 [mb]."""
        net, A, P, N = self.net, self.A, self.P, self.N
        mrx, pol = pos[:, 0], pos[:, 1:]
        mb = pos.shape[0]
        W1 = torch.stack([a[0].weight for a in net.actors])                     # [A, H, N]
        b1 = torch.stack([a[0].bias for a in net.actors])                       # [A, H]
        W2 = torch.stack([a[2].weight for a in net.actors])                     # [A, N, H]
        b2 = torch.stack([a[2].bias for a in net.actors])                       # [A, N]
        H = W1.shape[1]
        # The trainer's observations (mappo_trainer.py:173,197) as two [mb, N] matrices built without gradients: the
        # first layers are then plain matmuls — their backward is a matmul too, where a row-lookup form pays an
        # index_put with half a million colliding rows per minibatch (14 of round 2's 30 ms).
        oh_m = torch.zeros((mb, N), dtype=W1.dtype, device=pos.device).scatter_(1, mrx.unsqueeze(1), 1.0)
        oh_p = torch.zeros((mb, N), dtype=W1.dtype, device=pos.device).scatter_(1, pol, 1.0)
        lin = _OneHotLinear.apply
        h0 = lin(oh_m, W1[0].t())                                               # [mb, H]
        hp = lin(oh_p, W1[1:].reshape(P * H, N).t()).view(mb, P, H).transpose(0, 1)   # [P, mb, H]: all police actors in one matmul
        h = torch.relu(torch.cat([h0.unsqueeze(0), hp], 0) + b1.unsqueeze(1))   # [A, mb, H]
        logits = torch.baddbmm(b2.unsqueeze(1), h, W2.transpose(1, 2))          # [A, mb, N]
        # the affordable entries of every agent's ELL row: <= 16 per (row, agent)
        ent = self.ell[graph_of_row.unsqueeze(1), pos]                          # [mb, A, 16] int32
        nbr = (ent & 0xFFFF).long().clamp_max(N - 1)
        wgt = (ent >> 16) & 0xFFFF
        legal = wgt <= budget.unsqueeze(-1)                                     # padding weight 0xFFFF exceeds any budget
        l_ent = torch.gather(logits.transpose(0, 1), 2, nbr)                    # [mb, A, 16]
        l_ent = l_ent.masked_fill(~legal, float("-inf"))
        valid = act >= 0
        lse = torch.logsumexp(l_ent.masked_fill(~valid.unsqueeze(-1), 0.0), dim=-1)
        l_act = torch.gather(logits.transpose(0, 1), 2, act.clamp_min(0).unsqueeze(-1)).squeeze(-1)
        vf = valid.to(l_act.dtype)
        new_lp = (l_act - lse) * vf                                             # an agent without a legal action: ratio 1
        ratio = torch.exp(new_lp - old_lp * vf)
        surr = torch.min(ratio * adv, torch.clamp(ratio, 1.0 - self.clip, 1.0 + self.clip) * adv)
        actor_loss = -surr.mean()                                               # mappo_agent.py:284-291
        # CentralCritic on [mrx] + [police] * P (mappo_agent.py:32-44): the P copies of the police block share one input
        c1 = net.critic[0].weight.view(H, A, N)
        hc = torch.relu(lin(oh_m, c1[:, 0].t()) + lin(oh_p, c1[:, 1:].sum(1).t()) + net.critic[0].bias)
        value = net.critic[2](hc).squeeze(-1)
        critic_loss = torch.nn.functional.mse_loss(value, team_ret)             # :260-265
        return actor_loss, critic_loss

    def _step(self, st):
        al, cl = self._losses(st["pos"], st["budget"], st["act"], st["old_lp"], st["adv"], st["team_ret"], st["graph"])
        self.opt.zero_grad(set_to_none=False)
        (al + self.value_coef * cl).backward()
        self.opt.step()
        st["actor_loss"].copy_(al.detach())
        st["critic_loss"].copy_(cl.detach())

    # ------------------------------------------------------------------ a whole update
    def update(self, rec: Dict[str, torch.Tensor], returns: torch.Tensor, values: Optional[torch.Tensor] = None,
               generator: Optional[torch.Generator] = None):
        """One pass over the record in minibatches.  rec: `pos`, `budget`, `action`, `log_prob` [T, B, A] (an
        `env.alloc_rollout` record of a policy rollout); returns [T, B, A] (`collector.device_returns`); values [T, B]
        or None (advantage = standardised return).  Returns (actor_loss, critic_loss) of the last minibatch as device
        tensors — nothing here synchronises with the host."""
        if self.fused and rec.get("record") is not None:
            return self._update_fused(rec, returns, values, generator)
        T, B, A = rec["action"].shape
        R = T * B
        mb = min(self.minibatch, R)
        adv = returns if values is None else returns - values.unsqueeze(-1)
        adv = (adv - adv.mean()) / (adv.std() + 1e-8)                           # mappo_agent.py:256-258
        flat = {"pos": rec["pos"].reshape(R, A), "budget": rec["budget"].reshape(R, A), "act": rec["action"].reshape(R, A),
                "old_lp": rec["log_prob"].reshape(R, A), "adv": adv.reshape(R, A).float(),
                "team_ret": returns.reshape(R, A).sum(-1).float()}
        graph_rows = self.env_graph.repeat(T)                                   # row r = t * B + b -> board of env b
        perm = torch.randperm(R, device=self.device, generator=generator)
        shuf = {k: v[perm] for k, v in flat.items()}                            # ONE shuffle of the record per update:
        shuf["graph"] = graph_rows[perm]                                        # minibatches are contiguous slices of it
        if self._static is None or self._static["pos"].shape[0] != mb:
            z = lambda dt, *s: torch.zeros(s, dtype=dt, device=self.device)     # noqa: E731
            self._static = {"pos": z(torch.int64, mb, A), "budget": z(torch.int64, mb, A), "act": z(torch.int64, mb, A),
                            "old_lp": z(torch.float32, mb, A), "adv": z(torch.float32, mb, A), "team_ret": z(torch.float32, mb),
                            "graph": z(torch.int64, mb), "actor_loss": z(torch.float32), "critic_loss": z(torch.float32)}
            self._graph = None
        st = self._static
        nfull = R // mb
        for i in range(nfull):                                                  # (a ragged tail is dropped, as minibatch PPO does)
            for k in ("pos", "budget", "act", "old_lp", "adv", "team_ret", "graph"):
                st[k].copy_(shuf[k][i * mb:(i + 1) * mb])
            if not self.use_graph:
                self._step(st)
            else:
                if self._graph is None:                                        # first minibatch: eager (warm-up), then capture
                    self._step(st)
                    torch.cuda.synchronize(self.device)
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.device(self.device), torch.cuda.graph(g):
                        self._step(st)
                    self._graph = g
                else:
                    self._graph.replay()
        self.last_losses = (st["actor_loss"], st["critic_loss"])
        return self.last_losses
