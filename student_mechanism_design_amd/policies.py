"""Batched PyTorch-ROCm policies for the collector: the policy forward stays
in PyTorch, only the env is a HIP engine.

* `MappoPolicy` — the networks of `src/agent/mappo_agent.py:6-44` (per-agent 2-layer MLP actors with a
  softmax head, one central 2-layer MLP critic), evaluated for all B envs at once.  **Parity pinned**:
  tests/golden/mappo_networks_reference.npz holds weights / inputs / outputs of the unmodified reference
  networks (oracle/capture_mappo_networks.py); tests/test_policies_cpu.py loads them into this class.  Observations follow
  the trainer (`mappo_trainer.py:173,197`): MrX sees one-hot(MrX_pos), every police sees the multi-hot
  of all police positions; the critic sees their concatenation.
* `AntiSymmetricConvEll` / `GnnQModel` / `GnnQPolicy` — the model of `src/agent/gnn_agent.py:230-257` (2 x
  torch_geometric `AntiSymmetricConv` + Linear -> per-node Q) on GATHER tables over the <= 16 sources of a node
  (`GcnTables`), never an [N, N] matrix; `DeviceGnnPolicy` = the same forward + masked arg-max as one HIP kernel
  (`sy_gnn_q_act`).  torch_geometric is absent offline: the layer follows the published form
  x + eps * tanh((W - W^T - gamma I) x + A^ Theta x + b) (Gravina et al., ICLR 2023) with PyG's defaults
  (phi = GCNConv(in, in, bias=False), num_iters = 1) and is pinned to an independent float64 restatement
  (oracle/gnn_oracle.py); against the library itself parity stays unpinned.
* `ppo_loss` — the clipped surrogate / critic MSE of `MappoAgent.ppo_update` (mappo_agent.py:260-293).
"""
from typing import Dict, Optional

import torch
import torch.nn as nn

from .collector import masked_categorical_sample


def one_hot_nodes(pos: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """pos int[B] or int[B, K] -> float[B, N] (multi-hot over K)."""
    if pos.dim() == 1:
        pos = pos.unsqueeze(-1)
    out = torch.zeros((pos.shape[0], num_nodes), dtype=torch.float32, device=pos.device)
    out.scatter_(1, pos.long(), 1.0)
    return out


class MappoPolicy(nn.Module):
    def __init__(self, num_nodes: int, num_police: int, hidden_size: int = 64):
        super().__init__()
        self.N, self.P, self.A = num_nodes, num_police, num_police + 1

        def actor():
            return nn.Sequential(nn.Linear(num_nodes, hidden_size), nn.ReLU(), nn.Linear(hidden_size, num_nodes))

        self.actors = nn.ModuleList([actor() for _ in range(self.A)])       # AgentPolicy, mappo_agent.py:6-29
        self.critic = nn.Sequential(nn.Linear(num_nodes * self.A, hidden_size), nn.ReLU(),
                                    nn.Linear(hidden_size, 1))              # CentralCritic, :32-44

    def observations(self, obs: Dict[str, torch.Tensor]):
        mrx = one_hot_nodes(obs["MrX_pos"], self.N)
        pol = one_hot_nodes(obs["Polices_pos"], self.N)
        return mrx, pol

    def probs(self, obs: Dict[str, torch.Tensor]) -> torch.Tensor:
        """[B, A, N] action probabilities before masking."""
        mrx, pol = self.observations(obs)
        out = [torch.softmax(self.actors[0](mrx), -1)]
        out += [torch.softmax(self.actors[k](pol), -1) for k in range(1, self.A)]
        return torch.stack(out, dim=1)

    def value(self, obs: Dict[str, torch.Tensor]) -> torch.Tensor:
        mrx, pol = self.observations(obs)
        g = torch.cat([mrx] + [pol] * self.P, dim=-1)
        return self.critic(g).squeeze(-1)

    @torch.no_grad()
    def act(self, obs: Dict[str, torch.Tensor], generator: Optional[torch.Generator] = None):
        """Collector callback: masked sampling as in MappoAgent.select_action (mappo_agent.py:87-142)."""
        mask = obs["action_mask"]
        a, logp, _ = masked_categorical_sample(self.probs(obs), mask, generator=generator)
        a = torch.where(mask.sum(-1) == 0, torch.full_like(a, -1), a)   # nothing legal -> DEFAULT_ACTION
        return a.to(torch.int32), logp.float(), self.value(obs)

    # ---- inference path for the collector: the same networks evaluated without materialising the one-hot
    # observations.  A Linear layer applied to a one-hot (multi-hot) vector is a row lookup (a sum of row
    # lookups) in its weight, and the per-agent second layers are one batched matmul: ~12 kernels per step
    # instead of ~30.  Same parameters, same function up to float summation order.
    def probs_fast(self, obs: Dict[str, torch.Tensor]) -> torch.Tensor:
        mrx, pol = obs["MrX_pos"].long(), obs["Polices_pos"].long()            # [B], [B, P]
        W1 = torch.stack([a[0].weight for a in self.actors])                    # [A, H, N]
        b1 = torch.stack([a[0].bias for a in self.actors])                      # [A, H]
        W2 = torch.stack([a[2].weight for a in self.actors])                    # [A, N, H]
        b2 = torch.stack([a[2].bias for a in self.actors])                      # [A, N]
        h0 = W1[0].t()[mrx]                                                     # [B, H]
        hp = W1[1:].transpose(1, 2)[:, pol].sum(2)                              # [P, B, H]: sum over the P police nodes
        h = torch.relu(torch.cat([h0.unsqueeze(0), hp], 0) + b1.unsqueeze(1))   # [A, B, H]
        logits = torch.baddbmm(b2.unsqueeze(1), h, W2.transpose(1, 2))          # [A, B, N]
        return torch.softmax(logits, -1).transpose(0, 1)                        # [B, A, N]

    def value_fast(self, obs: Dict[str, torch.Tensor]) -> torch.Tensor:
        mrx, pol = obs["MrX_pos"].long(), obs["Polices_pos"].long()
        N, P = self.N, self.P
        blocks = torch.arange(1, P + 1, device=pol.device).view(1, P, 1) * N     # the police multi-hot is repeated P times
        idx = torch.cat([mrx.unsqueeze(1), (pol.unsqueeze(1) + blocks).reshape(pol.shape[0], P * P)], 1)
        h = torch.relu(self.critic[0].weight.t()[idx].sum(1) + self.critic[0].bias)
        return self.critic[2](h).squeeze(-1)

    @torch.no_grad()
    def act_fast(self, obs: Dict[str, torch.Tensor], generator: Optional[torch.Generator] = None):
        mask = obs["action_mask"]
        a, logp, _ = masked_categorical_sample(self.probs_fast(obs), mask, generator=generator)
        a = torch.where(mask.sum(-1) == 0, torch.full_like(a, -1), a)
        return a.to(torch.int32), logp.float(), self.value_fast(obs)


    @torch.no_grad()
    def act_device(self, obs: Dict[str, torch.Tensor], sampler):
        """Collector callback with the HIP sampling kernel (`collector.DeviceMaskedSampler`)."""
        a, logp, _ = sampler(self.probs_fast(obs), obs["action_mask"], default_on_empty=True)
        return a, logp, self.value_fast(obs)


class DeviceMappoPolicy:
    """`MappoPolicy.act` as ONE HIP kernel (`sy_mappo_policy_act`, include/sy_env.h): actor MLPs on the
    trainer's observations, masked sampling, central critic — MappoAgent.select_action for every
    (env, agent) of the batch in a single launch.  The first layers are row lookups in transposed weights, the
    second layers run from LDS; draws come from the engine's Philox stream with a device-resident call
    counter (fresh numbers on every replay of a captured HIP graph).  Call `refresh()` after the wrapped
    module's parameters change.  Fails loudly without the engine library / a GPU."""

    def __init__(self, net: "MappoPolicy", seed: int = 0):
        from . import _lib
        self._lib_mod, self.lib, self.net = _lib, _lib.load(), net
        self.device = next(net.parameters()).device
        if self.device.type != "cuda":
            raise _lib.EngineError("DeviceMappoPolicy needs the module on a GPU; there is no CPU fallback")
        self.H = net.actors[0][0].out_features
        if self.H > 128 or self.H % 4:
            raise ValueError("the fused kernels support hidden sizes that are multiples of 4, up to 128")
        self.seed = int(seed) & (2**64 - 1)
        self.counter = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._out = None
        self.refresh()

    @torch.no_grad()
    def refresh(self):
        """Re-pack the module's parameters in the kernels' layouts (after every optimiser step).  The packed
        buffers are allocated once and refreshed in place, so an env that holds them (`set_policy`) and captured
        HIP graphs keep seeing the current weights."""
        import ctypes as C
        f32 = dict(dtype=torch.float32, device=self.device)
        acts = self.net.actors
        W1 = torch.stack([a[0].weight for a in acts])                   # [A, H, N]
        b1 = torch.stack([a[0].bias for a in acts])                     # [A, H]
        W2 = torch.stack([a[2].weight for a in acts])                   # [A, N, H] (torch layout: a node's row contiguous)
        b2 = torch.stack([a[2].bias for a in acts])                     # [A, N]
        packed = {
            "w1t": W1.transpose(1, 2),                                  # [A, N, H]
            "b1": b1,
            "w2t": W2.transpose(1, 2),                                  # [A, H, N]
            "b2": b2,
            "c1t": self.net.critic[0].weight.t(),                       # [N * A, H]
            "cb1": self.net.critic[0].bias,
            "c2": self.net.critic[2].weight.reshape(-1),                # [H]
            "cb2": self.net.critic[2].bias,
            "w2": W2,
            "logit_bound": self._logit_bound(W1, b1, W2, b2),           # [A] (the in-kernel underflow rule's cheap test)
        }
        if getattr(self, "_packed", None) is None:
            self._packed = {k: v.to(**f32).contiguous().clone() for k, v in packed.items()}
            self._w = self._lib_mod.MappoWeights(*[C.c_void_p(self._packed[k].data_ptr()) for k in (
                "w1t", "b1", "w2t", "b2", "c1t", "cb1", "c2", "cb2", "w2", "logit_bound")])
        else:
            for k, v in packed.items():
                self._packed[k].copy_(v)

    @torch.no_grad()
    def _logit_bound(self, W1, b1, W2, b2) -> torch.Tensor:
        """[A] an upper bound of every logit actor a can produce, whatever the observation: the hidden units are
        relu(b1 + sum of `hot` columns of W1) (hot = 1: MrX's one-hot input; P: the police actors' multi-hot input), so
        h_k <= hmax_k = max(b1_k, 0) + hot * max_n max(W1[k, n], 0) and logit_n <= b2_n + sum_k max(W2[n, k], 0) * hmax_k.
        All actors at once on the stacked parameters (W1 [A, H, N], b1 [A, H], W2 [A, N, H], b2 [A, N])."""
        hot = torch.full((W1.shape[0], 1), float(self.net.P), device=W1.device)
        hot[0] = 1.0
        hmax = b1.clamp_min(0) + hot * W1.clamp_min(0).amax(dim=2)                               # [A, H]
        return (b2 + torch.bmm(W2.clamp_min(0), hmax.unsqueeze(-1)).squeeze(-1)).amax(dim=1).float()

    @torch.no_grad()
    def act(self, obs: Dict[str, torch.Tensor], want_probs: bool = False):
        """Collector callback: (actions int32 [B, A], log_prob [B, A], value [B]); with want_probs a 4th item,
        the actors' softmax [B, A, N].  The first three are persistent buffers, overwritten by the next call."""
        import ctypes as C
        pos, mask = obs["agent_position"], obs["action_mask"]
        B, A = pos.shape
        N = self.net.N
        if pos.dtype != torch.int32 or not pos.is_contiguous():
            pos = pos.to(torch.int32).contiguous()
        mk = mask.view(torch.uint8) if mask.dtype == torch.bool else mask
        if mk.stride(-1) != 1 or mk.stride(0) != A * mk.stride(1):     # rows may be padded (the engine's NS), not scattered
            mk = mk.contiguous()
        if self._out is None or self._out[0].shape[0] != B:
            self._out = (torch.empty((B, A), dtype=torch.int32, device=self.device),
                         torch.empty((B, A), dtype=torch.float32, device=self.device),
                         torch.empty((B,), dtype=torch.float32, device=self.device))
        act, logp, val = self._out
        probs = torch.empty((B, A, N), dtype=torch.float32, device=self.device) if want_probs else None
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        with torch.cuda.device(self.device):
            self._lib_mod.check(self.lib.sy_mappo_policy_act(
                C.c_void_p(pos.data_ptr()), C.c_void_p(mk.data_ptr()), C.c_int64(mk.stride(1)), C.byref(self._w),
                B, A - 1, N, self.H, C.c_uint64(self.seed), C.c_uint64(0), C.c_void_p(self.counter.data_ptr()),
                C.c_void_p(act.data_ptr()), C.c_void_p(logp.data_ptr()), C.c_void_p(val.data_ptr()),
                C.c_void_p(probs.data_ptr()) if probs is not None else None, stream), "sy_mappo_policy_act")
        self.counter.add_(1)
        return (act, logp, val, probs) if want_probs else (act, logp, val)


class GcnTables:
    """GCNConv's normalised propagation per board as gather tables (graph.py::gcn_tables): for every target node its
    source nodes and coefficients 1 / sqrt(deg(src) deg(dst)) — rows have at most 16 sources, so A^ x is a gather over
    a [G, N, K] index table, never an [N, N] matrix.  `directed=True` (default) is the reference's data flow: the
    stored edge list is handed to the model once per edge (training/utils.py:170), messages flow source -> target."""

    def __init__(self, boards, device="cpu", directed: bool = True):
        from .graph import gcn_tables
        nbr, coef, self_coef = gcn_tables(boards, directed=directed)
        self.width = max(1, int((nbr >= 0).sum(-1).max()))                        # widest row actually used
        self.nbr = torch.from_numpy(nbr).to(device)                                # int16 [G, N, 16] (kernel layout)
        self.coef = torch.from_numpy(coef).to(device)                              # float32 [G, N, 16]
        self.self_coef = torch.from_numpy(self_coef).to(device)                    # float32 [G, N]
        self.directed = directed
        # the kernel's packed form: [G, N, K, 2] = {source node, float bits of the coefficient}, padding {0, 0.0}
        import numpy as _np
        K = self.width
        packed = _np.zeros(nbr.shape[:2] + (K, 2), dtype=_np.uint32)
        packed[..., 0] = _np.where(nbr[:, :, :K] >= 0, nbr[:, :, :K], 0).astype(_np.uint32)
        packed[..., 1] = _np.where(nbr[:, :, :K] >= 0, coef[:, :, :K], 0.0).astype(_np.float32).view(_np.uint32)
        self.packed = torch.from_numpy(packed.view(_np.int32)).to(device).contiguous()

    def for_envs(self, env_graph: torch.Tensor):
        """(idx int64 [B, N, K], coef [B, N, K], self_coef [B, N]) of each env's board, trimmed to the widest row."""
        g = env_graph.long()
        K = self.width
        return self.nbr[g, :, :K].long().clamp_min(0), self.coef[g, :, :K], self.self_coef[g]


class AntiSymmetricConvEll(nn.Module):
    """torch_geometric's AntiSymmetricConv with its defaults as the reference uses it (gnn_agent.py:233-246:
    phi = GCNConv(F, F, bias=False), num_iters = 1, epsilon = gamma = 0.1, act = tanh), restated on gather tables:
        x' = x + epsilon * tanh(x (W - W^T - gamma I)^T + A^ (x Theta^T) + b)
    torch_geometric is absent offline: pinned to oracle/gnn_oracle.py (float64, dense A^ built edge by edge from the
    published formula), parity with the library itself stays unpinned."""

    def __init__(self, channels: int, epsilon: float = 0.1, gamma: float = 0.1):
        super().__init__()
        self.W = nn.Parameter(torch.empty(channels, channels))
        self.bias = nn.Parameter(torch.zeros(channels))
        self.phi = nn.Linear(channels, channels, bias=False)     # GCNConv's linear map (weight: out, in)
        self.epsilon, self.gamma = epsilon, gamma
        nn.init.kaiming_uniform_(self.W, a=5 ** 0.5)

    def antisymmetric(self) -> torch.Tensor:
        eye = torch.eye(self.W.shape[0], device=self.W.device, dtype=self.W.dtype)
        return self.W - self.W.t() - self.gamma * eye

    def forward(self, x: torch.Tensor, tables) -> torch.Tensor:
        """x [B, N, F]; tables = GcnTables.for_envs(env_graph)."""
        idx, coef, self_coef = tables
        xt = self.phi(x)                                                          # Theta x
        msg = self_coef.unsqueeze(-1) * xt
        F_ = xt.shape[-1]
        for k in range(idx.shape[-1]):                                            # at most 16 gathers of [B, N, F]
            msg = msg + coef[:, :, k, None] * torch.gather(xt, 1, idx[:, :, k, None].expand(-1, -1, F_))
        h = x @ self.antisymmetric().t() + msg + self.bias
        return x + self.epsilon * torch.tanh(h)


class GnnQModel(nn.Module):
    """GNNModel (gnn_agent.py:230-257): conv1 -> relu -> conv2 -> relu -> Linear(F, 1) -> Q per node."""

    def __init__(self, node_feature_size: int):
        super().__init__()
        self.conv1 = AntiSymmetricConvEll(node_feature_size)
        self.conv2 = AntiSymmetricConvEll(node_feature_size)
        self.out = nn.Linear(node_feature_size, 1)

    def forward(self, x: torch.Tensor, tables) -> torch.Tensor:
        x = torch.relu(self.conv1(x, tables))
        x = torch.relu(self.conv2(x, tables))
        return self.out(x).squeeze(-1)                                            # [B, N]


class GnnQPolicy(nn.Module):
    """The GNN trainer's two agents (gnn_trainer.py:148-178): MrX's model and ONE model shared by all police, both on
    the node features of training/utils.py:176-200 — column a = one-hot node of agent a (F = number of agents) —
    plus, optionally, the police belief over MrX's node as one more column (engine extension, default off)."""

    def __init__(self, num_agents: int, with_belief: bool = False):
        super().__init__()
        self.A = int(num_agents)
        self.F = self.A + (1 if with_belief else 0)
        self.with_belief = with_belief
        self.mrx = GnnQModel(self.F)
        self.police = GnnQModel(self.F)

    def features(self, obs: Dict[str, torch.Tensor], num_nodes: int) -> torch.Tensor:
        pos = obs["agent_position"].long()                                  # [B, A]
        x = torch.zeros((pos.shape[0], num_nodes, pos.shape[1]), device=pos.device)
        x.scatter_(1, pos.unsqueeze(1), 1.0)                                # node_features, training/utils.py:176-200
        if self.with_belief:
            x = torch.cat([x, obs["belief_map"].unsqueeze(-1)], dim=-1)
        return x

    def q_values(self, obs: Dict[str, torch.Tensor], tables, num_nodes: int) -> torch.Tensor:
        """[B, 2, N]: MrX's model, the police model."""
        x = self.features(obs, num_nodes)
        return torch.stack([self.mrx(x, tables), self.police(x, tables)], dim=1)

    @torch.no_grad()
    def act_greedy(self, obs: Dict[str, torch.Tensor], tables):
        """GNNAgent.select_action with epsilon = 0 for every agent (gnn_agent.py:62-74): masked arg-max, -1 (None)
        without a valid action; the police agents share the police model's Q map."""
        mask = obs["action_mask"]
        q2 = self.q_values(obs, tables, mask.shape[-1])
        A = mask.shape[1]
        q = torch.cat([q2[:, :1], q2[:, 1:].expand(-1, A - 1, -1)], dim=1)
        q = q.masked_fill(~mask, float("-inf"))
        a = q.argmax(-1)
        a = torch.where(mask.sum(-1) == 0, torch.full_like(a, -1), a)
        return a.to(torch.int32), None, None


class DeviceGnnPolicy:
    """`GnnQPolicy.act_greedy` (+ epsilon-greedy exploration) as ONE HIP kernel (`sy_gnn_q_act`, include/sy_env.h): one
    wave per env, node features in registers, the transformed features gathered through LDS over the <= 16 sources of
    every node, both models, the masked arg-max of every agent.  Call `refresh()` after the wrapped module's parameters
    change.  Fails loudly without the engine library / a GPU."""

    def __init__(self, net: GnnQPolicy, tables: GcnTables, env_graph: torch.Tensor, seed: int = 0, explore_eps: float = 0.0):
        from . import _lib
        self._lib_mod, self.lib, self.net, self.tables = _lib, _lib.load(), net, tables
        self.device = next(net.parameters()).device
        if self.device.type != "cuda":
            raise _lib.EngineError("DeviceGnnPolicy needs the module on a GPU; there is no CPU fallback")
        if net.F > 9:
            raise ValueError("sy_gnn_q_act supports at most 9 node features")
        self.env_graph = env_graph.to(device=self.device, dtype=torch.int32).contiguous()
        self.seed, self.explore_eps = int(seed) & (2**64 - 1), float(explore_eps)
        self.counter = torch.zeros(1, dtype=torch.int64, device=self.device)
        self.FP = int(self.lib.sy_gnn_padded_features(net.F))
        self._packed = torch.zeros((int(self.lib.sy_gnn_param_floats(net.F)), 2), dtype=torch.float32, device=self.device)
        self._out = None
        self.refresh()

    @torch.no_grad()
    def refresh(self):
        F_, FP = self.net.F, self.FP
        for m, model in enumerate((self.net.mrx, self.net.police)):
            parts = []
            for conv in (model.conv1, model.conv2):
                Wa = torch.zeros((FP, FP), device=self.device)
                Wa[:F_, :F_] = conv.antisymmetric()
                Th = torch.zeros((FP, FP), device=self.device)
                Th[:F_, :F_] = conv.phi.weight
                b = torch.zeros(FP, device=self.device)
                b[:F_] = conv.bias
                parts += [Wa.reshape(-1), Th.reshape(-1), b]
            wo = torch.zeros(FP, device=self.device)
            wo[:F_] = model.out.weight.reshape(-1)
            parts += [wo, model.out.bias.reshape(1), torch.tensor([model.conv1.epsilon], device=self.device)]
            self._packed[:, m].copy_(torch.cat([p.float() for p in parts]))

    @torch.no_grad()
    def act(self, obs: Dict[str, torch.Tensor], want_q: bool = False):
        """Collector callback: (actions int32 [B, A], None, None); with want_q a 4th item, Q [B, 2, N]."""
        import ctypes as C
        pos, mask = obs["agent_position"], obs["action_mask"]
        B, A = pos.shape
        N = mask.shape[-1]
        if pos.dtype != torch.int32 or not pos.is_contiguous():
            pos = pos.to(torch.int32).contiguous()
        mk = mask.view(torch.uint8) if mask.dtype == torch.bool else mask
        if mk.stride(-1) != 1 or mk.stride(0) != A * mk.stride(1):
            mk = mk.contiguous()
        bel, bel_stride = None, 0
        if self.net.with_belief:
            bel = obs["belief_map"]
            if bel.stride(-1) != 1:
                bel = bel.contiguous()
            bel_stride = bel.stride(0)
        if self._out is None or self._out.shape[0] != B:
            self._out = torch.empty((B, A), dtype=torch.int32, device=self.device)
        q = torch.empty((B, 2, N), dtype=torch.float32, device=self.device) if want_q else None
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        with torch.cuda.device(self.device):
            self._lib_mod.check(self.lib.sy_gnn_q_act(
                p(pos), p(bel), C.c_int64(bel_stride), p(mk), C.c_int64(mk.stride(1)), p(self.tables.packed), self.tables.width,
                p(self.tables.self_coef), p(self.env_graph), p(self._packed), B, A - 1, N, self.net.F,
                C.c_float(self.explore_eps), C.c_uint64(self.seed), C.c_uint64(0), p(self.counter), p(self._out), p(q), stream),
                "sy_gnn_q_act")
        self.counter.add_(1)
        return (self._out, None, None, q) if want_q else (self._out, None, None)


def ppo_loss(new_log_prob, old_log_prob, advantages, values, returns, clip: float = 0.2):
    """mappo_agent.py:260-293: critic MSE + clipped surrogate (no entropy / value coefficients)."""
    ratio = torch.exp(new_log_prob - old_log_prob)
    surr = torch.min(ratio * advantages, torch.clamp(ratio, 1 - clip, 1 + clip) * advantages)
    return -surr.mean(), torch.nn.functional.mse_loss(values, returns)
