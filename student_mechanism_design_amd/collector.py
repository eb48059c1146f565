"""On-device rollout collector (replaces the per-step Python loop of
src/training/mappo_trainer.py:161-287 / gnn_trainer.py:194-291).

* `RolloutCollector.collect()` — policy-driven: per step one batched policy forward (PyTorch-ROCm),
  one `sy_env_step` launch, everything written into preallocated `[T, B, ...]` device tensors.
  With `policy=None` it is the fused in-kernel random-policy rollout (`sy_env_rollout`, one launch).
* `masked_categorical_sample` — batched restatement of `MappoAgent.select_action`
  (agent/mappo_agent.py:87-142): probs*mask, uniform fallback, renormalise with +1e-8, sample.
* `discounted_returns` / `standardized_advantages` — `MappoAgent.ppo_update`'s return and advantage
  maths (mappo_agent.py:247-258) over the time axis; `gae` is the lambda-generalisation that reduces
  to those returns at lambda = 1 with a zero bootstrap (the parity check SURVEY.md section 8a-13 names).
* `TrajectoryExchange` / `gather_trajectories` — the ONE exchange of the multi-GPU path, zero-copy: the rollout's
  arena IS the send buffer, one `all_gather_into_tensor` into a preallocated `[world, bytes]` buffer, views out
  (RCCL on GPU, gloo in the CPU tests); rollouts themselves need no communication because episodes are
  independent.  `allreduce_gradients` is the cheaper data-parallel alternative (0.77 MB instead of GBs).
"""
from typing import Callable, Dict, Optional

import torch
import torch.distributed as dist


def masked_categorical_sample(probs: torch.Tensor, mask: torch.Tensor, generator: Optional[torch.Generator] = None):
    """probs, mask: [..., N].  Returns (action int64[...], log_prob[...], normalised probs)."""
    m = mask.to(probs.dtype)
    p = probs * m
    s = p.sum(-1, keepdim=True)
    msum = m.sum(-1, keepdim=True)
    uniform_mask = m / msum.clamp_min(1e-8)
    uniform_all = torch.full_like(p, 1.0 / p.shape[-1])
    fallback = torch.where(msum > 1e-8, uniform_mask, uniform_all)     # mappo_agent.py:120-127
    p = torch.where(s <= 1e-8, fallback, p / (s + 1e-8))               # :128-129
    flat = p.reshape(-1, p.shape[-1])
    # Categorical(probs=...) renormalises internally; multinomial on the same rows is equivalent
    a = torch.multinomial(flat / flat.sum(-1, keepdim=True), 1, generator=generator).squeeze(-1)
    norm = flat / flat.sum(-1, keepdim=True)
    logp = torch.log(norm.gather(-1, a.unsqueeze(-1)).squeeze(-1))
    return a.reshape(p.shape[:-1]), logp.reshape(p.shape[:-1]), p


class DeviceMaskedSampler:
    """`masked_categorical_sample` as ONE HIP kernel (`sy_masked_categorical_sample`, include/sy_env.h): the
    ~25 elementwise / reduction launches of the torch version — the bulk of a policy-in-the-loop step —
    become one wave per (env, agent) row.  Same normalisation rules (mappo_agent.py:112-142); draws come
    from the engine's Philox stream with a device-resident call counter, so a captured HIP graph samples
    fresh numbers on every replay.  Fails loudly without the engine library / a GPU."""

    def __init__(self, device, seed: int = 0):
        from . import _lib
        self._lib_mod = _lib
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.EngineError("DeviceMaskedSampler needs a GPU tensor device; there is no CPU fallback")
        self.seed = int(seed) & (2**64 - 1)
        self.counter = torch.zeros(1, dtype=torch.int64, device=self.device)   # advanced after every call

    def __call__(self, probs: torch.Tensor, mask: torch.Tensor, default_on_empty: bool = True, want_probs: bool = False):
        """probs float32 [..., N]; mask uint8 / bool [..., >= N] (row stride may exceed N: the engine's NS).
        Returns (action int32 [...], log_prob float32 [...], normalised probs [..., N] or None)."""
        import ctypes as C
        N = probs.shape[-1]
        lead = tuple(probs.shape[:-1])
        pr = probs.to(torch.float32).reshape(-1, N)
        if pr.stride(-1) != 1:
            pr = pr.contiguous()
        mk = mask.view(torch.uint8) if mask.dtype == torch.bool else mask
        mk = mk.reshape(-1, mk.shape[-1])
        if mk.stride(-1) != 1:
            mk = mk.contiguous()
        if mk.shape[0] != pr.shape[0] or mk.shape[-1] < N:
            raise ValueError("mask must have the rows of probs and at least N columns")
        rows = pr.shape[0]
        act = torch.empty(rows, dtype=torch.int32, device=self.device)
        logp = torch.empty(rows, dtype=torch.float32, device=self.device)
        norm = torch.empty((rows, N), dtype=torch.float32, device=self.device) if want_probs else None
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        with torch.cuda.device(self.device):
            self._lib_mod.check(self.lib.sy_masked_categorical_sample(
                C.c_void_p(pr.data_ptr()), C.c_int64(pr.stride(0)), C.c_void_p(mk.data_ptr()), C.c_int64(mk.stride(0)),
                rows, N, C.c_uint64(self.seed), C.c_uint64(0), C.c_void_p(self.counter.data_ptr()),
                1 if default_on_empty else 0, C.c_void_p(act.data_ptr()), C.c_void_p(logp.data_ptr()),
                C.c_void_p(norm.data_ptr()) if norm is not None else None, stream), "sy_masked_categorical_sample")
        self.counter.add_(1)
        return act.reshape(lead), logp.reshape(lead), (norm.reshape(lead + (N,)) if norm is not None else None)


def discounted_returns(rewards: torch.Tensor, dones: torch.Tensor, gamma: float) -> torch.Tensor:
    """R_t = r_t + gamma * R_{t+1} * (1 - done_t) along dim 0 (mappo_agent.py:248-254)."""
    T = rewards.shape[0]
    out = torch.zeros_like(rewards)
    run = torch.zeros_like(rewards[0])
    d = dones.to(rewards.dtype)
    while d.dim() < rewards.dim():
        d = d.unsqueeze(-1)
    for t in range(T - 1, -1, -1):
        run = rewards[t] + gamma * run * (1.0 - d[t])
        out[t] = run
    return out


def standardized_advantages(returns: torch.Tensor, values: torch.Tensor) -> torch.Tensor:
    """adv = R - V, standardised with std + 1e-8 (mappo_agent.py:256-258)."""
    adv = returns - values
    if adv.numel() > 1:
        adv = (adv - adv.mean()) / (adv.std() + 1e-8)
    return adv


def gae(rewards, values, dones, last_value, gamma: float, lam: float):
    """Generalised advantage estimation over dim 0; returns (advantages, returns = adv + values)."""
    T = rewards.shape[0]
    adv = torch.zeros_like(rewards)
    d = dones.to(rewards.dtype)
    while d.dim() < rewards.dim():
        d = d.unsqueeze(-1)
    run = torch.zeros_like(rewards[0])
    nxt = last_value
    for t in range(T - 1, -1, -1):
        nd = 1.0 - d[t]
        delta = rewards[t] + gamma * nxt * nd - values[t]
        run = delta + gamma * lam * nd * run
        adv[t] = run
        nxt = values[t]
    return adv, adv + values


def device_returns(reward: torch.Tensor, done_a: torch.Tensor, gamma: float, done_b: Optional[torch.Tensor] = None,
                   values: Optional[torch.Tensor] = None, lam: Optional[float] = None,
                   last_value: Optional[torch.Tensor] = None, dtype: torch.dtype = torch.float32):
    """Returns and advantages of a whole [T, B, A] rollout in ONE HIP launch (`sy_returns_advantages`,
    include/sy_env.h) instead of a Python loop over T issuing ~5 torch kernels per step.

    lam=None: the reference's recurrence, mappo_agent.py:247-258 — R_t = r_t + gamma * R_{t+1} * (1 - done_t),
    adv = R - V; with dtype=float32 it reproduces the reference's float32 tensors bit for bit.
    lam given: GAE(gamma, lam) with bootstrap `last_value`; returns = adv + V.
    `values` / `last_value` are float32 (a critic's output): a float64 critic is rounded to float32 before the recurrence,
    also with dtype=float64 — that dtype widens the reward arithmetic, not the values.
    `reward` [T, B, A] float32/float64, any T / B strides (the packed record's `reward` view is read in place);
    done = done_a | done_b, each [T, B] uint8 / bool / int32 (the record's `terminated`, `truncated`);
    `values` [T, B] (central critic) or [T, B, A]; `last_value` [B] or [B, A].  Returns (returns, advantages),
    contiguous [T, B, A] of `dtype` (float32 or float64).  No CPU fallback."""
    import ctypes as C
    from . import _lib
    lib = _lib.load()
    dev = reward.device
    if dev.type != "cuda":
        raise _lib.EngineError("device_returns needs GPU tensors; there is no CPU fallback "
                               "(`discounted_returns` / `gae` are the plain-torch forms)")
    if reward.dim() != 3 or reward.stride(-1) != 1 or reward.dtype not in (torch.float32, torch.float64):
        raise ValueError("reward must be float32/float64 [T, B, A] with a contiguous last dimension")
    if dtype not in (torch.float32, torch.float64):
        raise ValueError("dtype must be float32 or float64")
    T, B, A = reward.shape

    def flag(t, name):
        if t is None:
            return None
        if t.dtype == torch.bool:
            t = t.view(torch.uint8)
        if t.dtype not in (torch.uint8, torch.int32) or tuple(t.shape) != (T, B) or t.device != dev:
            raise ValueError(f"{name} must be uint8 / bool / int32 [T, B] on {dev}")
        return t

    da, db = flag(done_a, "done_a"), flag(done_b, "done_b")
    if db is not None and (db.dtype != da.dtype or db.stride() != da.stride()):
        db = db.to(da.dtype).contiguous()
        da = da.contiguous()
    vs = (0, 0, 0)
    if values is not None:
        values = values.to(device=dev, dtype=torch.float32)
        if tuple(values.shape) == (T, B):
            vs = (values.stride(0), values.stride(1), 0)
        elif tuple(values.shape) == (T, B, A):
            vs = tuple(values.stride())
        else:
            raise ValueError("values must be [T, B] or [T, B, A]")
    lv = (0, 0)
    if last_value is not None:
        last_value = last_value.to(device=dev, dtype=torch.float32)
        if tuple(last_value.shape) == (B,):
            lv = (last_value.stride(0), 0)
        elif tuple(last_value.shape) == (B, A):
            lv = tuple(last_value.stride())
        else:
            raise ValueError("last_value must be [B] or [B, A]")
    if lam is not None and values is None:
        raise ValueError("GAE needs values")
    ret = torch.empty((T, B, A), dtype=dtype, device=dev)
    adv = torch.empty((T, B, A), dtype=dtype, device=dev)
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
    args = _lib.ReturnsArgs(T, B, A, 0 if lam is None else 1, p(reward), 1 if reward.dtype == torch.float64 else 0,
                            reward.stride(0), reward.stride(1), p(da), p(db), da.element_size(), da.stride(0), da.stride(1),
                            p(values), vs[0], vs[1], vs[2], p(last_value), lv[0], lv[1], float(gamma),
                            1.0 if lam is None else float(lam), 1 if dtype == torch.float64 else 0, p(ret), p(adv))
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.sy_returns_advantages(C.byref(args), stream), "sy_returns_advantages")
    return ret, adv


def pack_record(record: Dict[str, Optional[torch.Tensor]]):
    """Generic fallback for records that are NOT arena-backed (plain dicts of separately allocated tensors):
    flatten {name: [T, B, ...]} into one uint8 buffer plus the metadata to unpack it.  Costs one full copy;
    `env.alloc_rollout` records never take this path."""
    names = sorted(k for k, v in record.items() if v is not None and k != "record")  # `record` aliases the named views
    parts, meta = [], []
    for k in names:
        v = record[k].contiguous()
        raw = v.view(torch.uint8).reshape(-1)
        parts.append(raw)
        meta.append((k, v.dtype, tuple(v.shape), raw.numel()))
    return torch.cat(parts), meta


def unpack_record(buf: torch.Tensor, meta, world: int):
    """Inverse of pack_record for `world` concatenated rank buffers; ranks are concatenated on dim 1 (B)."""
    per = sum(m[3] for m in meta)
    out = {}
    off = 0
    for k, dtype, shape, nbytes in meta:
        pieces = [buf[r * per + off: r * per + off + nbytes].view(dtype).reshape(shape) for r in range(world)]
        out[k] = pieces[0] if world == 1 else torch.cat(pieces, dim=1)
        off += nbytes
    return out


class TrajectoryExchange:
    """The ONE exchange of the multi-GPU path (BASELINE configs[3]: "RCCL gather at PPO update"), zero-copy:

    * the send buffer is the rollout's own arena (`env.alloc_rollout` carves every record tensor out of one
      contiguous byte buffer), so nothing is packed;
    * the receive buffer `[world, arena_bytes]` is allocated once and reused for every update;
    * the result is a dict of VIEWS `[world, T, B_local, ...]` of that buffer (rank-major; a PPO update flattens
      (world, T, B) anyway) — no `torch.cat`, no extra pass over the data.

    One `all_gather_into_tensor` per update: RCCL over xGMI on GPU (a direct all-gather drives all 7 links of a
    GPU at once), gloo in the CPU tests.  Cheaper alternative when every rank trains on its own shard: keep the
    trajectories local and all-reduce the GRADIENTS (`allreduce_gradients`): 0.77 MB for the N=200 / H=64 MAPPO
    networks instead of 2.06 GB per rank at T=256 — latency-bound, and the trajectories never move."""

    def __init__(self, record, group=None):
        if not hasattr(record, "arena"):
            raise ValueError("TrajectoryExchange needs an arena-backed record (env.alloc_rollout)")
        # only the arena travels: a tensor hung on the record from outside it would be dropped without a word
        lo, hi = record.arena.data_ptr(), record.arena.data_ptr() + record.arena.numel()
        for k, v in record.items():
            if isinstance(v, torch.Tensor) and not (lo <= v.data_ptr() and v.data_ptr() + v.numel() * v.element_size() <= hi + 256):
                raise ValueError(f"record tensor `{k}` lies outside the arena and would not be exchanged "
                                 "(allocate it with env.alloc_rollout(..., log_prob=True / value=True))")
        self.record, self.group = record, group
        self.world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
        self.nbytes = record.arena.numel()
        self._recv = None

    @property
    def bytes_received_per_rank(self) -> int:
        return (self.world - 1) * self.nbytes

    def gather(self) -> Dict[str, torch.Tensor]:
        from .env import RolloutRecord, record_fields
        rec = self.record
        if self.world == 1:
            recv = rec.arena.view(1, -1)
        else:
            if self._recv is None:
                self._recv = torch.empty(self.world * self.nbytes, dtype=torch.uint8, device=rec.arena.device)
            dist.all_gather_into_tensor(self._recv, rec.arena, group=self.group)   # rank r's arena lands at r * nbytes
            recv = self._recv.view(self.world, self.nbytes)
        out = RolloutRecord.views_of(recv, rec.layout, lead=(self.world,))
        A = rec["pos"].shape[-1]
        out.update(record_fields(out["record"], A))
        return out


def gather_trajectories(record: Dict[str, Optional[torch.Tensor]], group=None):
    """All ranks end up with the whole job's trajectories: ONE collective.

    Arena-backed records (`env.alloc_rollout`): zero-copy, returns views [world, T, B_local, ...] (see
    `TrajectoryExchange`; keep one exchange object alive to reuse its receive buffer across updates).
    Plain dicts (legacy, one extra copy): packed, gathered, returned as [T, world * B_local, ...] — NOTE the different
    layout (ranks concatenated on the env axis instead of a leading rank axis); prefer arena records."""
    if hasattr(record, "arena"):
        return TrajectoryExchange(record, group).gather()
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return {k: v for k, v in record.items() if v is not None and k != "record"}
    world = dist.get_world_size(group)
    buf, meta = pack_record(record)
    out = torch.empty(world * buf.numel(), dtype=torch.uint8, device=buf.device)
    dist.all_gather_into_tensor(out, buf, group=group)
    return unpack_record(out, meta, world)


def allreduce_gradients(module: torch.nn.Module, group=None):
    """The data-parallel alternative to gathering trajectories: every rank runs the PPO update on its own shard
    and the gradients are averaged with ONE flat all-reduce (parameters of the MAPPO networks at N=200, H=64:
    193 449 floats = 0.77 MB)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return 0
    grads = [p.grad for p in module.parameters() if p.grad is not None]
    if not grads:
        return 0
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, group=group)
    flat /= dist.get_world_size(group)
    off = 0
    for g in grads:
        n = g.numel()
        g.copy_(flat[off: off + n].view_as(g))
        off += n
    return flat.numel() * flat.element_size()


def allreduce_slab(grads: torch.Tensor, group=None):
    """`MappoUpdater(grad_sync=allreduce_slab)`: average the fused update's gradient slab [A + 1, S] across the ranks in
    place — ONE all-reduce of 0.6 MB per minibatch (N = 200, H = 64) instead of gathering trajectories.  The slab's loss
    words are averaged with it (each rank then reports the job's mean losses).  Every rank must run the same number of
    minibatches of the same size."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return grads
    dist.all_reduce(grads, group=group)
    grads /= dist.get_world_size(group)
    return grads


class RolloutCollector:
    """Collect T steps of B envs into device tensors.

    policy(obs: dict) -> (actions int[B, A], log_prob [B, A] or None, value [B] / [B, A] or None).

    `use_graph=True` (policy-driven collection on a GPU): the T-step loop — policy forward, `sy_env_step`,
    record copies, a few dozen small kernels per step — is captured once into a HIP graph and replayed, so
    a batched step costs one graph node chain instead of ~50 launches.  The first `collect()` runs eagerly
    (it doubles as the warm-up capture needs), the second captures and replays, later ones replay.  The
    policy must be capture-safe: fixed shapes, no host synchronisation, the default CUDA generator (or none).
    """

    def __init__(self, env, policy: Optional[Callable] = None, frames_per_batch: int = 64,
                 record_mask: bool = True, record_belief: bool = True, use_graph: bool = False):
        self.env, self.policy, self.T = env, policy, int(frames_per_batch)
        self.record_mask, self.record_belief = record_mask, record_belief
        # log-probabilities and values live in the same arena as the record: one buffer to exchange
        self._buf = env.alloc_rollout(self.T, record_mask, record_belief, log_prob=policy is not None,
                                      value=policy is not None)
        self._logp = self._buf.get("log_prob")
        self._value = None
        self.use_graph = bool(use_graph) and policy is not None
        self._graph = None
        self._calls = 0

    def _policy_loop(self):
        env, T, buf = self.env, self.T, self._buf
        for s in range(T):
            actions, logp, value = self.policy(env.observation())
            actions = actions.to(torch.int32).contiguous()
            if logp is not None:
                self._logp[s].copy_(logp)
            if value is not None:
                if self._value is None:      # central critic [B]: the arena's slot; per-agent values: own tensor
                    self._value = self._buf["value"] if tuple(value.shape) == (env.B,) else \
                        torch.zeros((T,) + tuple(value.shape), dtype=torch.float32, device=env.device)
                self._value[s].copy_(value)
            # one kernel: the transition plus row s of the record (observation before the step, action, outcome)
            env.step_record(actions, buf, s)

    @torch.no_grad()
    def collect(self, check: bool = True) -> Dict[str, torch.Tensor]:
        """One batch of T steps.  `check` (default): read the engine's status word afterwards (`env.check_status()`, one
        stream synchronisation) and raise if a launch of this batch lost an internal hand-off — its record must not reach
        an update.  Pass check=False to keep the call asynchronous and call `env.check_status()` yourself before using
        the data."""
        env, T, buf = self.env, self.T, self._buf
        if self.policy is None:
            env.rollout(T, out=buf, record=True)
            if check:
                env.check_status()
            return buf
        self._calls += 1
        if not self.use_graph or self._calls == 1:
            self._policy_loop()
        else:
            if self._graph is None:
                torch.cuda.synchronize(env.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.device(env.device), torch.cuda.graph(g):
                    self._policy_loop()            # recorded, not executed
                self._graph = g
            self._graph.replay()
        if check and not torch.cuda.is_current_stream_capturing():
            env.check_status()
        out = buf if self._value is None or self._value is buf.get("value") else dict(buf)
        if self._value is not None:
            out["value"] = self._value         # (per-agent values live outside the arena: `out` is then a plain dict and
        return out                             #  NOT exchangeable zero-copy — TrajectoryExchange refuses it)
