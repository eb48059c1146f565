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
* `gather_trajectories` — the ONE exchange of the multi-GPU path: the local record is packed into a
  single byte buffer and all-gathered once (RCCL on GPU, gloo in the CPU tests); rollouts themselves
  need no communication because episodes are independent.
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


def pack_record(record: Dict[str, Optional[torch.Tensor]]):
    """Flatten a rollout record {name: [T, B, ...]} into one uint8 buffer [B_total_bytes] per env-major
    layout, plus the metadata needed to unpack it."""
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
        out[k] = torch.cat(pieces, dim=1)
        off += nbytes
    return out


def gather_trajectories(record: Dict[str, Optional[torch.Tensor]], group=None):
    """All ranks end up with the whole job's trajectories [T, world*B, ...]: ONE collective."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return {k: v for k, v in record.items() if v is not None and k != "record"}
    world = dist.get_world_size(group)
    buf, meta = pack_record(record)
    out = torch.empty(world * buf.numel(), dtype=torch.uint8, device=buf.device)
    dist.all_gather_into_tensor(out, buf, group=group)
    return unpack_record(out, meta, world)


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
        self._buf = env.alloc_rollout(self.T, record_mask, record_belief)
        B, A, dev = env.B, env.A, env.device
        self._logp = torch.zeros((self.T, B, A), dtype=torch.float32, device=dev)
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
                if self._value is None:
                    self._value = torch.zeros((T,) + tuple(value.shape), dtype=torch.float32, device=env.device)
                self._value[s].copy_(value)
            # one kernel: the transition plus row s of the record (observation before the step, action, outcome)
            env.step_record(actions, buf, s)

    @torch.no_grad()
    def collect(self) -> Dict[str, torch.Tensor]:
        env, T, buf = self.env, self.T, self._buf
        if self.policy is None:
            env.rollout(T, out=buf, record=True)
            return {k: v for k, v in buf.items() if v is not None}
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
        out = {k: v for k, v in buf.items() if v is not None}
        out["log_prob"] = self._logp
        if self._value is not None:
            out["value"] = self._value
        return out
