"""Batched Scotland-Yard environment on one MI355X (host side of the C ABI in include/sy_env.h).

`BatchedScotlandYardEnv` owns the device tensors (torch is only the allocator / stream provider)
and forwards reset / step / rollout to the HIP engine.  Semantics per env are those of the
reference's CustomEnvironment (src/environment/yard.py:80-269); names follow the reference
(`MrX`, `Police{k}`, `agents_money`, `action_mask`, `Currency`, ...).  There is no CPU fallback:
without the HIP library or a GPU the constructor raises.
"""
import ctypes as C
import os
from typing import Dict, Optional, Sequence, Union

import numpy as np
import torch

from . import _lib
from .graph import Board, DevicePool, PackedPool, belief_lanes, pack_pool, reward_tables
from .graph import belief_layout as belief_layout_of

# order of REWARD_WEIGHT_NAMES, src/reward_net.py:5-17
REWARD_WEIGHT_NAMES = [
    "Police_distance", "Police_group", "Police_position", "Police_time",
    "Mrx_closest", "Mrx_average", "Mrx_position", "Mrx_time",
    "Police_coverage", "Police_proximity", "Police_overlap_penalty",
]
WINNER_NAMES = {0: None, 1: "Police", 2: "MrX"}
DEFAULT_ACTION = -1  # CustomEnvironment.DEFAULT_ACTION, yard.py:16


def weights_to_array(reward_weights) -> np.ndarray:
    """dict (reference style, values may be Python floats or 0-d torch tensors) or 11-sequence -> float64[11]."""
    if isinstance(reward_weights, dict):
        return np.array([float(reward_weights[k]) for k in REWARD_WEIGHT_NAMES], dtype=np.float64)
    w = np.asarray(reward_weights, dtype=np.float64).reshape(-1)
    if w.shape[0] != _lib.NUM_WEIGHTS:
        raise ValueError(f"expected {_lib.NUM_WEIGHTS} reward weights")
    return w


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream_handle(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _splitmix64(x: int) -> int:
    """One round of SplitMix64 (Steele et al. 2014): decorrelates consecutive reset epochs of one seed."""
    m = 2**64 - 1
    x = (x + 0x9E3779B97F4A7C15) & m
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & m
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & m
    return x ^ (x >> 31)


class RolloutRecord(dict):
    """Rollout buffers {name: tensor}; every tensor is a view of `arena` (one contiguous uint8 device buffer).
    `layout` = [(name, dtype, shape, byte offset, nbytes)] of the primary parts; `record_fields(rec)` gives the
    named views of a packed record tensor."""

    def __init__(self, arena: torch.Tensor, layout):
        super().__init__()
        self.arena, self.layout = arena, list(layout)
        for name, dtype, shape, off, nbytes in self.layout:
            self[name] = arena[off: off + nbytes].view(dtype).view(shape)

    @staticmethod
    def views_of(arena: torch.Tensor, layout, lead=()):
        """The primary parts of an arena with extra leading dims `lead` (e.g. (world,) after an all-gather:
        arena [world, nbytes]) as views — no copies."""
        out = {}
        for name, dtype, shape, off, nbytes in layout:
            out[name] = arena[..., off: off + nbytes].view(dtype).view(tuple(lead) + tuple(shape))
        return out


def record_fields(rec: torch.Tensor, num_agents: int) -> Dict[str, torch.Tensor]:
    """Named views of a packed record tensor [..., RW] (layout: include/sy_env.h)."""
    A = int(num_agents)
    return {"reward": rec[..., : 2 * A].view(torch.float64), "pos": rec[..., 2 * A: 3 * A],
            "budget": rec[..., 3 * A: 4 * A], "action": rec[..., 4 * A: 5 * A], "t": rec[..., 5 * A],
            "terminated": rec[..., 5 * A + 1], "truncated": rec[..., 5 * A + 2], "winner": rec[..., 5 * A + 3]}


def record_words(num_agents: int) -> int:
    """dwords per packed record row (mirror of sy_record_words, include/sy_env.h): 5A + 4 rounded up to 4."""
    return (5 * int(num_agents) + 4 + 3) & ~3


def make_rollout_record(T, B, A, NS, RW, device, mask=True, belief=True, log_prob=False, value=False) -> RolloutRecord:
    """Carve the buffers of one rollout out of one contiguous byte arena (see `alloc_rollout`)."""
    T = int(T)
    parts = [("record", torch.int32, (T, B, RW))]
    if mask:
        parts.append(("mask", torch.uint8, (T, B, A, NS)))
    if belief:
        parts.append(("belief", torch.float32, (T, B, NS)))
    if log_prob:
        parts.append(("log_prob", torch.float32, (T, B, A)))
    if value:
        parts.append(("value", torch.float32, (T, B)))
    layout, off = [], 0
    for name, dtype, shape in parts:
        nbytes = int(np.prod(shape)) * torch.empty((), dtype=dtype).element_size()
        layout.append((name, dtype, shape, off, nbytes))
        off = (off + nbytes + 255) // 256 * 256
    buf = RolloutRecord(torch.zeros(off, dtype=torch.uint8, device=device), layout)
    buf.update(record_fields(buf["record"], A))
    for k in ("mask", "belief"):
        buf.setdefault(k, None)
    return buf


class BatchedScotlandYardEnv:
    """B independent episodes stepped by one kernel launch.

    Observation tensors (views of the live state, updated in place by every call):
      pos int32[B,A], budget int32[B,A], t int32[B], action_mask bool[B,A,N], belief float32[B,N],
      visits uint16[B,N]; step outputs: reward float64[B,A], terminated/truncated bool[B], winner int8[B].
    """

    def __init__(self, num_envs: int, boards: Union[PackedPool, Sequence[Board]], num_police: int,
                 agent_money: int, reward_weights, max_timestep: int = 250, reveal_interval: int = 0,
                 police_evidence: bool = False, belief_init_onehot: bool = False, auto_reset: bool = True,
                 with_belief: bool = True, env_graph=None, env_id_offset: int = 0, waves_per_block: int = 0,
                 device: Union[str, torch.device] = "cuda", seed: int = 0, belief_layout: bool = True):
        self.lib = _lib.load()  # raises if the HIP library is missing
        if not torch.cuda.is_available():
            raise _lib.EngineError("BatchedScotlandYardEnv needs a GPU (torch.cuda.is_available() is False); "
                                   "there is no CPU fallback")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.EngineError("device must be a cuda (ROCm) device")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._device_pool = boards if isinstance(boards, DevicePool) else None
        if self._device_pool is not None:
            self.pool = boards.to_packed()       # host mirror of the device-built pool (arrays only)
        else:
            self.pool = boards if isinstance(boards, PackedPool) else pack_pool(boards)
        self.B, self.P, self.A = int(num_envs), int(num_police), int(num_police) + 1
        self.N, self.NS, self.G = self.pool.num_nodes, self.pool.node_stride, len(self.pool.boards)
        self.agent_money, self.max_timestep = int(agent_money), int(max_timestep)
        self.possible_agents = ["MrX"] + [f"Police{k}" for k in range(self.P)]
        self.seed = int(seed)
        self.env_id_offset = int(env_id_offset)   # global index of env 0 (rank * B): names the Philox sub-streams
        self.reset_epoch = 0          # seedless full resets since the last explicit seed (see `reset`)
        cfg = _lib.EnvConfig(self.B, self.N, self.P, self.agent_money, self.max_timestep, self.G, self.NS,
                             int(reveal_interval), int(bool(police_evidence)), int(bool(belief_init_onehot)),
                             int(bool(auto_reset)), int(waves_per_block), int(env_id_offset))
        self._handle = C.c_void_p()
        self._policy = None
        self._checked_record = None
        _lib.check(self.lib.sy_env_create(C.byref(cfg), C.byref(self._handle)), "sy_env_create")
        wpb, blocks, lds = C.c_int32(), C.c_int32(), C.c_int32()
        _lib.check(self.lib.sy_env_launch_info(self._handle, C.byref(wpb), C.byref(blocks), C.byref(lds)))
        self.waves_per_block, self.launch_blocks, self.lds_bytes = wpb.value, blocks.value, lds.value

        dev = self.device
        # ---- graph pool (per-graph constants, resident in HBM / L2)
        if self._device_pool is not None:            # tables were built on this GPU: use them in place
            self.ell = self._device_pool.ell.to(dev).contiguous()
            self.apsp = self._device_pool.apsp.to(dev).contiguous()
            self.inv_deg = self._device_pool.inv_deg.to(dev).contiguous()
        else:
            self.ell = torch.from_numpy(self.pool.ell.view(np.int32).copy()).to(dev)
            self.apsp = torch.from_numpy(self.pool.apsp.view(np.int16).copy()).to(dev)
            self.inv_deg = torch.from_numpy(self.pool.inv_deg).to(dev)
        if env_graph is None:
            # contiguous slabs of envs per graph, aligned to launch blocks
            per = -(-self.B // self.G)
            per = -(-per // self.waves_per_block) * self.waves_per_block
            env_graph = np.minimum(np.arange(self.B) // per, self.G - 1)
        env_graph = np.asarray(env_graph, dtype=np.int32).reshape(-1)
        if env_graph.shape[0] != self.B or env_graph.min() < 0 or env_graph.max() >= self.G:
            raise ValueError("env_graph must be int[B] with values in [0, G)")
        first_of_block = env_graph[(np.arange(self.B) // self.waves_per_block) * self.waves_per_block]
        if (env_graph != first_of_block).any():
            raise ValueError(f"all envs of one launch block ({self.waves_per_block} consecutive envs) must share a graph")
        self.env_graph_host = env_graph
        self.env_graph = torch.from_numpy(env_graph).to(dev)
        self.max_degree = int(((self.pool.ell & 0xFFFF) < self.N).sum(axis=2).max())
        _lib.check(self.lib.sy_env_set_graph_pool(self._handle, _ptr(self.ell), _ptr(self.apsp), _ptr(self.inv_deg),
                                                  _ptr(self.env_graph), self.max_degree), "sy_env_set_graph_pool")
        # ---- the belief filter's bank-aware LDS layout (graph.belief_layout: per board, which scratch entry a node's value goes
        # to and in which order a node visits its neighbours, so that a half-wave's gathers hit 32 different bank pairs)
        self.belief_gather = self.belief_slot = None
        if with_belief and belief_layout and belief_lanes(self.N) > 0 and not os.environ.get("SY_NO_BELIEF_LAYOUT"):   # (env var: A/B knob of the tools)
            lay = [belief_layout_of(self.pool.ell[g], self.N, self.NS) for g in range(self.G)]
            if all(x is not None for x in lay):
                self.belief_slot = torch.from_numpy(np.stack([x[0] for x in lay]).view(np.int16).copy()).to(dev)
                self.belief_gather = torch.from_numpy(np.stack([x[1] for x in lay]).view(np.int16).copy()).to(dev)
                _lib.check(self.lib.sy_env_set_belief_layout(self._handle, _ptr(self.belief_gather), _ptr(self.belief_slot)),
                           "sy_env_set_belief_layout")
        # ---- reward weights + tables
        exp_tab, cov_tab = reward_tables()
        self.exp_tab = torch.from_numpy(exp_tab).to(dev)
        self.cov_tab = torch.from_numpy(cov_tab).to(dev)
        self.set_reward_weights(reward_weights)
        # ---- live state
        B, A, NS = self.B, self.A, self.NS
        self.pos = torch.zeros((B, A), dtype=torch.int32, device=dev)
        self.budget = torch.zeros((B, A), dtype=torch.int32, device=dev)
        self.t = torch.zeros((B,), dtype=torch.int32, device=dev)
        self.step_count = torch.zeros((B,), dtype=torch.int32, device=dev)
        self._visits = torch.zeros((B, NS), dtype=torch.int16, device=dev)
        self._belief = torch.zeros((B, NS), dtype=torch.float32, device=dev) if with_belief else None
        self._mask = torch.zeros((B, A, NS), dtype=torch.uint8, device=dev)
        self.reward = torch.zeros((B, A), dtype=torch.float64, device=dev)
        self._terminated = torch.zeros((B,), dtype=torch.uint8, device=dev)
        self._truncated = torch.zeros((B,), dtype=torch.uint8, device=dev)
        self.winner = torch.zeros((B,), dtype=torch.int8, device=dev)
        st = _lib.EnvState(*[t.data_ptr() if t is not None else None for t in (
            self.pos, self.budget, self.t, self.step_count, self._visits, self._belief, self._mask, self.reward,
            self._terminated, self._truncated, self.winner)])
        _lib.check(self.lib.sy_env_bind_state(self._handle, C.byref(st)), "sy_env_bind_state")
        # failure word: kernels OR a bit into it when a bounded in-kernel wait runs out (see `check_status`)
        self._status = torch.zeros((1,), dtype=torch.int32, device=dev)
        _lib.check(self.lib.sy_env_bind_status(self._handle, _ptr(self._status)), "sy_env_bind_status")
        self.reset(seed=self.seed)

    # ------------------------------------------------------------------ housekeeping
    def close(self):
        if getattr(self, "_handle", None):
            self.lib.sy_env_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def status(self) -> int:
        """The engine's device status word (SY_STATUS_* bits, include/sy_env.h).  Synchronises the current stream."""
        w = C.c_uint32(0)
        with torch.cuda.device(self.device):
            self.lib.sy_env_status(self._handle, _stream_handle(self.device), C.byref(w))
        return int(w.value)

    def check_status(self):
        """Raise `EngineError` if any launch since the last check reported an internal failure (a hand-off between
        the move and belief waves of the fused rollout that ran out of its bounded wait): such a launch drains, but
        its belief / record must not be trusted.  Call it at a host sync point (after a rollout batch, before an
        update); it synchronises the current stream.  The word is cleared so that later launches are judged afresh."""
        with torch.cuda.device(self.device):
            rc = self.lib.sy_env_status(self._handle, _stream_handle(self.device), None)
        if rc != 0:
            msg = self.lib.sy_last_error().decode("utf-8", "replace")
            self._status.zero_()
            raise _lib.EngineError(msg)

    def rollout_kernel_name(self, record: bool = True) -> str:
        """The kernel instance `rollout` launches for this env as configured now (what rocprofv3 reports): asked from
        the library (`sy_env_rollout_kernel_name`), which owns the selection rules."""
        buf = C.create_string_buffer(128)
        _lib.check(self.lib.sy_env_rollout_kernel_name(self._handle, 1 if record else 0, buf, len(buf)),
                   "sy_env_rollout_kernel_name")
        return buf.value.decode()

    def set_reward_weights(self, reward_weights):
        """The 11 weights of reward_calculator.py (RewardWeightNet output, reward_net.py:5-17)."""
        self.reward_weights = weights_to_array(reward_weights)
        w = (C.c_double * _lib.NUM_WEIGHTS)(*self.reward_weights.tolist())
        _lib.check(self.lib.sy_env_set_rewards(self._handle, w, _ptr(self.exp_tab), self.exp_tab.numel(),
                                               _ptr(self.cov_tab), self.cov_tab.numel()), "sy_env_set_rewards")

    # ------------------------------------------------------------------ observation views
    @property
    def action_mask(self) -> torch.Tensor:
        return self._mask[:, :, : self.N].view(torch.bool)

    @property
    def belief(self) -> Optional[torch.Tensor]:
        return None if self._belief is None else self._belief[:, : self.N]

    @property
    def visits(self) -> torch.Tensor:
        return self._visits[:, : self.N]

    @property
    def terminated(self) -> torch.Tensor:
        return self._terminated.view(torch.bool)

    @property
    def truncated(self) -> torch.Tensor:
        return self._truncated.view(torch.bool)

    @property
    def done(self) -> torch.Tensor:
        """Trainer rule: terminated["Police0"] or all(truncated) (training/utils.py:241-251)."""
        return (self._terminated | self._truncated).view(torch.bool)

    def observation(self) -> Dict[str, torch.Tensor]:
        """Batched equivalents of the per-agent observation dict (yard.py:319-332)."""
        return {
            "MrX_pos": self.pos[:, 0], "Polices_pos": self.pos[:, 1:], "Currency": self.budget[:, 1:],
            "agent_position": self.pos, "agent_budget": self.budget, "action_mask": self.action_mask,
            "belief_map": self.belief, "timestep": self.t,
        }

    # ------------------------------------------------------------------ engine calls
    @property
    def stream_key(self) -> int:
        """The Philox key the engine currently runs with: the seed itself after an explicit `reset(seed=...)`,
        a SplitMix64 mix of (seed, reset epoch) after seedless full resets (what a checker must be keyed with)."""
        if self.reset_epoch == 0:
            return self.seed & (2**64 - 1)
        return _splitmix64((self.seed + 0xD1B54A32D192ED03 * self.reset_epoch) & (2**64 - 1))

    def reset(self, seed: Optional[int] = None, env_mask: Optional[torch.Tensor] = None):
        """CustomEnvironment.reset for all envs (or those selected by env_mask bool/uint8[B]).

        RNG rule (engine-defined; the reference draws from numpy's global stream, yard.py:112-116):
          * `reset(seed=s)`      restarts every stream: key = s, step counters zeroed — reproducible episodes.
          * `reset()`            a NEW set of episodes, like the reference's reset: the reset epoch is mixed into the
                                 key (`stream_key`), so start nodes and action draws differ from every earlier reset.
          * `reset(env_mask=m)`  the selected envs restart from their own running counters (fresh starts); the key and
                                 the other envs' streams are untouched.  A new seed cannot be combined with a mask: it
                                 would re-key the envs that were not selected."""
        if seed is not None and env_mask is not None:
            raise ValueError("reset(seed=..., env_mask=...) would re-key the streams of the envs that are not selected; "
                             "reseed with a full reset")
        if seed is not None:
            self.seed = int(seed)
            self.reset_epoch = 0
        elif env_mask is None:
            self.reset_epoch += 1
        sel = None
        if env_mask is not None:
            sel = env_mask.to(device=self.device, dtype=torch.uint8).contiguous()
            if tuple(sel.shape) != (self.B,):
                raise ValueError(f"env_mask must have shape ({self.B},)")
        with torch.cuda.device(self.device):   # launches go to the HIP current device
            _lib.check(self.lib.sy_env_reset(self._handle, _ptr(sel), C.c_uint64(self.stream_key),
                                             _stream_handle(self.device)), "sy_env_reset")
        return self.observation()

    def reset_to(self, starts):
        """Reset every env to caller-given start nodes int[B,A] (golden replays)."""
        st = torch.as_tensor(starts, dtype=torch.int32).to(self.device).contiguous()
        if tuple(st.shape) != (self.B, self.A):
            raise ValueError(f"starts must have shape ({self.B}, {self.A})")
        with torch.cuda.device(self.device):
            _lib.check(self.lib.sy_env_reset_to(self._handle, _ptr(st), _stream_handle(self.device)), "sy_env_reset_to")
        return self.observation()

    def step(self, actions: torch.Tensor):
        """CustomEnvironment.step for the batch: actions int[B,A] node ids, -1 = no-op (None)."""
        act = actions
        if not (isinstance(act, torch.Tensor) and act.dtype == torch.int32 and act.is_contiguous()
                and act.device == self.device):
            act = torch.as_tensor(actions).to(device=self.device, dtype=torch.int32).contiguous()
        if tuple(act.shape) != (self.B, self.A):
            raise ValueError(f"actions must have shape ({self.B}, {self.A})")
        with torch.cuda.device(self.device):
            _lib.check(self.lib.sy_env_step(self._handle, _ptr(act), _stream_handle(self.device)), "sy_env_step")
        return self.observation(), self.reward, self.terminated, self.truncated

    def _check_rollout_buffers(self, out: Dict[str, torch.Tensor], T: int, need_log_prob: bool = False):
        """The C ABI receives raw pointers: a buffer that is too short, of another dtype / device or not contiguous
        would be written past its end.  Every buffer must hold at least T rows of the engine's row shape."""
        B, A, NS = self.B, self.A, self.NS
        RW = record_words(A)
        want = {"record": ((B, RW), torch.int32), "mask": ((B, A, NS), torch.uint8), "belief": ((B, NS), torch.float32),
                "log_prob": ((B, A), torch.float32)}
        if not isinstance(out, dict) or out.get("record") is None:
            raise ValueError("a rollout record needs the packed `record` tensor (use alloc_rollout)")
        if need_log_prob and out.get("log_prob") is None:
            raise ValueError("a policy rollout needs a `log_prob` buffer")
        for k, (shape, dtype) in want.items():
            v = out.get(k)
            if v is None:
                continue
            if not isinstance(v, torch.Tensor) or v.dtype != dtype:
                raise ValueError(f"rollout buffer `{k}` must be a {dtype} tensor")
            if v.device != self.device:
                raise ValueError(f"rollout buffer `{k}` lives on {v.device}, the env on {self.device}")
            if v.dim() != len(shape) + 1 or tuple(v.shape[1:]) != shape or v.shape[0] < T:
                raise ValueError(f"rollout buffer `{k}` must have shape (>= {T}, {', '.join(map(str, shape))}), "
                                 f"got {tuple(v.shape)}")
            if not v.is_contiguous():
                raise ValueError(f"rollout buffer `{k}` must be contiguous")
        if out.get("belief") is not None and self._belief is None:
            raise ValueError("this env tracks no belief (with_belief=False): pass belief=None")

    def step_record(self, actions: torch.Tensor, out: Dict[str, torch.Tensor], s: int):
        """`step` that also fills row `s` of a rollout record from `alloc_rollout` (observation before the
        step, packed outcome row) inside the same kernel: no copy launches in a policy-driven collector."""
        act = actions
        if not (isinstance(act, torch.Tensor) and act.dtype == torch.int32 and act.is_contiguous()
                and act.device == self.device):
            act = torch.as_tensor(actions).to(device=self.device, dtype=torch.int32).contiguous()
        if tuple(act.shape) != (self.B, self.A):
            raise ValueError(f"actions must have shape ({self.B}, {self.A})")
        key = (id(out), out["record"].data_ptr() if isinstance(out, dict) and out.get("record") is not None else 0)
        if self._checked_record != key:          # the same record row after row (a collector loop): validated once
            self._check_rollout_buffers(out, 1)
            self._checked_record = key
        if not 0 <= s < out["record"].shape[0]:
            raise IndexError("record row out of range")
        for k in ("mask", "belief"):
            if out.get(k) is not None and out[k].shape[0] <= s:
                raise IndexError(f"`{k}` has no row {s}")
        row = _lib.RolloutBuffers(out["record"][s].data_ptr(),
                                  out["mask"][s].data_ptr() if out.get("mask") is not None else None,
                                  out["belief"][s].data_ptr() if out.get("belief") is not None else None, None)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.sy_env_step_record(self._handle, _ptr(act), C.byref(row), _stream_handle(self.device)),
                       "sy_env_step_record")
        return self.observation(), self.reward, self.terminated, self.truncated

    def alloc_rollout(self, T: int, record_mask=True, record_belief=True, log_prob: Optional[bool] = None,
                      value: bool = False) -> "RolloutRecord":
        """Device buffers of one rollout, carved out of ONE contiguous byte arena (`.arena`, uint8) so that the
        multi-GPU exchange (`collector.gather_trajectories`) is a single collective on memory that already is the
        send buffer — nothing is packed or copied.  `record` is the packed [T, B, RW] int32 tensor the engine
        writes with one store per env-step (include/sy_env.h); `reward` ... `winner` are views of it; `mask`
        [T, B, A, NS], `belief` [T, B, NS], `log_prob` [T, B, A] (default: when an in-kernel policy is set) and
        `value` [T, B] (for policy-driven collectors) follow, each 256-byte aligned."""
        if log_prob is None:
            log_prob = self._policy is not None
        return make_rollout_record(T, self.B, self.A, self.NS, int(self.lib.sy_record_words(self.A)), self.device,
                                   mask=record_mask, belief=record_belief and self._belief is not None,
                                   log_prob=log_prob, value=value)

    def set_policy(self, policy=None):
        """The reference's rollout loop with its own policy in it (mappo_trainer.py:161-287): after this call
        `rollout` samples every action from the MAPPO actors inside the fused kernel and records the
        log-probabilities (`log_prob` [T, B, A]).  `policy`: a `policies.DeviceMappoPolicy` (its packed weights are
        used; call its `refresh()` after optimiser steps, no need to call `set_policy` again) or None for the
        uniform-random policy."""
        if policy is None:
            _lib.check(self.lib.sy_env_set_policy(self._handle, None, 0), "sy_env_set_policy")
            self._policy = None
            return
        if policy.net.N != self.N or policy.net.P != self.P:
            raise ValueError("the policy was built for another number of nodes / police")
        _lib.check(self.lib.sy_env_set_policy(self._handle, C.byref(policy._w), int(policy.H)), "sy_env_set_policy")
        self._policy = policy     # keeps the packed weights alive

    def rollout(self, T: int, out: Optional[Dict[str, torch.Tensor]] = None, record: bool = True,
                record_mask=True, record_belief=True):
        """T fused steps in one launch with the in-kernel policy — uniform-random, or the MAPPO actors after
        `set_policy`; returns the record."""
        if record and out is None:
            out = self.alloc_rollout(T, record_mask, record_belief)
        T = int(T)
        if T < 1:
            raise ValueError("T must be >= 1")
        rb = None
        if record:
            if self._policy is not None and isinstance(out, dict) and out.get("log_prob") is None:
                if hasattr(out, "arena"):
                    # a tensor bolted on outside the arena would silently miss the multi-GPU exchange, which ships the arena
                    raise ValueError("this record was allocated without `log_prob` (alloc_rollout before set_policy?): "
                                     "allocate it again after set_policy, or pass log_prob=True")
                out["log_prob"] = torch.zeros((T, self.B, self.A), dtype=torch.float32, device=self.device)
            self._check_rollout_buffers(out, T, need_log_prob=self._policy is not None)
            rb = _lib.RolloutBuffers(*[out[k].data_ptr() if out.get(k) is not None else None
                                       for k in ("record", "mask", "belief", "log_prob")])
        with torch.cuda.device(self.device):
            _lib.check(self.lib.sy_env_rollout(self._handle, int(T), C.byref(rb) if rb is not None else None,
                                               _stream_handle(self.device)), "sy_env_rollout")
        return out
