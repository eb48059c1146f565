"""Batched Scotland-Yard environment on one MI355X (host side of the C ABI in include/sy_env.h).

`BatchedScotlandYardEnv` owns the device tensors (torch is only the allocator / stream provider)
and forwards reset / step / rollout to the HIP engine.  Semantics per env are those of the
reference's CustomEnvironment (src/environment/yard.py:80-269); names follow the reference
(`MrX`, `Police{k}`, `agents_money`, `action_mask`, `Currency`, ...).  There is no CPU fallback:
without the HIP library or a GPU the constructor raises.
"""
import ctypes as C
from typing import Dict, Optional, Sequence, Union

import numpy as np
import torch

from . import _lib
from .graph import Board, DevicePool, PackedPool, pack_pool, reward_tables

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
                 device: Union[str, torch.device] = "cuda", seed: int = 0):
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
        cfg = _lib.EnvConfig(self.B, self.N, self.P, self.agent_money, self.max_timestep, self.G, self.NS,
                             int(reveal_interval), int(bool(police_evidence)), int(bool(belief_init_onehot)),
                             int(bool(auto_reset)), int(waves_per_block), int(env_id_offset))
        self._handle = C.c_void_p()
        self._policy = None
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
    def reset(self, seed: Optional[int] = None, env_mask: Optional[torch.Tensor] = None):
        """CustomEnvironment.reset for all envs (or those selected by env_mask bool/uint8[B])."""
        if seed is not None:
            self.seed = int(seed)
        sel = None
        if env_mask is not None:
            sel = env_mask.to(device=self.device, dtype=torch.uint8).contiguous()
        with torch.cuda.device(self.device):   # launches go to the HIP current device
            _lib.check(self.lib.sy_env_reset(self._handle, _ptr(sel), C.c_uint64(self.seed & (2**64 - 1)),
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

    def step_record(self, actions: torch.Tensor, out: Dict[str, torch.Tensor], s: int):
        """`step` that also fills row `s` of a rollout record from `alloc_rollout` (observation before the
        step, packed outcome row) inside the same kernel: no copy launches in a policy-driven collector."""
        act = actions
        if not (isinstance(act, torch.Tensor) and act.dtype == torch.int32 and act.is_contiguous()
                and act.device == self.device):
            act = torch.as_tensor(actions).to(device=self.device, dtype=torch.int32).contiguous()
        if tuple(act.shape) != (self.B, self.A):
            raise ValueError(f"actions must have shape ({self.B}, {self.A})")
        if not 0 <= s < out["record"].shape[0]:
            raise IndexError("record row out of range")
        row = _lib.RolloutBuffers(out["record"][s].data_ptr(),
                                  out["mask"][s].data_ptr() if out.get("mask") is not None else None,
                                  out["belief"][s].data_ptr() if out.get("belief") is not None else None, None)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.sy_env_step_record(self._handle, _ptr(act), C.byref(row), _stream_handle(self.device)),
                       "sy_env_step_record")
        return self.observation(), self.reward, self.terminated, self.truncated

    def alloc_rollout(self, T: int, record_mask=True, record_belief=True) -> Dict[str, torch.Tensor]:
        """Device buffers of one rollout.  `record` is the packed [T, B, RW] int32 tensor the engine
        writes with one store per env-step (include/sy_env.h); the named entries are views of it."""
        B, A, NS, dev = self.B, self.A, self.NS, self.device
        RW = int(self.lib.sy_record_words(A))
        rec = torch.zeros((T, B, RW), dtype=torch.int32, device=dev)
        buf = {
            "record": rec,
            "reward": rec[..., : 2 * A].view(torch.float64),
            "pos": rec[..., 2 * A: 3 * A], "budget": rec[..., 3 * A: 4 * A], "action": rec[..., 4 * A: 5 * A],
            "t": rec[..., 5 * A], "terminated": rec[..., 5 * A + 1], "truncated": rec[..., 5 * A + 2],
            "winner": rec[..., 5 * A + 3],
            "mask": torch.empty((T, B, A, NS), dtype=torch.uint8, device=dev) if record_mask else None,
            "belief": torch.empty((T, B, NS), dtype=torch.float32, device=dev)
            if (record_belief and self._belief is not None) else None,
        }
        return buf

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
        rb = None
        if record:
            if self._policy is not None and out.get("log_prob") is None:
                out["log_prob"] = torch.zeros((int(T), self.B, self.A), dtype=torch.float32, device=self.device)
            rb = _lib.RolloutBuffers(*[out[k].data_ptr() if out.get(k) is not None else None
                                       for k in ("record", "mask", "belief", "log_prob")])
        with torch.cuda.device(self.device):
            _lib.check(self.lib.sy_env_rollout(self._handle, int(T), C.byref(rb) if rb is not None else None,
                                               _stream_handle(self.device)), "sy_env_rollout")
        return out
