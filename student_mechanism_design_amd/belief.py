"""Device belief tracker: the deterministic forward filter that the reference's stochastic
ParticleBeliefTracker (src/environment/belief_module.py:41-111) estimates by Monte-Carlo.

Same method names / argument meaning (`reset`, `update(adjacency, observation_hint, reveal)`),
batched over Q independent beliefs on one board.  Computation: HIP kernel behind `sy_belief_update`.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import ELL_WIDTH
from .graph import PAD_WEIGHT, node_stride_for


class DeviceBeliefTracker:
    def __init__(self, num_nodes: int, adjacency, num_beliefs: int = 1, device="cuda"):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.EngineError("DeviceBeliefTracker needs a GPU; there is no CPU fallback")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.EngineError("DeviceBeliefTracker needs a cuda (ROCm) device")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.num_nodes = int(num_nodes)
        self.NS = node_stride_for(num_nodes)
        self.Q = int(num_beliefs)
        self.set_adjacency(adjacency)
        self._belief = torch.zeros((self.Q, self.NS), dtype=torch.float32, device=self.device)
        self.reset()

    def set_adjacency(self, adjacency):
        """Rows are read exactly as the reference reads them (belief_module.py:88-96: a particle on node i moves to
        a uniform element of nonzero(adjacency[i])): directed rows are kept directed, a non-zero diagonal entry is
        a self-neighbour ("staying put"), an all-zero row keeps its particles.  The filter gathers over the
        IN-neighbours of every node with weights 1 / out-degree: ell[j] lists {i : adjacency[i, j] != 0},
        inv_deg[i] = 1 / |nonzero(adjacency[i])|.  At most 16 in-neighbours per node (ELL width)."""
        adj = np.asarray(adjacency) != 0
        n = self.num_nodes
        if adj.shape != (n, n):
            raise ValueError(f"adjacency must be [{n}, {n}]")
        indeg = adj.sum(axis=0)
        if indeg.max(initial=0) > ELL_WIDTH:
            raise ValueError(f"a node has {int(indeg.max())} in-neighbours; the engine's ELL width is {ELL_WIDTH}")
        ell = np.full((n, ELL_WIDTH), (PAD_WEIGHT << 16) | n, dtype=np.uint32)
        for j in range(n):
            src = np.nonzero(adj[:, j])[0]
            ell[j, : src.shape[0]] = (np.uint32(1) << 16) | src.astype(np.uint32)
        outdeg = adj.sum(axis=1)
        inv = np.zeros(self.NS, dtype=np.float32)
        inv[:n][outdeg > 0] = (1.0 / outdeg[outdeg > 0]).astype(np.float32)
        self.ell = torch.from_numpy(ell.view(np.int32).copy()).to(self.device)
        self.inv_deg = torch.from_numpy(inv).to(self.device)

    def reset(self, mr_x_position=None):
        """belief_module.py:57-67: uniform prior, or a delta when the position is known."""
        self._belief.zero_()
        if mr_x_position is None:
            self._belief[:, : self.num_nodes] = 1.0 / self.num_nodes
        else:
            pos = torch.as_tensor(mr_x_position, device=self.device).reshape(-1).expand(self.Q).long()
            self._belief[torch.arange(self.Q, device=self.device), pos] = 1.0
        return self.distribution()

    def distribution(self):
        return self._belief[:, : self.num_nodes]

    def update(self, adjacency=None, observation_hint=None, reveal=None):
        """belief_module.py:69-111.  observation_hint: list of nodes (shared) or list of lists (per
        belief); reveal: node id (shared) or int[Q] with -1 = no reveal."""
        if adjacency is not None:
            self.set_adjacency(adjacency)
        hint_t, hw = None, 0
        if observation_hint is not None and len(observation_hint) > 0:
            rows = observation_hint if isinstance(observation_hint[0], (list, tuple, np.ndarray)) else [observation_hint] * self.Q
            hw = max(1, max(len(r) for r in rows))
            h = np.full((self.Q, hw), -1, dtype=np.int32)
            for i, r in enumerate(rows):
                h[i, : len(r)] = np.asarray(r, dtype=np.int32)
            hint_t = torch.from_numpy(h).to(self.device)
        rev_t = None
        if reveal is not None:
            r = np.asarray(reveal, dtype=np.int32).reshape(-1)
            rev_t = torch.from_numpy(np.broadcast_to(r, (self.Q,)).copy()).to(self.device)
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
        with torch.cuda.device(self.device):      # the launch goes to the HIP current device: make it the tensors' device
            stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            _lib.check(self.lib.sy_belief_update(p(self.ell), p(self.inv_deg), self.num_nodes, self.NS, p(self._belief),
                                                 p(hint_t), hw, p(rev_t), self.Q, stream), "sy_belief_update")
        return self.distribution()
