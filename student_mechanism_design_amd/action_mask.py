"""Device mirror of the reference's action-mask utilities (src/environment/action_mask.py).

Same names, argument meaning and result type as the reference; the mask itself is computed by
the HIP kernel behind `sy_action_mask_dense` (include/sy_env.h).  `compute_action_mask_batch`
is the batched form (Q queries against one set of dense matrices).
"""
import ctypes as C
from dataclasses import dataclass
from typing import Dict, List

import numpy as np
import torch

from . import _lib


@dataclass
class ActionMaskResult:
    """Same fields as action_mask.py:8-28."""
    mask: np.ndarray
    index_to_node: Dict[int, int]
    valid_actions: List[int]
    node_to_index: Dict[int, int]

    @property
    def num_valid_actions(self) -> int:
        return len(self.valid_actions)


def _normalize_tolls(tolls, num_nodes: int):
    """action_mask.py:87-97: None -> no toll, scalar -> everywhere, 1-D -> per destination, 2-D as is."""
    if tolls is None:
        return None
    if np.isscalar(tolls):
        return np.full((num_nodes, num_nodes), float(tolls))
    t = np.asarray(tolls, dtype=float)
    if t.ndim == 1:
        return np.tile(t.reshape(1, -1), (num_nodes, 1))
    return t


def _dev(a, device):
    return None if a is None else torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).to(device)


def compute_action_mask_batch(adjacency, current_nodes, budgets, tolls=None, edge_weights=None, device="cuda"):
    """mask bool[Q,N] for Q (current_node, budget) queries; matrices may be numpy or torch."""
    lib = _lib.load()
    if not torch.cuda.is_available():
        raise _lib.EngineError("compute_action_mask needs a GPU; there is no CPU fallback")
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.EngineError("compute_action_mask needs a cuda (ROCm) device")
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    adj = adjacency if isinstance(adjacency, torch.Tensor) else _dev(adjacency, device)
    adj = adj.to(device=device, dtype=torch.float64).contiguous()
    n = adj.shape[0]
    w = None if edge_weights is None else (edge_weights if isinstance(edge_weights, torch.Tensor)
                                           else _dev(edge_weights, device)).to(device=device, dtype=torch.float64).contiguous()
    tl = _normalize_tolls(tolls, n) if not isinstance(tolls, torch.Tensor) else tolls
    tl = None if tl is None else (tl if isinstance(tl, torch.Tensor) else _dev(tl, device)).to(
        device=device, dtype=torch.float64).contiguous()
    cur = torch.as_tensor(np.asarray(current_nodes, dtype=np.int32).reshape(-1)).to(device)
    bud = torch.as_tensor(np.asarray(budgets, dtype=np.float64).reshape(-1)).to(device)
    q = cur.shape[0]
    mask = torch.empty((q, n), dtype=torch.uint8, device=device)
    p = lambda t: None if t is None else C.c_void_p(t.data_ptr())  # noqa: E731
    with torch.cuda.device(device):               # launch on the tensors' device, with that device's stream
        stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
        _lib.check(lib.sy_action_mask_dense(p(adj), p(w), p(tl), n, p(cur), p(bud), q, p(mask), stream),
                   "sy_action_mask_dense")
    return mask.view(torch.bool)


def compute_action_mask(adjacency, current_node, budget, tolls=None, edge_weights=None, device="cuda") -> ActionMaskResult:
    """action_mask.py:30-84, fixed identity index<->node mapping."""
    m = compute_action_mask_batch(adjacency, [current_node], [budget], tolls=tolls, edge_weights=edge_weights,
                                  device=device)[0].cpu().numpy()
    n = m.shape[0]
    ident = {i: i for i in range(n)}
    return ActionMaskResult(mask=m, index_to_node=ident, valid_actions=[int(i) for i in np.nonzero(m)[0]],
                            node_to_index=dict(ident))


def get_action_mask_for_agent(adjacency, edge_weights, agent_position, agent_budget, tolls=None, device="cuda"):
    """action_mask.py:115-143."""
    return compute_action_mask(adjacency=adjacency, current_node=agent_position, budget=agent_budget, tolls=tolls,
                               edge_weights=edge_weights, device=device)
