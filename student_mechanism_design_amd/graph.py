"""Host-side board handling: sampler, ELL packing, all-pairs shortest paths, reward tables.

Reset-side work of the reference that is NOT a kernel:
  * `sample_board` follows ConnectedGraph.sample / _create_tree (graph_layout.py:9-80): random-Prim
    spanning tree, then shuffled extra edges under a degree cap of 4, weights uniform in {1..4}.
    It uses its own numpy Generator — parity is on GIVEN boards, not on Python's Mersenne stream.
  * `pack_ell` / `all_pairs_shortest_paths` build the device tables that replace the per-step
    edge-list scans (yard.py:420-472) and per-query Dijkstra (pathfinding.py:34-137).
"""
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from ._lib import ELL_WIDTH, MAX_NODES

MAX_WEIGHT = 5  # ConnectedGraph.MAX_WEIGHT, graph_layout.py:7 (weights are randint(1, 5) -> 1..4)
PAD_WEIGHT = 0xFFFF


@dataclass
class Board:
    """Same three arrays as the reference's GraphInstance (graph_layout.py:52)."""
    nodes: np.ndarray       # int64 [N]
    edges: np.ndarray       # int64 [E]   edge weights
    edge_links: np.ndarray  # int32 [E,2]

    @property
    def num_nodes(self):
        return int(self.nodes.shape[0])

    @property
    def num_edges(self):
        return int(self.edge_links.shape[0])


def make_board(num_nodes, edge_links, edge_weights) -> Board:
    links = np.asarray(edge_links, dtype=np.int32).reshape(-1, 2)
    w = np.asarray(edge_weights, dtype=np.int64).reshape(-1)
    if links.shape[0] != w.shape[0]:
        raise ValueError("edge_links and edge_weights differ in length")
    if links.size and (links.min() < 0 or links.max() >= num_nodes):
        raise ValueError("edge endpoint out of range")
    if (links[:, 0] == links[:, 1]).any():
        raise ValueError("self loops are not supported")
    if (w < 0).any() or (w >= PAD_WEIGHT).any():
        raise ValueError("edge weights must be in [0, 65534]")
    return Board(np.arange(num_nodes, dtype=np.int64), w, links)


def sample_board(num_nodes=10, num_edges=None, max_edges_per_node=4, rng=None) -> Board:
    """graph_layout.py:9-52 with a numpy Generator."""
    rng = rng if rng is not None else np.random.default_rng()
    n = int(num_nodes)
    # _create_tree (graph_layout.py:54-80): an edge drawn uniformly from visited x unvisited is a
    # uniform visited endpoint and, independently, a uniform unvisited endpoint.
    order = rng.permutation(n)
    visited = [int(order[0])]
    unvisited = [int(x) for x in order[1:]]
    links = []
    while unvisited:
        u = visited[int(rng.integers(0, len(visited)))]
        k = int(rng.integers(0, len(unvisited)))
        v = unvisited[k]
        unvisited[k] = unvisited[-1]
        unvisited.pop()
        links.append((u, v))
        visited.append(v)
    if num_edges is None:
        num_edges = n - 1
    extra = int(num_edges) - len(links)
    if extra > 0:
        deg = np.zeros(n, dtype=np.int64)
        present = np.zeros((n, n), dtype=bool)
        for u, v in links:
            deg[u] += 1
            deg[v] += 1
            present[u, v] = present[v, u] = True
        iu, ju = np.triu_indices(n, k=1)
        keep = ~present[iu, ju]
        cand = np.stack([iu[keep], ju[keep]], axis=1)
        cand = cand[rng.permutation(cand.shape[0])]
        for i, j in cand:
            if extra <= 0:
                break
            if deg[i] < max_edges_per_node and deg[j] < max_edges_per_node:
                links.append((int(i), int(j)))
                deg[i] += 1
                deg[j] += 1
                extra -= 1
    weights = rng.integers(1, MAX_WEIGHT, size=len(links))
    return make_board(n, np.array(links, dtype=np.int32).reshape(-1, 2), weights)


def sample_board_pool(num_graphs, num_nodes, num_edges, seed=0, max_attempts=100) -> List[Board]:
    """Boards with one common edge count, as CustomEnvironment does (yard.py:65-101): the first
    sample fixes the achievable count, later samples are redrawn until they match."""
    rng = np.random.default_rng(seed)
    first = sample_board(num_nodes, num_edges, rng=rng)
    boards = [first]
    while len(boards) < num_graphs:
        for attempt in range(max_attempts):
            b = sample_board(num_nodes, num_edges, rng=rng)
            if b.num_edges == first.num_edges:
                boards.append(b)
                break
        else:
            raise RuntimeError(
                f"Failed to generate graph with {first.num_edges} edges after {max_attempts} attempts.")
    return boards


def min_weight_matrix(board: Board) -> np.ndarray:
    """Dense min edge weight (parallel edges collapse to the cheapest, yard.py:460-465); -1 = no edge."""
    n = board.num_nodes
    w = np.full((n, n), -1, dtype=np.int64)
    for (u, v), c in zip(board.edge_links, board.edges):
        if w[u, v] < 0 or c < w[u, v]:
            w[u, v] = w[v, u] = c
    return w


def pack_ell(board: Board) -> np.ndarray:
    """uint32 [N][16]: neighbour id | (weight << 16), rows ascending by neighbour id; padding entries
    are N | 0xFFFF0000 (index N is a zero slot in the belief scratch vector)."""
    n = board.num_nodes
    if n > MAX_NODES:
        raise ValueError(f"at most {MAX_NODES} nodes")
    w = min_weight_matrix(board)
    ell = np.full((n, ELL_WIDTH), (PAD_WEIGHT << 16) | n, dtype=np.uint32)
    for u in range(n):
        nb = np.nonzero(w[u] >= 0)[0]
        if nb.shape[0] > ELL_WIDTH:
            raise ValueError(f"node {u} has {nb.shape[0]} neighbours; the engine's ELL width is {ELL_WIDTH}")
        ell[u, : nb.shape[0]] = (w[u, nb].astype(np.uint32) << 16) | nb.astype(np.uint32)
    return ell


def all_pairs_shortest_paths(board: Board) -> np.ndarray:
    """Weighted APSP (Floyd-Warshall on exact integers) == the reference's Dijkstra distances."""
    n = board.num_nodes
    w = min_weight_matrix(board)
    inf = np.int64(1) << 40
    d = np.where(w >= 0, w, inf)
    np.fill_diagonal(d, 0)
    for k in range(n):
        np.minimum(d, d[:, k:k + 1] + d[k:k + 1, :], out=d)
    if (d >= inf).any():
        raise ValueError("board is not connected (the reference's sampler always builds a spanning tree)")
    if d.max() >= 0xFFFF:
        raise ValueError("shortest-path length does not fit uint16")
    return d.astype(np.uint16)


def inverse_degree(board: Board, node_stride: int) -> np.ndarray:
    w = min_weight_matrix(board)
    deg = (w >= 0).sum(axis=1)
    out = np.zeros(node_stride, dtype=np.float32)
    nz = deg > 0
    out[: board.num_nodes][nz] = (1.0 / deg[nz]).astype(np.float32)
    return out


def reward_tables(n_exp=1024, n_cov=512):
    """exp(-d) and exp(-log1p(v)) exactly as the reference evaluates them with numpy
    (reward_calculator.py:184-207), tabulated over the integer arguments that can occur."""
    exp_tab = np.exp(-np.arange(n_exp, dtype=np.float64))
    cov_tab = np.exp(-np.log1p(np.arange(n_cov, dtype=np.float64)))
    return exp_tab, cov_tab


def node_stride_for(num_nodes: int) -> int:
    return (int(num_nodes) + 15) // 16 * 16


@dataclass
class PackedPool:
    """Host copies of the device graph pool (include/sy_env.h layouts)."""
    boards: List[Board]
    num_nodes: int
    node_stride: int
    ell: np.ndarray      # uint32 [G][N][16]
    apsp: np.ndarray     # uint16 [G][N][N]
    inv_deg: np.ndarray  # float32 [G][NS]


def device_all_pairs_shortest_paths(ell, num_nodes: int, device="cuda"):
    """APSP of a whole pool on the GPU (`sy_build_apsp`): ell uint32 [G, N, 16] (numpy or int32 torch
    tensor) -> uint16-valued torch int16 tensor [G, N, N] on `device`.  No CPU fallback."""
    import ctypes as C
    import torch
    from . import _lib
    lib = _lib.load()
    if not torch.cuda.is_available():
        raise _lib.EngineError("device_all_pairs_shortest_paths needs a GPU; use all_pairs_shortest_paths on the host")
    dev = torch.device(device)
    ell_t = ell if isinstance(ell, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(ell).view(np.int32).copy())
    ell_t = ell_t.to(dev).contiguous()
    G = ell_t.shape[0]
    out = torch.empty((G, num_nodes, num_nodes), dtype=torch.int16, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(lib.sy_build_apsp(C.c_void_p(ell_t.data_ptr()), int(num_nodes), int(G), C.c_void_p(out.data_ptr()), stream),
               "sy_build_apsp")
    return out


def pack_pool(boards: Sequence[Board], node_stride: Optional[int] = None) -> PackedPool:
    boards = list(boards)
    n = boards[0].num_nodes
    if any(b.num_nodes != n for b in boards):
        raise ValueError("all boards of a pool must have the same node count")
    ns = node_stride or node_stride_for(n)
    return PackedPool(boards, n, ns,
                      np.stack([pack_ell(b) for b in boards]),
                      np.stack([all_pairs_shortest_paths(b) for b in boards]),
                      np.stack([inverse_degree(b, ns) for b in boards]))
