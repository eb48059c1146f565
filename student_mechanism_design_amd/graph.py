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


def sample_board(num_nodes=10, num_edges=None, max_edges_per_node=4, rng=None, max_redraws=100) -> Board:
    """graph_layout.py:9-52 with a numpy Generator.  The random-Prim tree puts no cap on a node's degree
    (graph_layout.py:54-80); a board with a node of more than 16 neighbours cannot be packed (`pack_ell`), so
    such a draw is rejected and redrawn here (probability ~1e-9 per node at N=200) instead of failing later."""
    rng = rng if rng is not None else np.random.default_rng()
    for _ in range(max_redraws):
        b = _sample_board_once(num_nodes, num_edges, max_edges_per_node, rng)
        if max_degree(b) <= ELL_WIDTH:
            return b
    raise RuntimeError(f"board sampler: no board with max degree <= {ELL_WIDTH} in {max_redraws} draws")


def max_degree(board: Board) -> int:
    """Widest neighbour row (distinct neighbours; parallel edges collapse, yard.py:460-465)."""
    if board.num_edges == 0:
        return 0
    lo = np.minimum(board.edge_links[:, 0], board.edge_links[:, 1]).astype(np.int64)
    hi = np.maximum(board.edge_links[:, 0], board.edge_links[:, 1]).astype(np.int64)
    uniq = np.unique(lo * board.num_nodes + hi)
    deg = np.bincount(np.concatenate([uniq // board.num_nodes, uniq % board.num_nodes]), minlength=board.num_nodes)
    return int(deg.max())


def _sample_board_once(num_nodes, num_edges, max_edges_per_node, rng) -> Board:
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
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    ell_t = ell if isinstance(ell, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(ell).view(np.int32).copy())
    ell_t = ell_t.to(dev).contiguous()
    G = ell_t.shape[0]
    out = torch.empty((G, num_nodes, num_nodes), dtype=torch.int16, device=dev)
    with torch.cuda.device(dev):
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


class DevicePool:
    """A board pool that was sampled and packed ON the GPU (`sy_sample_boards` + `sy_build_apsp`).
    Device tensors: ell int32 [G,N,16] (uint32 bits), apsp int16 [G,N,N] (uint16 bits), inv_deg
    float32 [G,NS], edge_links int32 [G,E,2], edge_w int32 [G,E]; `to_packed()` gives host copies
    (incl. `Board` objects) for code that wants the reference's arrays."""

    def __init__(self, num_nodes, node_stride, ell, apsp, inv_deg, edge_links, edge_w, num_edges):
        self.num_nodes, self.node_stride = int(num_nodes), int(node_stride)
        self.ell, self.apsp, self.inv_deg = ell, apsp, inv_deg
        self.edge_links, self.edge_w, self.num_edges = edge_links, edge_w, int(num_edges)

    def __len__(self):
        return int(self.ell.shape[0])

    def to_packed(self) -> PackedPool:
        links = self.edge_links.cpu().numpy()
        w = self.edge_w.cpu().numpy()
        boards = [make_board(self.num_nodes, links[g, : self.num_edges], w[g, : self.num_edges]) for g in range(len(self))]
        return PackedPool(boards, self.num_nodes, self.node_stride,
                          self.ell.cpu().numpy().view(np.uint32), self.apsp.cpu().numpy().view(np.uint16),
                          self.inv_deg.cpu().numpy())


def sample_boards_device_raw(num_graphs, num_nodes, num_edges, seed=0, max_edges_per_node=4, device="cuda"):
    """One `sy_sample_boards` launch, no redraws: independent draws of ConnectedGraph.sample's process on the GPU.
    Returns host arrays (edge_links int32 [G, E, 2], edge_w int32 [G, E], edges_realised int32 [G]; -1 = a node
    exceeded the ELL width) — boards keep their own realised edge counts, as the reference's sampler does before
    `yard.py:90-101` filters them."""
    import ctypes as C
    import torch
    from . import _lib
    lib = _lib.load()
    if not torch.cuda.is_available():
        raise _lib.EngineError("sample_boards_device_raw needs a GPU; use sample_board on the host")
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    G, N = int(num_graphs), int(num_nodes)
    NS = node_stride_for(N)
    E = max(int(num_edges) if num_edges is not None else N - 1, N - 1)
    ell = torch.empty((G, N, ELL_WIDTH), dtype=torch.int32, device=dev)
    inv_deg = torch.empty((G, NS), dtype=torch.float32, device=dev)
    links = torch.zeros((G, E, 2), dtype=torch.int32, device=dev)
    w = torch.zeros((G, E), dtype=torch.int32, device=dev)
    counts = torch.empty((G,), dtype=torch.int32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    with torch.cuda.device(dev):
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.sy_sample_boards(N, NS, E, int(max_edges_per_node), C.c_uint64(int(seed) & (2**64 - 1)), G, p(ell),
                                        p(inv_deg), p(links), p(w), p(counts), E, stream), "sy_sample_boards")
    return links.cpu().numpy(), w.cpu().numpy(), counts.cpu().numpy()


def sample_board_pool_device(num_graphs, num_nodes, num_edges, seed=0, max_edges_per_node=4, max_attempts=100,
                             device="cuda") -> DevicePool:
    """`sample_board_pool` on the GPU: boards with one common edge count (yard.py:65-101 — the first
    board fixes the achievable count, the others are redrawn until they match; RuntimeError after
    `max_attempts`).  No CPU fallback."""
    import ctypes as C
    import torch
    from . import _lib
    lib = _lib.load()
    if not torch.cuda.is_available():
        raise _lib.EngineError("sample_board_pool_device needs a GPU; use sample_board_pool on the host")
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    G, N = int(num_graphs), int(num_nodes)
    NS = node_stride_for(N)
    E = max(int(num_edges) if num_edges is not None else N - 1, N - 1)
    ell = torch.empty((G, N, ELL_WIDTH), dtype=torch.int32, device=dev)
    inv_deg = torch.empty((G, NS), dtype=torch.float32, device=dev)
    links = torch.zeros((G, E, 2), dtype=torch.int32, device=dev)
    w = torch.zeros((G, E), dtype=torch.int32, device=dev)
    counts = torch.empty((G,), dtype=torch.int32, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731

    def draw(n, s):
        e_, i_, l_, w_, c_ = (torch.empty((n,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
                              for t in (ell, inv_deg, links, w, counts))
        l_.zero_(); w_.zero_()
        with torch.cuda.device(dev):
            stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.sy_sample_boards(N, NS, E, int(max_edges_per_node), C.c_uint64(s & (2**64 - 1)), n, p(e_), p(i_),
                                            p(l_), p(w_), p(c_), E, stream), "sy_sample_boards")
        return e_, i_, l_, w_, c_

    e_, i_, l_, w_, c_ = draw(G, int(seed))
    ell.copy_(e_); inv_deg.copy_(i_); links.copy_(l_); w.copy_(w_); counts.copy_(c_)
    host_counts = counts.cpu().numpy().copy()
    valid = host_counts[host_counts >= 0]
    if valid.size == 0:
        raise RuntimeError("board sampler: every board exceeded the ELL width")
    target = int(host_counts[np.argmax(host_counts >= 0)])    # the first valid board fixes the count
    for attempt in range(1, max_attempts + 1):
        bad = np.nonzero(host_counts != target)[0]
        if bad.size == 0:
            break
        e_, i_, l_, w_, c_ = draw(int(bad.size), int(seed) + 0x9E3779B97F4A7C15 * attempt)
        idx = torch.as_tensor(bad, device=dev, dtype=torch.long)
        ell[idx] = e_; inv_deg[idx] = i_; links[idx] = l_; w[idx] = w_
        host_counts[bad] = c_.cpu().numpy()
    else:
        raise RuntimeError(f"Failed to generate graph with {target} edges after {max_attempts} attempts.")
    apsp = device_all_pairs_shortest_paths(ell, N, device=dev)
    return DevicePool(N, NS, ell, apsp, inv_deg, links[:, :target].contiguous(), w[:, :target].contiguous(), target)


def gcn_tables(boards: Sequence[Board], directed: bool = True):
    """Message-passing tables of GCNConv's normalised propagation  A^ = D^-1/2 (A + I) D^-1/2  per board, as the GNN
    policy kernels read them (`sy_gnn_q_act`, `policies.GnnQModel`): for every TARGET node its source nodes and their
    coefficients 1 / sqrt(deg(source) deg(target)), deg = number of incoming edges + 1 (the self loop).

    directed=True is the reference's data flow: `create_graph_data` hands `board.edge_links.T` to the model
    (training/utils.py:170), i.e. every stored edge (u, v) ONCE, and GCNConv propagates source -> target along the
    edges as given — messages flow u -> v only.  directed=False adds both directions (what `to_undirected` would give).
    Returns (nbr int16 [G, N, 16] sources, -1 = padding; coef float32 [G, N, 16]; self_coef float32 [G, N])."""
    boards = list(boards)
    n = boards[0].num_nodes
    G = len(boards)
    nbr = np.full((G, n, ELL_WIDTH), -1, dtype=np.int16)
    coef = np.zeros((G, n, ELL_WIDTH), dtype=np.float32)
    self_coef = np.zeros((G, n), dtype=np.float32)
    for g, b in enumerate(boards):
        src = b.edge_links[:, 0].astype(np.int64)
        dst = b.edge_links[:, 1].astype(np.int64)
        if not directed:
            src, dst = np.concatenate([src, dst]), np.concatenate([dst, src])
        deg = 1.0 + np.bincount(dst, minlength=n).astype(np.float64)          # in-degree + self loop (gcn_norm)
        dinv = 1.0 / np.sqrt(deg)
        self_coef[g] = (dinv * dinv).astype(np.float32)
        fill = np.zeros(n, dtype=np.int64)
        for u, v in zip(src, dst):
            row = nbr[g, v, : fill[v]]
            hit = np.nonzero(row == u)[0]
            if hit.size:                                                        # a parallel edge: one more message
                coef[g, v, hit[0]] += np.float32(dinv[u] * dinv[v])
                continue
            if fill[v] >= ELL_WIDTH:
                raise ValueError(f"node {v} has more than {ELL_WIDTH} incoming edges")
            nbr[g, v, fill[v]] = u
            coef[g, v, fill[v]] = np.float32(dinv[u] * dinv[v])
            fill[v] += 1
    return nbr, coef, self_coef


# ---- the belief filter's LDS layout: which scratch slot a node's c = b / deg goes to, and in which order a node gathers its
# neighbours, chosen per board so that the gathers of a wave hit distinct LDS banks ------------------------------------------
def belief_lanes(num_nodes: int) -> int:
    """Nodes per lane of the pipeline kernel's belief filter (csrc/sy_rollout3.hpp::BeliefLanes): lane L owns nodes
    NR * L ... NR * L + NR - 1; boards of more than 256 nodes run on the older kernels (no layout)."""
    return 1 if num_nodes <= 64 else (2 if num_nodes <= 128 else (4 if num_nodes <= 256 else 0))


def _edge_colour(edges, num_left, num_right, K):
    """Proper K-edge-colouring of a bipartite multigraph (edges = [(x, y)], x < num_left, y < num_right) by alternating
    paths (König): colour[e] in [0, K) with no two edges at one vertex sharing a colour, wherever both endpoints have at
    most K edges; an edge at an over-full right vertex gets a colour free at its left vertex only (a conflict remains)."""
    at_l = -np.ones((num_left, K), dtype=np.int64)       # edge holding colour k at left vertex x
    at_r = -np.ones((num_right, K), dtype=np.int64)
    colour = -np.ones(len(edges), dtype=np.int64)
    for e, (x, y) in enumerate(edges):
        free_l = np.flatnonzero(at_l[x] < 0)
        free_r = np.flatnonzero(at_r[y] < 0)
        if free_l.size == 0:
            raise ValueError("a left vertex has more than K edges")
        a = int(free_l[0])
        if free_r.size == 0:                             # over-full right vertex: the conflict stays
            colour[e] = a
            at_l[x, a] = e
            continue
        both = np.intersect1d(free_l, free_r)
        if both.size:
            a = int(both[0])
        else:
            b = int(free_r[0])
            # free colour a at y: flip a <-> b along the alternating path that starts at y with its a-edge (it cannot
            # reach x: x has no a-edge, and the path enters left vertices by a-edges only)
            path, side, v, want = [], 1, y, a
            while True:
                f = at_r[v, want] if side == 1 else at_l[v, want]
                if f < 0:
                    break
                path.append(int(f))
                v = edges[f][0] if side == 1 else edges[f][1]
                side ^= 1
                want = b if want == a else a
            for f in path:
                fx, fy = edges[f]
                at_l[fx, colour[f]] = -1
                at_r[fy, colour[f]] = -1
            for f in path:
                fx, fy = edges[f]
                colour[f] = b if colour[f] == a else a
                at_l[fx, colour[f]] = f
                at_r[fy, colour[f]] = f
        colour[e] = a
        at_l[x, a] = e
        at_r[y, a] = e
    return colour


def belief_layout(ell_rows: np.ndarray, num_nodes: int, node_stride: int):
    """(slot uint16 [NS], gather uint16 [N][16]) for one board, or None for boards the pipeline kernel does not run.

    The filter's diffusion step gathers, for every node, c[u] = b[u] / deg(u) of its neighbours u from an LDS scratch of 8-byte
    entries (both episodes of a pair): 8 gathers per node (8 more only on boards with a node of more than 8 neighbours),
    each a wave-wide `ds_read_b64` of 64 scattered addresses.  LDS banking is per half-wave (MI355X: 64 banks of 4 bytes, an
    8-byte entry takes a pair: entry s sits on pair s mod 32); in node order the 32 addresses of a half-wave land ~3.5-deep
    on the fullest pair: the filter was the engine's bank-conflict source (48.9 of its 154 LDS cycles per env-step).
    Both the entry a node's value goes to and the ORDER in which a node visits its neighbours are free, so:
      1. nodes get one of 32 colours (bank pairs), greedily, largest in-degree first, such that no colour is asked for more
         than 8 times by the nodes of one (lane part, half-wave) group — the 8 gather instructions of that group;
      2. per group, the bipartite multigraph (node -> colours of its neighbours) is edge-coloured with 8 colours (König):
         the colour of an edge is the gather instruction in which that neighbour is visited, so no instruction of a
         half-wave asks one bank pair for two different entries;
      3. entries: colour g owns slots g, g + 32, ...; slot N stays the zero entry that padding gathers read.
    `gather[v][k]` is the scratch BYTE offset of node v's k-th visit (slot * 8; padding -> N * 8)."""
    N, NS = int(num_nodes), int(node_stride)
    NR = belief_lanes(N)
    if NR == 0:
        return None
    rows = np.asarray(ell_rows, dtype=np.uint32).reshape(N, 16)
    nbr = (rows & 0xFFFF).astype(np.int64)
    nslots = NS + 16
    zero = N
    K, C = 8, 32
    neigh = [[int(u) for u in nbr[v] if u < N] for v in range(N)]
    if max((len(x) for x in neigh), default=0) > 16:
        return None
    group = lambda v: (v % NR) * 2 + ((v // NR) >> 5)            # noqa: E731   (lane part r, half-wave) of node v's lane
    ngroups = 2 * NR
    first = [x[:K] for x in neigh]                                # visits 0..7; the rest (rare) in visits 8..15
    second = [x[K:] for x in neigh]
    # 1. colours
    cap = np.array([sum(1 for s in range(c, nslots, C) if s != zero) for c in range(C)])
    want = np.zeros((N, ngroups), dtype=np.int64)                 # how often group g asks for node u (first visits)
    for v in range(N):
        for u in first[v]:
            want[u, group(v)] += 1
    cap[zero % C] = 0                                             # nobody shares the zero entry's bank pair
    load = np.zeros((ngroups, C), dtype=np.int64)
    used = np.zeros(C, dtype=np.int64)
    colour = -np.ones(N, dtype=np.int64)
    # the lanes' own entries are written by `ds_write_b64` (16 contiguous lanes per LDS cycle, 32 banks: entry mod 16): the 16
    # nodes of one (lane part, quarter-wave) take 16 different residues mod 16
    wgroup = lambda v: (v % NR) * 4 + ((v // NR) >> 4)           # noqa: E731
    taken = np.zeros((4 * NR, 16), dtype=bool)
    for u in np.argsort(-want.sum(1), kind="stable"):
        peak = (load + want[u][:, None]).max(0).astype(np.float64)   # fullest group-colour cell if u took colour c
        peak[used >= cap] = np.inf
        peak[np.tile(taken[wgroup(u)], 2)] = np.inf
        if not np.isfinite(peak).any():                           # (cannot happen while 16 residues x 2 >= the group's nodes)
            peak = (load + want[u][:, None]).max(0).astype(np.float64)
            peak[used >= cap] = np.inf
        best = np.flatnonzero(peak == peak.min())
        c = int(best[np.argmin(used[best])])
        colour[u] = c
        used[c] += 1
        taken[wgroup(u), c % 16] = True
        load[:, c] += want[u]
    # 3. slots
    slot = np.zeros(NS, dtype=np.uint16)
    nxt = {c: [s for s in range(c, nslots, C) if s != zero] for c in range(C)}
    for u in range(N):
        slot[u] = nxt[int(colour[u])].pop(0)
    for u in range(N, NS):
        slot[u] = zero
    # 2. visit order per group
    gather = np.full((N, 16), zero * 8, dtype=np.uint16)
    for which, base in ((first, 0), (second, K)):
        for g in range(ngroups):
            members = [v for v in range(N) if group(v) == g and which[v]]
            if not members:
                continue
            index = {v: i for i, v in enumerate(members)}
            edges = [(index[v], int(colour[u])) for v in members for u in which[v]]
            who = [u for v in members for u in which[v]]
            col = _edge_colour(edges, len(members), C, K)
            for (x, _), u, k in zip(edges, who, col):
                gather[members[x], base + int(k)] = int(slot[u]) * 8
    return slot, gather


def belief_bank_conflicts(gather: np.ndarray, slot: np.ndarray, num_nodes: int, node_stride: int) -> dict:
    """Extra LDS cycles of one diffusion step under the MI355X bank model (a `ds_read_b64` is served per half-wave, bank pair
    = entry mod 32, identical addresses broadcast, every further distinct entry on a busy pair costs a cycle): the 8 (or 16)
    gathers of every lane part, and the scatter of the lanes' own entries."""
    N, NR = int(num_nodes), belief_lanes(num_nodes)
    zero = N * 8
    wide = bool((gather[:, 8:] != zero).any())
    extra_g = extra_w = 0
    for r in range(NR):
        for half in range(2):
            lanes = range(32 * half, 32 * half + 32)
            for k in range(16 if wide else 8):
                addr = {int(gather[NR * L + r, k]) if NR * L + r < N else zero for L in lanes}
                pairs = {}
                for a in addr:
                    pairs[(a // 8) % 32] = pairs.get((a // 8) % 32, 0) + 1
                extra_g += sum(c - 1 for c in pairs.values())
        # the lanes' own entries: ds_write_b64, 4 x 16 contiguous lanes, bank pair of the 32-bank store view = entry mod 16
        for q in range(4):
            own = {int(slot[NR * L + r]) for L in range(16 * q, 16 * q + 16) if NR * L + r < N}
            pairs = {}
            for s in own:
                pairs[s % 16] = pairs.get(s % 16, 0) + 1
            extra_w += sum(c - 1 for c in pairs.values())
    return {"gather_extra_cycles": extra_g, "store_extra_cycles": extra_w, "gather_instructions": NR * (16 if wide else 8)}
