"""ctypes front-end of the CPU oracle (oracle/sy_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg as the checker — never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libsy_oracle.so")
MAX_AGENTS = 8
NUM_WEIGHTS = 11
MRX_MONEY = 1000

_i32p = C.POINTER(C.c_int32)
_u32p = C.POINTER(C.c_uint32)
_u8p = C.POINTER(C.c_uint8)
_i8p = C.POINTER(C.c_int8)
_f64p = C.POINTER(C.c_double)


class _BatchConfig(C.Structure):
    _fields_ = [("B", C.c_int32), ("N", C.c_int32), ("P", C.c_int32), ("money0", C.c_int32),
                ("max_t", C.c_int32), ("node_stride", C.c_int32), ("reveal_interval", C.c_int32),
                ("police_evidence", C.c_int32), ("belief_init_onehot", C.c_int32),
                ("auto_reset", C.c_int32), ("env_id_offset", C.c_uint64), ("threads", C.c_int32)]


class _BatchState(C.Structure):
    _fields_ = [("pos", _i32p), ("money", _i32p), ("t", _i32p), ("step_count", _u32p),
                ("visits", _i32p), ("belief", _f64p), ("mask", _u8p), ("reward", _f64p),
                ("terminated", _u8p), ("truncated", _u8p), ("winner", _i8p)]


class _Traj(C.Structure):
    _fields_ = [("pos", _i32p), ("money", _i32p), ("t", _i32p), ("action", _i32p), ("mask", _u8p),
                ("belief", _f64p), ("reward", _f64p), ("terminated", _u8p), ("truncated", _u8p),
                ("winner", _i8p)]


_lib = None


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or \
            os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(HERE, "sy_oracle.c")):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return LIB_PATH


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    lib.syo_graph_create.restype = C.c_void_p
    lib.syo_graph_create.argtypes = [C.c_int32, C.c_int32, _i32p, _i32p]
    lib.syo_graph_destroy.argtypes = [C.c_void_p]
    lib.syo_graph_dist.restype = _i32p
    lib.syo_graph_dist.argtypes = [C.c_void_p]
    lib.syo_graph_wmin.restype = _i32p
    lib.syo_graph_wmin.argtypes = [C.c_void_p]
    lib.syo_possible_moves.restype = C.c_int32
    lib.syo_possible_moves.argtypes = [C.c_void_p, C.c_int32, C.c_int64, _i32p, _i32p]
    lib.syo_action_mask_dense.argtypes = [_f64p, _f64p, _f64p, C.c_int32, C.c_int32, C.c_double, _u8p]
    lib.syo_env_masks.argtypes = [C.c_void_p, C.c_int32, _i32p, _i32p, _u8p, C.c_int32]
    lib.syo_default_tables.argtypes = [_f64p, C.c_int32, _f64p, C.c_int32]
    lib.syo_step_one.restype = C.c_int32
    lib.syo_step_one.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _i32p, _i32p, _i32p, _i32p, _i32p,
                                 _f64p, _f64p, C.c_int32, _f64p, C.c_int32, _f64p, _u8p, _u8p, _i8p]
    lib.syo_belief_update.argtypes = [C.c_void_p, _f64p, _i32p, C.c_int32, C.c_int32, _i32p, C.c_int32]
    lib.syo_philox4x32.argtypes = [C.c_uint32] * 6 + [_u32p]
    gpp = C.POINTER(C.c_void_p)
    lib.syo_batch_reset.argtypes = [C.POINTER(_BatchConfig), gpp, _i32p, C.POINTER(_BatchState), _u8p, C.c_uint64]
    lib.syo_batch_reset_to.argtypes = [C.POINTER(_BatchConfig), gpp, _i32p, C.POINTER(_BatchState), _i32p]
    lib.syo_batch_step.argtypes = [C.POINTER(_BatchConfig), gpp, _i32p, C.POINTER(_BatchState), _i32p,
                                   _f64p, _f64p, C.c_int32, _f64p, C.c_int32, C.c_uint64]
    lib.syo_batch_rollout.argtypes = [C.POINTER(_BatchConfig), gpp, _i32p, C.POINTER(_BatchState), C.c_int32,
                                      _f64p, _f64p, C.c_int32, _f64p, C.c_int32, C.c_uint64, C.POINTER(_Traj)]
    _f32p = C.POINTER(C.c_float)
    lib.syo_discounted_returns_f32.argtypes = [_f32p, _u8p, _f32p, C.c_int32, C.c_int32, C.c_float, _f32p, _f32p]
    lib.syo_gae_f64.argtypes = [_f64p, _u8p, _f64p, _f64p, C.c_int32, C.c_int32, C.c_double, C.c_double, _f64p, _f64p]
    _lib = lib
    return lib


def _p(a, typ):
    return None if a is None else a.ctypes.data_as(typ)


class OracleGraph:
    """Board from the reference's arrays: edge_links int32[E,2], edge weights [E]."""

    def __init__(self, num_nodes, edge_links, edge_w):
        lib = load()
        self.N = int(num_nodes)
        self.edge_links = np.ascontiguousarray(edge_links, dtype=np.int32).reshape(-1, 2)
        self.edge_w = np.ascontiguousarray(edge_w, dtype=np.int32)
        self.E = self.edge_links.shape[0]
        self.handle = lib.syo_graph_create(self.N, self.E, _p(self.edge_links, _i32p), _p(self.edge_w, _i32p))
        if not self.handle:
            raise ValueError("bad graph")
        n2 = self.N * self.N
        self.dist = np.ctypeslib.as_array(lib.syo_graph_dist(self.handle), shape=(n2,)).reshape(self.N, self.N).copy()
        self.wmin = np.ctypeslib.as_array(lib.syo_graph_wmin(self.handle), shape=(n2,)).reshape(self.N, self.N).copy()

    def __del__(self):
        try:
            if self.handle:
                load().syo_graph_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def possible_moves(self, pos, money):
        nodes = np.zeros(self.N, dtype=np.int32)
        w = np.zeros(self.N, dtype=np.int32)
        k = load().syo_possible_moves(self.handle, int(pos), int(money), _p(nodes, _i32p), _p(w, _i32p))
        return nodes[:k].copy(), w[:k].copy()

    def env_masks(self, pos, money, stride=None):
        P = len(pos) - 1
        stride = stride or self.N
        pos = np.ascontiguousarray(pos, dtype=np.int32)
        money = np.ascontiguousarray(money, dtype=np.int32)
        m = np.zeros((P + 1, stride), dtype=np.uint8)
        load().syo_env_masks(self.handle, P, _p(pos, _i32p), _p(money, _i32p), _p(m, _u8p), stride)
        return m


def default_tables(n_exp=1024, n_cov=512):
    e = np.zeros(n_exp, dtype=np.float64)
    c = np.zeros(n_cov, dtype=np.float64)
    load().syo_default_tables(_p(e, _f64p), n_exp, _p(c, _f64p), n_cov)
    return e, c


def action_mask_dense(adjacency, current_node, budget, tolls=None, edge_weights=None):
    """Same argument meaning as the reference's compute_action_mask (tolls already a matrix or None)."""
    adj = np.ascontiguousarray(adjacency, dtype=np.float64)
    n = adj.shape[0]
    w = None if edge_weights is None else np.ascontiguousarray(edge_weights, dtype=np.float64)
    t = None if tolls is None else np.ascontiguousarray(tolls, dtype=np.float64)
    mask = np.zeros(n, dtype=np.uint8)
    load().syo_action_mask_dense(_p(adj, _f64p), _p(w, _f64p), _p(t, _f64p), n, int(current_node),
                                 float(budget), _p(mask, _u8p))
    return mask.astype(bool)


def normalize_tolls(tolls, n):
    """action_mask.py:87-97 restated for the checker (scalar -> full, vector -> per destination)."""
    if tolls is None:
        return None
    if np.isscalar(tolls):
        return np.full((n, n), float(tolls))
    t = np.asarray(tolls, dtype=float)
    if t.ndim == 1:
        return np.tile(t.reshape(1, -1), (n, 1))
    return t


class OracleEpisode:
    """Single env stepped by syo_step_one (used to replay golden traces)."""

    def __init__(self, graph, P, money0, starts, weights, max_t=250, tables=None):
        self.g, self.P, self.max_t = graph, int(P), int(max_t)
        self.pos = np.ascontiguousarray(starts, dtype=np.int32).copy()
        self.money = np.array([MRX_MONEY] + [int(money0)] * self.P, dtype=np.int32)
        self.t = np.zeros(1, dtype=np.int32)
        self.visits = np.zeros(graph.N, dtype=np.int32)
        self.weights = np.ascontiguousarray(weights, dtype=np.float64)
        self.tables = tables
        self.reward = np.zeros(self.P + 1, dtype=np.float64)
        self.flags = np.zeros(2, dtype=np.uint8)
        self.winner = np.zeros(1, dtype=np.int8)

    def masks(self):
        return self.g.env_masks(self.pos, self.money).astype(bool)

    def step(self, actions):
        act = np.ascontiguousarray(actions, dtype=np.int32)
        et, ct = (self.tables if self.tables is not None else (None, None))
        ended = load().syo_step_one(
            self.g.handle, self.P, self.max_t, _p(self.pos, _i32p), _p(self.money, _i32p), _p(self.t, _i32p),
            _p(self.visits, _i32p), _p(act, _i32p), _p(self.weights, _f64p),
            _p(et, _f64p), 0 if et is None else len(et), _p(ct, _f64p), 0 if ct is None else len(ct),
            _p(self.reward, _f64p), _p(self.flags[0:1], _u8p), _p(self.flags[1:2], _u8p), _p(self.winner, _i8p))
        return bool(ended)


def belief_update(graph, belief, hint=None, reveal=None, zero_nodes=None):
    b = np.ascontiguousarray(belief, dtype=np.float64).copy()
    h = None if not hint else np.ascontiguousarray(hint, dtype=np.int32)
    z = None if zero_nodes is None or len(zero_nodes) == 0 else np.ascontiguousarray(zero_nodes, dtype=np.int32)
    load().syo_belief_update(graph.handle, _p(b, _f64p), _p(h, _i32p), 0 if h is None else len(h),
                             -1 if reveal is None else int(reveal), _p(z, _i32p), 0 if z is None else len(z))
    return b


def philox(c0, c1, c2, c3, k0, k1):
    out = np.zeros(4, dtype=np.uint32)
    load().syo_philox4x32(c0, c1, c2, c3, k0, k1, _p(out, _u32p))
    return out


class OracleBatch:
    """Batched engine mirror (auto-reset + uniform-random policy) — the checker for the device engine."""

    def __init__(self, graphs, env_graph, B, P, money0, max_t=250, node_stride=None, reveal_interval=0,
                 police_evidence=False, belief_init_onehot=False, auto_reset=True, env_id_offset=0,
                 threads=1, weights=None, tables=None, with_belief=True):
        self.graphs = list(graphs)
        self.N = self.graphs[0].N
        self.B, self.P, self.A = int(B), int(P), int(P) + 1
        self.NS = int(node_stride or self.N)
        self.cfg = _BatchConfig(self.B, self.N, self.P, int(money0), int(max_t), self.NS, int(reveal_interval),
                                int(bool(police_evidence)), int(bool(belief_init_onehot)), int(bool(auto_reset)),
                                int(env_id_offset), int(threads))
        self.env_graph = np.ascontiguousarray(env_graph, dtype=np.int32)
        self._garr = (C.c_void_p * len(self.graphs))(*[g.handle for g in self.graphs])
        A, NS = self.A, self.NS
        self.pos = np.zeros((B, A), np.int32)
        self.money = np.zeros((B, A), np.int32)
        self.t = np.zeros(B, np.int32)
        self.step_count = np.zeros(B, np.uint32)
        self.visits = np.zeros((B, NS), np.int32)
        self.belief = np.zeros((B, NS), np.float64) if with_belief else None
        self.mask = np.zeros((B, A, NS), np.uint8)
        self.reward = np.zeros((B, A), np.float64)
        self.terminated = np.zeros(B, np.uint8)
        self.truncated = np.zeros(B, np.uint8)
        self.winner = np.zeros(B, np.int8)
        self.state = _BatchState(_p(self.pos, _i32p), _p(self.money, _i32p), _p(self.t, _i32p),
                                 _p(self.step_count, _u32p), _p(self.visits, _i32p), _p(self.belief, _f64p),
                                 _p(self.mask, _u8p), _p(self.reward, _f64p), _p(self.terminated, _u8p),
                                 _p(self.truncated, _u8p), _p(self.winner, _i8p))
        self.weights = np.ascontiguousarray(weights if weights is not None else np.full(NUM_WEIGHTS, 0.5), np.float64)
        self.tables = tables
        self.seed = 0

    def _tabs(self):
        et, ct = (self.tables if self.tables is not None else (None, None))
        return _p(et, _f64p), 0 if et is None else len(et), _p(ct, _f64p), 0 if ct is None else len(ct)

    def reset(self, seed=0, env_sel=None):
        self.seed = int(seed)
        sel = None if env_sel is None else np.ascontiguousarray(env_sel, dtype=np.uint8)
        load().syo_batch_reset(C.byref(self.cfg), self._garr, _p(self.env_graph, _i32p), C.byref(self.state),
                               _p(sel, _u8p), self.seed)

    def reset_to(self, starts):
        st = np.ascontiguousarray(starts, dtype=np.int32)
        load().syo_batch_reset_to(C.byref(self.cfg), self._garr, _p(self.env_graph, _i32p), C.byref(self.state),
                                  _p(st, _i32p))

    def step(self, actions):
        act = np.ascontiguousarray(actions, dtype=np.int32)
        load().syo_batch_step(C.byref(self.cfg), self._garr, _p(self.env_graph, _i32p), C.byref(self.state),
                              _p(act, _i32p), _p(self.weights, _f64p), *self._tabs(), self.seed)

    def rollout(self, T, record=True, record_mask=True, record_belief=True):
        B, A, NS = self.B, self.A, self.NS
        tr = None
        ctr = None
        if record:
            tr = dict(pos=np.zeros((T, B, A), np.int32), money=np.zeros((T, B, A), np.int32),
                      t=np.zeros((T, B), np.int32), action=np.zeros((T, B, A), np.int32),
                      mask=np.zeros((T, B, A, NS), np.uint8) if record_mask else None,
                      belief=np.zeros((T, B, NS), np.float64) if (record_belief and self.belief is not None) else None,
                      reward=np.zeros((T, B, A), np.float64), terminated=np.zeros((T, B), np.uint8),
                      truncated=np.zeros((T, B), np.uint8), winner=np.zeros((T, B), np.int8))
            ctr = _Traj(_p(tr["pos"], _i32p), _p(tr["money"], _i32p), _p(tr["t"], _i32p), _p(tr["action"], _i32p),
                        _p(tr["mask"], _u8p), _p(tr["belief"], _f64p), _p(tr["reward"], _f64p),
                        _p(tr["terminated"], _u8p), _p(tr["truncated"], _u8p), _p(tr["winner"], _i8p))
        load().syo_batch_rollout(C.byref(self.cfg), self._garr, _p(self.env_graph, _i32p), C.byref(self.state),
                                 int(T), _p(self.weights, _f64p), *self._tabs(), self.seed,
                                 C.byref(ctr) if ctr is not None else None)
        return tr


def discounted_returns_f32(reward, done, gamma, values=None):
    """mappo_agent.py:247-258 over [T, cols] float32 columns; returns (returns, advantages = returns - values)."""
    r = np.ascontiguousarray(reward, dtype=np.float32)
    T = r.shape[0]
    cols = int(np.prod(r.shape[1:])) if r.ndim > 1 else 1
    d = np.ascontiguousarray(np.broadcast_to(np.asarray(done).reshape(np.asarray(done).shape + (1,) * (r.ndim - np.asarray(done).ndim)), r.shape), dtype=np.uint8)
    v = None if values is None else np.ascontiguousarray(np.broadcast_to(values, r.shape), dtype=np.float32)
    ret, adv = np.zeros_like(r), np.zeros_like(r)
    f32p = C.POINTER(C.c_float)
    load().syo_discounted_returns_f32(_p(r, f32p), _p(d, _u8p), _p(v, f32p), T, cols, float(gamma), _p(ret, f32p), _p(adv, f32p))
    return ret, adv


def gae_f64(reward, done, values, gamma, lam, last_value=None):
    r = np.ascontiguousarray(reward, dtype=np.float64)
    T = r.shape[0]
    cols = int(np.prod(r.shape[1:])) if r.ndim > 1 else 1
    dd = np.asarray(done)
    d = np.ascontiguousarray(np.broadcast_to(dd.reshape(dd.shape + (1,) * (r.ndim - dd.ndim)), r.shape), dtype=np.uint8)
    v = np.ascontiguousarray(np.broadcast_to(values, r.shape), dtype=np.float64)
    lv = None if last_value is None else np.ascontiguousarray(np.broadcast_to(last_value, r.shape[1:]), dtype=np.float64)
    adv, ret = np.zeros_like(r), np.zeros_like(r)
    load().syo_gae_f64(_p(r, _f64p), _p(d, _u8p), _p(v, _f64p), _p(lv, _f64p), T, cols, float(gamma), float(lam),
                       _p(adv, _f64p), _p(ret, _f64p))
    return adv, ret
