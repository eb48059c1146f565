"""Shape-only stand-ins for gymnasium.spaces (see package docstring)."""
from collections import namedtuple

import numpy as np

GraphInstance = namedtuple("GraphInstance", ["nodes", "edges", "edge_links"])


class Space:
    dtype = None
    shape = None


class Discrete(Space):
    def __init__(self, n, start=0, seed=None):
        self.n = int(n)
        self.start = int(start)
        self.dtype = np.dtype(np.int64)
        self.shape = ()

    def __repr__(self):
        return f"Discrete({self.n}, start={self.start})"


class MultiDiscrete(Space):
    def __init__(self, nvec, dtype=np.int64, seed=None):
        self.nvec = np.asarray(nvec, dtype=dtype)
        self.dtype = np.dtype(dtype)
        self.shape = self.nvec.shape


class MultiBinary(Space):
    def __init__(self, n, seed=None):
        self.n = n
        self.dtype = np.dtype(np.int8)
        self.shape = (n,) if np.isscalar(n) else tuple(n)


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        self.low, self.high = low, high
        self.shape = tuple(shape) if shape is not None else np.shape(low)
        self.dtype = np.dtype(dtype)


class Dict(Space):
    def __init__(self, spaces=None, seed=None, **kw):
        self.spaces = dict(spaces or {})
        self.spaces.update(kw)

    def __getitem__(self, k):
        return self.spaces[k]

    def keys(self):
        return self.spaces.keys()

    def items(self):
        return self.spaces.items()


class Graph(Space):
    def __init__(self, node_space, edge_space, seed=None):
        self.node_space = node_space
        self.edge_space = edge_space

    def from_jsonable(self, sample_n):
        out = []
        for s in sample_n:
            nodes = np.asarray(s["nodes"], dtype=self.node_space.dtype)
            edges = np.asarray(s["edges"], dtype=self.edge_space.dtype)
            links = np.asarray(s["edge_links"], dtype=np.int32).reshape(-1, 2)
            out.append(GraphInstance(nodes, edges, links))
        return out
