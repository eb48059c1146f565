#!/usr/bin/env python3
"""Statistics of the reference's board sampler (build container only; TEST INFRASTRUCTURE).

Runs the UNMODIFIED `ConnectedGraph.sample` / `_create_tree` (/root/reference/src/environment/graph_layout.py:9-80,
imported by file path with oracle/refstubs standing in for gymnasium's `Graph` container) for the board sizes the
reference's configs use — (15, 20) test/env_test.py, (50, 110) main.py:185-186, (100, 190) big_graph/config.yml,
(200, 400) the BASELINE workload — and records what the engine's own samplers (host `graph.sample_board`, device
`sy_sample_boards`; own RNG streams, so parity is statistical) are held to:
  * degree histogram over all nodes of all boards, histogram of the per-board maximum degree,
  * histogram of the degrees the spanning tree alone produces (the tree is not degree-capped),
  * edge-weight histogram (np.random.randint(1, 5) -> 1..4), realised edge counts.
Writes tests/golden/sampler_stats.json.

    python oracle/capture_sampler_stats.py
"""
import importlib.util
import json
import os
import random
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SY_REFERENCE", "/root/reference")

CONFIGS = [(15, 20, 2000), (50, 110, 600), (100, 190, 300), (200, 400, 150)]   # (nodes, edges, boards drawn)


def main():
    sys.path.insert(0, os.path.join(HERE, "refstubs"))
    spec = importlib.util.spec_from_file_location("ref_graph_layout", os.path.join(REF, "src", "environment", "graph_layout.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    from gymnasium.spaces import Box, Discrete
    out = {"source": "environment/graph_layout.py:9-80 (unmodified) through oracle/refstubs", "configs": []}
    for n, e, boards in CONFIGS:
        random.seed(1000 + n)
        np.random.seed(2000 + n)
        space = mod.ConnectedGraph(node_space=Discrete(1), edge_space=Discrete(mod.ConnectedGraph.MAX_WEIGHT, start=1))
        deg_hist = np.zeros(64, dtype=np.int64)
        tree_hist = np.zeros(64, dtype=np.int64)
        maxdeg_hist = np.zeros(64, dtype=np.int64)
        w_hist = np.zeros(8, dtype=np.int64)
        edge_counts = []
        t0 = time.time()
        for _ in range(boards):
            g = space.sample(num_nodes=n, num_edges=e)
            links = np.asarray(g.edge_links).reshape(-1, 2)
            w = np.asarray(g.edges).reshape(-1)
            deg = np.bincount(links.reshape(-1), minlength=n)
            tdeg = np.bincount(links[: n - 1].reshape(-1), minlength=n)       # the first n-1 edges are the tree (:15-16)
            deg_hist += np.bincount(deg, minlength=64)[:64]
            tree_hist += np.bincount(tdeg, minlength=64)[:64]
            maxdeg_hist[int(deg.max())] += 1
            w_hist += np.bincount(w, minlength=8)[:8]
            edge_counts.append(int(links.shape[0]))
        out["configs"].append({"nodes": n, "edges_requested": e, "boards": boards, "max_edges_per_node": 4,
                               "degree_hist": deg_hist.tolist(), "tree_degree_hist": tree_hist.tolist(),
                               "max_degree_hist": maxdeg_hist.tolist(), "weight_hist": w_hist.tolist(),
                               "edge_counts": edge_counts})
        print(f"N={n} E={e}: {boards} boards in {time.time() - t0:.1f}s, edges {min(edge_counts)}..{max(edge_counts)}, "
              f"max degree up to {int(np.nonzero(maxdeg_hist)[0].max())}")
    path = os.path.join(HERE, "..", "tests", "golden", "sampler_stats.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", os.path.abspath(path))


if __name__ == "__main__":
    main()
