#!/usr/bin/env python3
"""Golden vectors of the reference's evaluation metrics (build container only; TEST INFRASTRUCTURE).

Runs the UNMODIFIED /root/reference/src/eval/metrics.py (imported by file path: numpy only):
`belief_cross_entropy` on random beliefs, and `MetricsTracker` fed with a list of synthetic episodes
(winner, length) -> `get_aggregated_metrics()`.  Writes tests/golden/metrics_reference.json.

    python oracle/capture_metrics.py
"""
import importlib.util
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SY_REFERENCE", "/root/reference")


def main():
    spec = importlib.util.spec_from_file_location("ref_eval_metrics", os.path.join(REF, "src", "eval", "metrics.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_eval_metrics"] = mod          # dataclasses look the module up by name
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(99)
    ce_cases = []
    for i in range(24):
        n = int(rng.integers(3, 40))
        b = rng.random(n)
        if i % 5 == 0:
            b[rng.integers(0, n)] = 0.0             # exercises the 1e-8 clip
        if i % 7 == 0:
            b[:] = 0.0
            b[rng.integers(0, n)] = 1.0             # a delta
        b = b / b.sum()
        k = int(rng.integers(0, n))
        ce_cases.append({"belief": [float(x) for x in b], "true_index": k, "ce": mod.belief_cross_entropy(b.copy(), k)})
    episodes = []
    tracker = mod.MetricsTracker()
    reveal_k, n_nodes = 5, 10
    for i in range(37):
        winner = "Police" if rng.random() < 0.35 else "MrX"
        length = int(rng.integers(1, 60))
        tracker.start_episode(initial_budget=20.0)
        reveals = []
        for step in range(1, length + 1):
            # belief quality is recorded at reveal times only (eval/metrics.py:138-141): every 5th step here
            if step % reveal_k == 0:
                b = rng.random(n_nodes)
                if rng.random() < 0.2:
                    b[rng.integers(0, n_nodes)] = 0.0
                b = b / b.sum()
                k = int(rng.integers(0, n_nodes))
                tracker.record_step(step, belief=b.copy(), true_mrx_pos=k, is_reveal=True)
                reveals.append({"step": step, "belief": [float(x) for x in b], "true_index": k})
            else:
                tracker.record_step(step)
        tracker.end_episode(winner)
        episodes.append({"winner": winner, "length": length, "reveals": reveals})
    agg = tracker.get_aggregated_metrics().to_dict()
    out = os.path.join(HERE, "..", "tests", "golden", "metrics_reference.json")
    with open(out, "w") as f:
        json.dump({"source": "eval/metrics.py (unmodified)", "reveal_interval": reveal_k, "num_nodes": n_nodes, "ce_cases": ce_cases, "episodes": episodes,
                   "aggregated": {k: (float(v) if isinstance(v, (int, float, np.floating, np.integer)) else v)
                                  for k, v in agg.items()}}, f)
    print("wrote", os.path.abspath(out), sorted(agg)[:20])


if __name__ == "__main__":
    main()
