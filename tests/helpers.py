"""Shared helpers for the parity tests (fixture loading, oracle replay)."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
NONE_ACTION = -2  # fixture encoding of a Python None action; the engine treats None as -1 (no-op)


def trace_index():
    with open(os.path.join(GOLDEN, "traces_index.json")) as f:
        return json.load(f)


def load_trace(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: z[k] for k in z.files}


def engine_actions(actions):
    """None (-2 in fixtures) and -1 are the same no-op for every agent (yard.py:161,210-215)."""
    a = np.array(actions, dtype=np.int32, copy=True)
    a[a == NONE_ACTION] = -1
    return a


def metrics_golden_record(g, device="cpu"):
    """tests/golden/metrics_reference.json laid out as a rollout record: one env per episode of the reference's
    MetricsTracker run, finishing at row length - 1 with the recorded winner; rows whose step ends with a reveal
    carry the belief / MrX node the tracker was given (eval/metrics.py:138-141), all other rows a uniform belief."""
    import torch
    eps, N = g["episodes"], int(g["num_nodes"])
    T, B, A, NS = max(e["length"] for e in eps) + 1, len(eps), 3, 16
    rec = {
        "terminated": torch.zeros(T, B, dtype=torch.int32), "truncated": torch.zeros(T, B, dtype=torch.int32),
        "winner": torch.zeros(T, B, dtype=torch.int32), "t": torch.arange(T).unsqueeze(1).expand(T, B).clone().int(),
        "budget": torch.full((T, B, A), 5, dtype=torch.int32), "pos": torch.zeros(T, B, A, dtype=torch.int32),
        "belief": torch.zeros(T, B, NS, dtype=torch.float64),
    }
    rec["belief"][..., :N] = 1.0 / N
    for b, e in enumerate(eps):
        s = e["length"] - 1
        rec["terminated"][s, b] = 1
        rec["winner"][s, b] = 1 if e["winner"] == "Police" else 2
        # rows after the end belong to a next episode that finishes nothing and reveals nothing inside the record
        rec["t"][s + 1:, b] = 1
        for r in e["reveals"]:
            row = r["step"] - 1                      # tracker step k = the k-th transition = record row k - 1 (t = k - 1)
            rec["belief"][row, b, :N] = torch.tensor(r["belief"], dtype=torch.float64)
            rec["pos"][row, b, 0] = r["true_index"]
    return {k: v.to(device) for k, v in rec.items()}


# ---- board-sampler statistics (tests/golden/sampler_stats.json, oracle/capture_sampler_stats.py)
def sampler_stats(num_nodes, edge_links_list, weights_list):
    """Same statistics as the capture script records, for boards given as (edge_links [E,2], weights [E]) pairs
    in insertion order (the first N-1 edges are the spanning tree)."""
    n = int(num_nodes)
    out = {"degree_hist": np.zeros(64, dtype=np.int64), "tree_degree_hist": np.zeros(64, dtype=np.int64),
           "max_degree_hist": np.zeros(64, dtype=np.int64), "weight_hist": np.zeros(8, dtype=np.int64), "edge_counts": []}
    for links, w in zip(edge_links_list, weights_list):
        links = np.asarray(links).reshape(-1, 2)
        deg = np.bincount(links.reshape(-1), minlength=n)
        out["degree_hist"] += np.bincount(deg, minlength=64)[:64]
        out["tree_degree_hist"] += np.bincount(np.bincount(links[: n - 1].reshape(-1), minlength=n), minlength=64)[:64]
        out["max_degree_hist"][int(deg.max())] += 1
        out["weight_hist"] += np.bincount(np.asarray(w).reshape(-1), minlength=8)[:8]
        out["edge_counts"].append(int(links.shape[0]))
    return out


def assert_sampler_stats_close(ref, got, what="", z=6.0):
    """Two-sample comparison with the stated sampling error: every histogram bin's frequency within
    z standard errors of the pooled binomial (+1e-3 absolute; node degrees of one board are negatively
    correlated, which only tightens the true spread), means of max degree / realised edge count within
    z standard errors of the mean."""
    for key in ("degree_hist", "tree_degree_hist", "weight_hist", "max_degree_hist"):
        a, b = np.asarray(ref[key], dtype=np.float64), np.asarray(got[key], dtype=np.float64)
        na, nb = a.sum(), b.sum()
        p = (a + b) / (na + nb)
        se = np.sqrt(p * (1 - p) * (1 / na + 1 / nb))
        bad = np.abs(a / na - b / nb) > z * se + 1e-3
        assert not bad.any(), f"{what} {key}: bins {np.nonzero(bad)[0]} ref {a[bad] / na} got {b[bad] / nb} (se {se[bad]})"
    k = np.arange(64)
    for key in ("max_degree_hist",):
        a, b = np.asarray(ref[key], dtype=np.float64), np.asarray(got[key], dtype=np.float64)
        ma, mb = (a * k).sum() / a.sum(), (b * k).sum() / b.sum()
        va = (a * (k - ma) ** 2).sum() / a.sum()
        assert abs(ma - mb) <= z * np.sqrt(va * (1 / a.sum() + 1 / b.sum())) + 0.05, f"{what} mean max degree {ma} vs {mb}"
    ea, eb = np.asarray(ref["edge_counts"], dtype=np.float64), np.asarray(got["edge_counts"], dtype=np.float64)
    assert abs(ea.mean() - eb.mean()) <= z * np.sqrt(ea.var() * (1 / ea.size + 1 / eb.size)) + 0.05, \
        f"{what} realised edges {ea.mean()} vs {eb.mean()}"
    assert eb.min() >= ea.min() - 3 and eb.max() <= ea.max() + 3, f"{what} realised edge range {eb.min()}..{eb.max()}"
