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
