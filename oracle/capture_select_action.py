#!/usr/bin/env python3
"""Golden vectors of the reference's masked action sampling (build container only; TEST INFRASTRUCTURE).

Drives the UNMODIFIED `MappoAgent.select_action` (/root/reference/src/agent/mappo_agent.py:87-142,
imported by file path: it only needs torch).  The actor network is a data holder here: it is swapped
for a callable that returns a prescribed probability vector, so the function under test sees exactly
(probs, mask) and returns (action, log_prob, normalised probs).  Writes
tests/golden/select_action_reference.json.

    python oracle/capture_select_action.py
"""
import importlib.util
import json
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SY_REFERENCE", "/root/reference")


def main():
    spec = importlib.util.spec_from_file_location("ref_mappo_agent", os.path.join(REF, "src", "agent", "mappo_agent.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(20240607)
    cases = []
    for ci in range(16):
        n = int(rng.choice([6, 15, 40, 200]))
        logits = rng.normal(size=n) * 2.0
        probs = np.exp(logits - logits.max())
        probs = (probs / probs.sum()).astype(np.float32)
        mask = (rng.random(n) < (0.5 if n <= 15 else 0.05)).astype(np.float32)
        kind = "random"
        if ci == 3:                      # every legal action has zero probability -> uniform over the mask
            mask[:] = 0
            mask[[1, 4]] = 1
            probs[[1, 4]] = 0.0
            kind = "mass_masked_out"
        if ci == 5:                      # empty mask -> uniform over all nodes
            mask[:] = 0
            kind = "empty_mask"
        if ci == 7:
            mask[:] = 1
            kind = "all_legal"
        agent = mod.MappoAgent(n_agents=1, obs_size=n, global_obs_size=n, action_size=n, hidden_size=4)
        vec = torch.tensor(probs)
        agent.policies[0] = lambda obs, vec=vec: vec.unsqueeze(0)     # the actor's softmax output, prescribed
        torch.manual_seed(1000 + ci)
        action, log_prob, current = agent.select_action(0, torch.zeros(n), torch.tensor(mask))
        cases.append({"kind": kind, "probs": [float(x) for x in probs], "mask": [int(x) for x in mask],
                      "action": int(action), "log_prob": float(log_prob), "current_probs": [float(x) for x in current]})
    out = os.path.join(HERE, "..", "tests", "golden", "select_action_reference.json")
    with open(out, "w") as f:
        json.dump({"source": "agent/mappo_agent.py:87-142 (unmodified), torch " + torch.__version__, "cases": cases}, f)
    print("wrote", os.path.abspath(out), len(cases), "cases")


if __name__ == "__main__":
    main()
