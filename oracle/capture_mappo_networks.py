#!/usr/bin/env python3
"""Golden vectors of the reference's MAPPO networks (build container only; TEST INFRASTRUCTURE).

Instantiates the UNMODIFIED `AgentPolicy` / `CentralCritic` (/root/reference/src/agent/mappo_agent.py:6-44,
imported by file path), seeds their weights, evaluates them on the observations the reference trainer
builds (one-hot MrX node for MrX's actor, multi-hot police nodes for the police actors, their
concatenation [mrx] + [police] * P for the critic: mappo_trainer.py:173,197) and writes weights, inputs
and outputs to tests/golden/mappo_networks_reference.npz (N=14, P=3, hidden 8) and
tests/golden/mappo_networks_reference_h128.npz (N=24, P=4, hidden 128 — the reference's default hidden size,
src/configs/agent/default.yaml:2).

    python oracle/capture_mappo_networks.py
"""
import importlib.util
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SY_REFERENCE", "/root/reference")


def capture(mod, seed, N, P, H, B, fname):
    torch.manual_seed(seed)
    A = P + 1
    actors = [mod.AgentPolicy(N, N, H) for _ in range(A)]
    critic = mod.CentralCritic(N * A, H)
    g = torch.Generator().manual_seed(5)
    pos = torch.stack([torch.randperm(N, generator=g)[:A] for _ in range(B)])      # distinct nodes per env
    mrx = torch.zeros(B, N).scatter_(1, pos[:, :1], 1.0)
    pol = torch.zeros(B, N).scatter_(1, pos[:, 1:], 1.0)
    with torch.no_grad():
        probs = torch.stack([actors[0](mrx)] + [actors[k](pol) for k in range(1, A)], dim=1)   # [B, A, N]
        value = critic(torch.cat([mrx] + [pol] * P, dim=-1)).squeeze(-1)                      # [B]
    out = {"N": N, "P": P, "H": H, "pos": pos.numpy().astype(np.int64), "probs": probs.numpy(), "value": value.numpy()}
    for k, a in enumerate(actors):
        for name, t in a.state_dict().items():
            out[f"actor{k}.{name}"] = t.numpy()
    for name, t in critic.state_dict().items():
        out[f"critic.{name}"] = t.numpy()
    path = os.path.join(HERE, "..", "tests", "golden", fname)
    np.savez_compressed(path, **out)
    print("wrote", os.path.abspath(path), sorted(k for k in out if "." in k)[:6], "...")


def main():
    spec = importlib.util.spec_from_file_location("ref_mappo_agent", os.path.join(REF, "src", "agent", "mappo_agent.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    capture(mod, 77, 14, 3, 8, 9, "mappo_networks_reference.npz")
    capture(mod, 78, 24, 4, 128, 12, "mappo_networks_reference_h128.npz")


if __name__ == "__main__":
    main()
