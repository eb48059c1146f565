#!/usr/bin/env python3
"""Golden vectors of the reference's return / advantage computation (build container only; TEST INFRASTRUCTURE).

Drives the UNMODIFIED `MappoAgent.ppo_update` (/root/reference/src/agent/mappo_agent.py:156-298, imported by
file path: it only needs torch) on prescribed (reward, done) sequences and records what its lines 247-258
compute:
  * `returns`     — taken from the argument the update hands to `nn.MSELoss()(new_values, returns)` (:264);
  * `advantages`  — taken from `torch.min(surr1, surr2)` (:291) of agent 0: the stored log-probabilities are the
                    ones the unchanged actor reproduces, so ratio == 1.0 exactly and surr1 == advantages;
  * `values`      — the critic's output on the stored global observations before the update (:245-246).
Nothing in the reference file is edited; the two call sites are observed through wrappers installed for the
duration of the call.  Writes tests/golden/ppo_returns_reference.npz.

    python oracle/capture_ppo_returns.py
"""
import importlib.util
import os

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SY_REFERENCE", "/root/reference")


def run_case(mod, rewards, dones, gamma, seed):
    L = len(rewards)
    n, n_agents = 6, 2
    torch.manual_seed(seed)
    agent = mod.MappoAgent(n_agents=n_agents, obs_size=n, global_obs_size=n * n_agents, action_size=n, hidden_size=8,
                           gamma=gamma, buffer_size=L)
    g = torch.Generator().manual_seed(seed + 1)
    obs = [[torch.rand(n, generator=g) for _ in range(n_agents)] for _ in range(L)]
    gobs = [torch.rand(n * n_agents, generator=g) for _ in range(L)]
    actions = [int(x) for x in torch.randint(0, n, (L,), generator=g)]
    with torch.no_grad():
        stacked0 = torch.stack([o[0] for o in obs])                      # what the update stacks for agent 0 (:270-275)
        lp0 = torch.distributions.Categorical(agent.policies[0](stacked0)).log_prob(torch.tensor(actions))
        values = agent.critic(torch.stack(gobs)).squeeze()
    for i in range(L):
        agent.store(obs[i], gobs[i], actions[i], float(rewards[i]), lp0[i], float(dones[i]))
    seen = {}
    real_mse, real_min = nn.MSELoss, torch.min

    class SpyMSE(real_mse):
        def forward(self, a, b):
            seen.setdefault("returns", b.detach().clone())
            return super().forward(a, b)

    def spy_min(a, *rest, **kw):
        if rest and isinstance(rest[0], torch.Tensor) and "advantages" not in seen:
            seen["advantages"] = a.detach().clone()
        return real_min(a, *rest, **kw)

    mod.nn.MSELoss, mod.torch.min = SpyMSE, spy_min
    try:
        agent.ppo_update()
    finally:
        mod.nn.MSELoss, mod.torch.min = real_mse, real_min
    assert "returns" in seen and "advantages" in seen and len(agent.memory) == 0
    return seen["returns"].numpy(), seen["advantages"].numpy(), values.numpy()


def main():
    spec = importlib.util.spec_from_file_location("ref_mappo_agent", os.path.join(REF, "src", "agent", "mappo_agent.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(20241005)
    out = {}
    cases = [("short_no_done", 7, 0.0, 0.99), ("episodes", 97, 0.08, 0.99), ("dense_dones", 64, 0.5, 0.9),
             ("long", 300, 0.03, 0.995), ("single_pair", 2, 0.0, 0.5), ("all_done", 16, 1.0, 0.99),
             ("undiscounted", 40, 0.1, 1.0)]
    for ci, (name, L, p_done, gamma) in enumerate(cases):
        rewards = rng.normal(size=L).astype(np.float64) * 2.0
        dones = (rng.random(L) < p_done).astype(np.float64)
        ret, adv, val = run_case(mod, rewards, dones, gamma, 500 + ci)
        out[f"{name}/reward"] = rewards
        out[f"{name}/done"] = dones.astype(np.uint8)
        out[f"{name}/gamma"] = np.float64(gamma)
        out[f"{name}/values"] = val.astype(np.float32)
        out[f"{name}/returns"] = ret.astype(np.float32)
        out[f"{name}/advantages"] = adv.astype(np.float32)
    out["case_names"] = np.array([c[0] for c in cases])
    out["source"] = np.array("agent/mappo_agent.py:247-258 via the unmodified ppo_update, torch " + torch.__version__)
    path = os.path.join(HERE, "..", "tests", "golden", "ppo_returns_reference.npz")
    np.savez_compressed(path, **out)
    print("wrote", os.path.abspath(path), len(cases), "cases")


if __name__ == "__main__":
    main()
