#!/usr/bin/env python3
"""Wall-clock of the UNMODIFIED reference env.step (build container only; the reference never
travels to the GPU box).  Same loader as capture_goldens.py; random legal actions, single process —
the reference is single-threaded Python.

    python oracle/time_reference.py            # prints ms/step and agent-steps/s for C1 and C2 (one env)
"""
import os
import random
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("SY_REFERENCE", "/root/reference")
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "refstubs"))     # shape-only gymnasium / pettingzoo stand-ins
sys.path.insert(0, os.path.join(REF, "src"))
sys.dont_write_bytecode = True
import capture_goldens as cg  # noqa: E402  (weights, null logger, visualisation switches)
from environment.yard import CustomEnvironment  # noqa: E402  (the reference file, unmodified)


def new_env(n_police, money, nodes, edges, seed):
    np.random.seed(seed)
    random.seed(seed)
    weights = cg.make_weights(np.random.default_rng(1000 + seed))
    return CustomEnvironment(number_of_agents=n_police, agent_money=money, reward_weights=weights,
                             logger=cg.NullLogger(), epoch=0, graph_nodes=nodes, graph_edges=edges,
                             vis_configs=cg.VIS_OFF)


def time_config(n_police, money, nodes, edges, steps, seed=0):
    env = new_env(n_police, money, nodes, edges, seed)
    rng = random.Random(seed)
    t_reset0 = time.perf_counter()
    env.reset(episode=0)
    t_reset = time.perf_counter() - t_reset0
    done_steps, t_total = 0, 0.0
    while done_steps < steps:
        if not env.agents:
            env.reset(episode=done_steps)
        actions = {}
        for idx, ag in enumerate(env.possible_agents):
            moves = env.get_possible_moves(idx)
            actions[ag] = int(rng.choice(list(moves))) if len(moves) else -1
        t0 = time.perf_counter()
        env.step(actions)
        t_total += time.perf_counter() - t0
        done_steps += 1
    ms = 1e3 * t_total / steps
    return ms, (n_police + 1) / (ms * 1e-3), t_reset


if __name__ == "__main__":
    print("host:", os.popen("grep -m1 'model name' /proc/cpuinfo").read().strip().split(":")[-1].strip(),
          "| cores:", os.cpu_count(), "| single process")
    for name, cfg in (("C1 N=15 E=20 P=2 money=10", (2, 10, 15, 20, 200)),
                      ("C2 N=200 E=400 P=4 money=20 (one env)", (4, 20, 200, 400, 12))):
        ms, rate, t_reset = time_config(*cfg)
        print(f"{name}: {ms:.1f} ms/step, {rate:.1f} agent-steps/s, reset {t_reset:.2f} s")
