#!/usr/bin/env python3
"""MAPPO training on the device-resident engine (the loop of src/training/mappo_trainer.py, batched):

  collect   T fused env-steps of B envs with the CURRENT actors sampling inside the rollout kernel
            (env.set_policy -> sy_env_set_policy; log-probs recorded by the kernel)
  returns   discounted returns per agent in one launch (sy_returns_advantages), advantages standardised as in
            MappoAgent.ppo_update
  update    update.MappoUpdater: critic MSE + clipped surrogate over minibatches of the recorded (observation, action)
            pairs (torch autograd on the same MappoPolicy module, no host sync, each step one HIP graph), then
            DeviceMappoPolicy.refresh()

Everything stays in HBM; the only host work is the Python loop.  Needs an MI355X and the built engine.

    python examples/ppo_train.py --iters 30
"""
import argparse
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
import student_mechanism_design_amd as sy  # noqa: E402
from student_mechanism_design_amd import collector as col  # noqa: E402
from student_mechanism_design_amd.metrics import rollout_metrics  # noqa: E402
from student_mechanism_design_amd.policies import DeviceMappoPolicy, MappoPolicy  # noqa: E402
from student_mechanism_design_amd.update import MappoUpdater  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--nodes", type=int, default=200)
    ap.add_argument("--police", type=int, default=4)
    ap.add_argument("--steps", type=int, default=64, help="env steps per rollout (T)")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--minibatch", type=int, default=32768)
    ap.add_argument("--lr", type=float, default=3e-3)
    ap.add_argument("--gamma", type=float, default=0.99)
    args = ap.parse_args()
    B, N, P, T = args.envs, args.nodes, args.police, args.steps
    A = P + 1
    boards = sy.sample_board_pool(8, N, 2 * N, seed=0)
    env = sy.BatchedScotlandYardEnv(B, boards, P, 20, np.full(11, 0.5), seed=0, reveal_interval=5)
    env.reset(seed=0)
    dev = env.device
    net = MappoPolicy(N, P, hidden_size=64).to(dev)
    fused = DeviceMappoPolicy(net, seed=0)
    env.set_policy(fused)
    up = MappoUpdater(net, env.ell, env.env_graph, lr=args.lr, minibatch=args.minibatch)
    out = env.alloc_rollout(T)
    t_collect = t_update = 0.0
    for it in range(args.iters):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rec = env.rollout(T, out=out)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        # returns as ONE launch on the packed record (mappo_agent.py:247-254), then a sync-free minibatch pass
        # (critic MSE + clipped surrogate, :256-293) and the refresh of the weights the kernel reads
        ret, _ = col.device_returns(rec["reward"], rec["terminated"], args.gamma, done_b=rec["truncated"])
        al, cl = up.update(rec, ret)
        fused.refresh()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        t_collect += t1 - t0
        t_update += t2 - t1
        env.check_status()
        m = rollout_metrics(rec, N)                                                # (reporting only: host reads below)
        reward = rec["reward"]
        print(f"iter {it:3d}  episodes {int(m['num_episodes']):6d}  MrX win rate {float(m['win_rate']):.3f}  "
              f"mean length {float(m['mean_episode_length']):5.1f}  mean reward MrX {float(reward[..., 0].mean()):+.3f} "
              f"police {float(reward[..., 1:].mean()):+.3f}  actor loss {float(al):+.4f}  critic loss {float(cl):.3f}")
    n = args.iters * T * B * A
    print(f"collect: {n / t_collect / 1e9:.2f} G agent-steps/s ({t_collect / args.iters * 1e3:.1f} ms per rollout); "
          f"update: {t_update / args.iters * 1e3:.1f} ms per iteration")


if __name__ == "__main__":
    main()
