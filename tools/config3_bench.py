#!/usr/bin/env python3
"""BASELINE configs[2] end to end on one MI355X: 200-node board, 4 police, 4096 envs, policy forward in
PyTorch-ROCm (or inside the fused kernel), the rollout record and returns / advantages kept on the device, one PPO
(or DQN) update per iteration.  Reference loops: src/training/mappo_trainer.py:161-291 (MAPPO: collect, then
`ppo_update` mappo_agent.py:156-298) and src/training/gnn_trainer.py:194-291 (GNN DQN: collect, `agent.update`).

One iteration = collect T steps of B envs + returns/advantages (`sy_returns_advantages`, one launch) + one update over
the T*B*(P+1) transitions in minibatches.  Policies:
  gnn      GnnQPolicy (AntiSymmetricConv x2 + Linear on gather tables; greedy masked arg-max as one HIP kernel per step) driving the per-step collector;
           update = one TD(0) step of the DQN loss on the collected transitions (gnn_agent.py's update, batched)
  mappo    MappoPolicy (torch MLP actors + central critic) driving the per-step collector (torch forward, HIP sampling
           kernel, eager launches); update = clipped PPO surrogate + critic MSE (mappo_agent.py:260-293)
  kernel   the same networks as ONE HIP policy kernel per step (DeviceMappoPolicy), the T-step loop replayed as a HIP graph
  fused    the same MAPPO networks sampled INSIDE the fused rollout kernel (sy_env_set_policy); same update
Prints one JSON line per policy: agent-steps/s for collect alone and for the whole iteration, and ms per phase.

    python tools/config3_bench.py [--policies gnn,mappo,fused] [--iters 5] [--steps 64]
"""
import argparse
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
import student_mechanism_design_amd as sy  # noqa: E402
from student_mechanism_design_amd import collector as col  # noqa: E402
from student_mechanism_design_amd.policies import DeviceGnnPolicy, DeviceMappoPolicy, GcnTables, GnnQPolicy, MappoPolicy, ppo_loss  # noqa: E402


def timed(fn, dev):
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize(dev)
    return out, (time.perf_counter() - t0) * 1e3


def mappo_update(net, opt, rec, ret, adv_std, N, P, minibatch):
    """Clipped surrogate + critic MSE over the whole record in minibatches (mappo_agent.py:260-293)."""
    T, B, A = rec["action"].shape
    pos = rec["pos"].reshape(T * B, A)
    act = rec["action"].reshape(T * B, A).long()
    mask = rec["mask"][..., :N].reshape(T * B, A, N)
    old_lp = rec["log_prob"].reshape(T * B, A)
    ret_f, adv_f = ret.reshape(T * B, A), adv_std.reshape(T * B, A)
    valid = (act >= 0).float()
    perm = torch.randperm(T * B, device=pos.device)
    for i in range(0, T * B, minibatch):
        idx = perm[i:i + minibatch]
        obs = {"MrX_pos": pos[idx, 0], "Polices_pos": pos[idx, 1:]}
        value = net.value_fast(obs)
        pm = net.probs_fast(obs) * mask[idx].float()
        pm = pm / (pm.sum(-1, keepdim=True) + 1e-8)
        new_lp = torch.log(torch.gather(pm, -1, act[idx].clamp_min(0).unsqueeze(-1)).squeeze(-1) + 1e-8)
        al, cl = ppo_loss(new_lp * valid[idx], old_lp[idx] * valid[idx], adv_f[idx], value, ret_f[idx].sum(-1))
        opt.zero_grad()
        (al + 0.5 * cl).backward()
        opt.step()


def gnn_update(gnn, opt, rec, tabs, N, gamma, minibatch_steps):
    """One TD(0) step of the DQN loss per chunk of steps: Q(s, a_police0) vs r + gamma * max_a' Q(s') (gnn_agent.py update)."""
    T, B, A = rec["action"].shape
    done = (rec["terminated"] | rec["truncated"]).bool()
    for s0 in range(0, T - 1, minibatch_steps):
        s1 = min(T - 1, s0 + minibatch_steps)
        loss = 0.0
        for s in range(s0, s1):
            obs = {"agent_position": rec["pos"][s], "belief_map": rec["belief"][s][..., :N], "action_mask": rec["mask"][s][..., :N].bool()}
            nxt = {"agent_position": rec["pos"][s + 1], "belief_map": rec["belief"][s + 1][..., :N]}
            q = gnn.police(gnn.features(obs, N), tabs)                                    # [B, N]
            with torch.no_grad():
                qn = gnn.police(gnn.features(nxt, N), tabs).max(-1).values
            a1 = rec["action"][s][:, 1].long().clamp_min(0)                                # Police0's node
            target = rec["reward"][s][:, 1].float() + gamma * qn * (~done[s]).float()
            loss = loss + torch.nn.functional.mse_loss(q.gather(1, a1.unsqueeze(1)).squeeze(1), target)
        opt.zero_grad()
        loss.backward()
        opt.step()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--policies", default="gnn,mappo,kernel,fused")
    ap.add_argument("--envs", type=int, default=4096)
    ap.add_argument("--nodes", type=int, default=200)
    ap.add_argument("--police", type=int, default=4)
    ap.add_argument("--steps", type=int, default=64, help="env steps per iteration (T)")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--minibatch", type=int, default=32768)
    ap.add_argument("--gamma", type=float, default=0.99)
    args = ap.parse_args()
    B, N, P, T = args.envs, args.nodes, args.police, args.steps
    A = P + 1
    boards = sy.sample_board_pool(8, N, 2 * N, seed=0)
    for name in args.policies.split(","):
        env = sy.BatchedScotlandYardEnv(B, boards, P, 20, np.full(11, 0.5), seed=1234, reveal_interval=5)
        dev = env.device
        torch.manual_seed(0)
        if name == "gnn":
            gnn = GnnQPolicy(A).to(dev)
            gt = GcnTables(env.pool.boards, device=dev)
            a_hat_b = gt.for_envs(env.env_graph)                                                  # gather tables [B, N, K]
            opt = torch.optim.Adam(gnn.parameters(), lr=1e-3)
            dgnn = DeviceGnnPolicy(gnn, gt, env.env_graph)
            collector = col.RolloutCollector(env, dgnn.act, frames_per_batch=T, use_graph=True)
            collect = collector.collect
        else:
            net = MappoPolicy(N, P, hidden_size=64).to(dev)
            opt = torch.optim.Adam(net.parameters(), lr=3e-4)
            fused = DeviceMappoPolicy(net, seed=3)
            if name == "mappo":
                smp = col.DeviceMaskedSampler(dev, seed=7)
                collector = col.RolloutCollector(env, lambda obs: net.act_device(obs, smp), frames_per_batch=T)
                collect = collector.collect
            elif name == "kernel":
                collector = col.RolloutCollector(env, fused.act, frames_per_batch=T, use_graph=True)
                collect = collector.collect
            else:
                env.set_policy(fused)
                out = env.alloc_rollout(T)
                collect = lambda: env.rollout(T, out=out)     # noqa: E731
        for _ in range(3):            # warm-up (graph capture on the second call)
            rec = collect()
        t_c = t_r = t_u = 0.0
        for _ in range(args.iters):
            rec, ms = timed(collect, dev)
            t_c += ms
            (ret, adv), ms = timed(lambda: col.device_returns(rec["reward"], rec["terminated"], args.gamma, done_b=rec["truncated"],
                                                              values=rec.get("value")), dev)
            t_r += ms
            if name == "gnn":
                _, ms = timed(lambda: (gnn_update(gnn, opt, rec, a_hat_b, N, args.gamma, 8), dgnn.refresh()), dev)
            else:
                def upd():
                    adv_std = col.standardized_advantages(ret, torch.zeros_like(ret)) if rec.get("value") is None else \
                        col.standardized_advantages(ret, rec["value"].unsqueeze(-1).expand_as(ret))
                    mappo_update(net, opt, rec, ret, adv_std, N, P, args.minibatch)
                    fused.refresh()
                _, ms = timed(upd, dev)
            t_u += ms
        env.check_status()
        n = args.iters * T * B * A
        line = {"config": "BASELINE configs[2]: N=%d, P=%d, B=%d, T=%d" % (N, P, B, T), "policy": name,
                "collect_agent_steps_per_s": n / (t_c * 1e-3), "iteration_agent_steps_per_s": n / ((t_c + t_r + t_u) * 1e-3),
                "collect_ms": t_c / args.iters, "returns_ms": t_r / args.iters, "update_ms": t_u / args.iters,
                "episodes_per_iteration": int((rec["terminated"] | rec["truncated"]).sum())}
        print(json.dumps(line), flush=True)
        env.close()
        del env


if __name__ == "__main__":
    main()
