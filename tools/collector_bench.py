#!/usr/bin/env python3
"""SURVEY section 8f rows on the GPU: the on-device collector with a learned policy in the loop
(BASELINE configs[2] shape: N=200, P=4, B=4096; MAPPO MLP actors + central critic in PyTorch-ROCm,
env.step on the HIP engine).  Prints agent-steps/s for the policy-in-the-loop path; the fused
random-policy rollout (bench.py) is the engine-only number.
"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
import student_mechanism_design_amd as sy  # noqa: E402
from student_mechanism_design_amd.collector import RolloutCollector  # noqa: E402
from student_mechanism_design_amd.policies import MappoPolicy  # noqa: E402

B, N, P, T = 4096, 200, 4, 64
boards = sy.sample_board_pool(8, N, 400, seed=0)
env = sy.BatchedScotlandYardEnv(B, boards, P, 20, np.full(11, 0.5), seed=1234, reveal_interval=5)
env.reset(seed=1)
pol = MappoPolicy(N, P).to(env.device)
from student_mechanism_design_amd.collector import DeviceMaskedSampler  # noqa: E402
smp = DeviceMaskedSampler(env.device, seed=7)
from student_mechanism_design_amd.policies import DeviceMappoPolicy  # noqa: E402
fused = DeviceMappoPolicy(pol, seed=7)
# NOTE (round 2): `probs_fast` (index lookups + baddbmm) replayed from a HIP graph produced a few NaN probabilities per
# collect on torch 2.10 + ROCm 7.0 (eager runs of the same code never do; cause not found, the engine's tensors were
# verified clean), and torch.multinomial's device-side assert then aborts the process.  The torch forward is therefore
# timed with eager launches only; graph replay is timed with the HIP policy kernel, which is capture-safe and tested.
for use_graph, fn, name in ((False, pol.act, "reference-shaped forward, torch sampling"),
                            (False, pol.act_fast, "lookup + batched-matmul forward, torch sampling"),
                            (False, lambda obs: pol.act_device(obs, smp), "lookup + batched-matmul forward, HIP sampling kernel"),
                            (False, fused.act, "fused HIP policy kernel (actors + sampling + critic)"),
                            (True, fused.act, "fused HIP policy kernel (actors + sampling + critic)")):
    col = RolloutCollector(env, fn, frames_per_batch=T, use_graph=use_graph)
    col.collect()
    col.collect()          # (graph mode: capture + first replay)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        col.collect()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"collector + MAPPO policy ({name}), {'HIP graph replay' if use_graph else 'eager launches'}: "
          f"{reps * T * B * (P + 1) / dt / 1e6:.1f} M agent-steps/s ({dt / (reps * T) * 1e3:.3f} ms per batched step of {B} envs)")
# env.step alone, same loop without the policy (actions replayed)
rec = env.rollout(1)
act = rec["action"][0].contiguous()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(200):
    env.step(act)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(f"sy_env_step alone: {200 * B * (P + 1) / dt / 1e6:.1f} M agent-steps/s ({dt / 200 * 1e6:.1f} us per launch)")
