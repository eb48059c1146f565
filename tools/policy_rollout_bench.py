#!/usr/bin/env python3
"""Fused rollout with the MAPPO actors sampling inside the kernel (sy_env_set_policy) vs the uniform-random
policy, BASELINE configs[1] shape.  Prints agent-steps/s for both."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
import student_mechanism_design_amd as sy  # noqa: E402
from student_mechanism_design_amd.policies import DeviceMappoPolicy, MappoPolicy  # noqa: E402

B, N, P, T = 4096, 200, 4, 256
boards = sy.sample_board_pool(8, N, 400, seed=0)
env = sy.BatchedScotlandYardEnv(B, boards, P, 20, np.full(11, 0.5), seed=1234, reveal_interval=5)
env.reset(seed=1)
net = MappoPolicy(N, P, hidden_size=64).to(env.device)
fused = DeviceMappoPolicy(net, seed=3)
out = env.alloc_rollout(T)
for name, policy in (("uniform-random policy", None), ("MAPPO actors in the kernel", fused), ("uniform-random policy", None)):
    env.set_policy(policy)
    for _ in range(3):
        env.rollout(T, out=out)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 10
    for _ in range(reps):
        env.rollout(T, out=out)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{name}: {reps * T * B * (P + 1) / dt / 1e9:.2f} G agent-steps/s ({dt / reps * 1e3:.3f} ms per launch of {T} steps)")
