#!/usr/bin/env python3
"""Fused rollout with the MAPPO actors sampling inside the kernel (sy_env_set_policy) vs the uniform-random policy:
BASELINE configs[1]/[2] shape (4 police), the configs[3] shard (6 police, env ids of rank 3) and hidden 128.
Prints agent-steps/s and the kernel instance for each."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, '.')
import student_mechanism_design_amd as sy  # noqa: E402
from student_mechanism_design_amd.policies import DeviceMappoPolicy, MappoPolicy  # noqa: E402

B, N, T = 4096, 200, 256
boards = sy.sample_board_pool(8, N, 400, seed=0)
for P, H, offset, wpb in ((4, 64, 0, 0), (4, 128, 0, 0), (6, 64, 3 * B, 0), (6, 128, 3 * B, 12)):
    env = sy.BatchedScotlandYardEnv(B, boards, P, 20, np.full(11, 0.5), seed=1234, reveal_interval=5, env_id_offset=offset,
                                    waves_per_block=wpb)
    env.reset(seed=1)
    net = MappoPolicy(N, P, hidden_size=H).to(env.device)
    fused = DeviceMappoPolicy(net, seed=3)
    for name, policy in (("uniform-random policy", None), ("MAPPO actors in the kernel", fused)):
        env.set_policy(policy)
        out = env.alloc_rollout(T)
        for _ in range(3):
            env.rollout(T, out=out)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            env.rollout(T, out=out)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        env.check_status()
        print(f"P={P} hidden={H} waves_per_block={env.waves_per_block} {name}: {reps * T * B * (P + 1) / dt / 1e9:.2f} G agent-steps/s "
              f"({dt / reps * 1e3:.3f} ms per launch of {T} steps)  {env.rollout_kernel_name()}", flush=True)
        del out
    env.close()
