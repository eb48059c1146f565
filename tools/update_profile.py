#!/usr/bin/env python3
"""Where one MappoUpdater.update spends its GPU time (torch.profiler, top kernels).  Run through gpurun."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import student_mechanism_design_amd as sy  # noqa: E402
from student_mechanism_design_amd import collector as col, policies as pol  # noqa: E402
from student_mechanism_design_amd.update import MappoUpdater  # noqa: E402

N, P, B, T = 200, 4, 4096, 64
boards = sy.sample_board_pool(8, N, 400, seed=0)
env = sy.BatchedScotlandYardEnv(B, boards, P, 20, np.full(11, 0.5), seed=1, reveal_interval=5)
net = pol.MappoPolicy(N, P, hidden_size=64).to(env.device)
fused = pol.DeviceMappoPolicy(net, seed=3)
env.set_policy(fused)
rec = env.rollout(T)
ret, _ = col.device_returns(rec["reward"], rec["terminated"], 0.99, done_b=rec["truncated"])
up = MappoUpdater(net, env.ell, env.env_graph, minibatch=int(sys.argv[1]) if len(sys.argv) > 1 else 32768, use_graph=False)
for _ in range(2):
    up.update(rec, ret)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(3):
    up.update(rec, ret)
torch.cuda.synchronize()
print("update ms", (time.perf_counter() - t0) / 3 * 1e3)
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA]) as prof:
    up.update(rec, ret)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=70))
