"""PPO update timing on the configs[2] shape: the torch form vs the fused HIP gradient kernel (eager and graph-replayed).
Run through gpurun:  python tools/update_bench.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import student_mechanism_design_amd as sy  # noqa: E402
from student_mechanism_design_amd import collector as col, policies as pol  # noqa: E402
from student_mechanism_design_amd.update import MappoUpdater  # noqa: E402

dev = torch.device("cuda", 0)
N, P, B, T = 200, 4, 4096, 64
boards = sy.sample_board_pool(8, N, 400, seed=0)
env = sy.BatchedScotlandYardEnv(B, boards, P, 20, np.full(11, 0.5), seed=1, reveal_interval=5, device=dev)
torch.manual_seed(0)
net = pol.MappoPolicy(N, P, hidden_size=64).to(dev)
fused = pol.DeviceMappoPolicy(net, seed=3)
env.set_policy(fused)
out = env.alloc_rollout(T)
rec = env.rollout(T, out=out)
ret, _ = col.device_returns(rec["reward"], rec["terminated"], 0.99, done_b=rec["truncated"])


def timed(fn, reps=7):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize(dev)
        ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))


only = sys.argv[1] if len(sys.argv) > 1 else ""
for name, kw in (("torch form, 8 minibatches, graph", dict(fused=False, use_graph=True, minibatch=32768)),
                 ("fused kernel, 8 minibatches, eager", dict(fused=True, use_graph=False, minibatch=32768)),
                 ("fused kernel, 8 minibatches, graph", dict(fused=True, use_graph=True, minibatch=32768)),
                 ("fused kernel, one full-batch step", dict(fused=True, use_graph=False, minibatch=T * B)),
                 ("torch form, one full-batch step", dict(fused=False, use_graph=False, minibatch=T * B))):
    if only and only not in name:
        continue
    up = MappoUpdater(net, env.ell, env.env_graph, **kw)
    ms = timed(lambda: up.update(rec, ret))
    al, cl = (float(x) for x in up.last_losses)
    print("%-40s %8.3f ms per update of %d agent transitions   (losses %.5f %.1f)" % (name, ms, T * B * (P + 1), al, cl), flush=True)
