#!/usr/bin/env python3
"""Measured HBM rates of the box with plain device kernels (context for the 8 TB/s spec peak the
roofline fraction is quoted against): write-only fill, copy (read + write), read-only reduction."""
import torch

dev = torch.device("cuda", 0)
n = 1 << 30                       # 4 GiB of float32 per buffer
a = torch.empty(n, dtype=torch.float32, device=dev)
b = torch.empty(n, dtype=torch.float32, device=dev)


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(reps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / reps * 1e-3


t = timed(lambda: a.fill_(1.0))
print(f"write-only fill : {4 * n / t / 1e12:.2f} TB/s")
t = timed(lambda: b.copy_(a))
print(f"copy (rd + wr)  : {8 * n / t / 1e12:.2f} TB/s moved")
t = timed(lambda: a.sum())
print(f"read-only sum   : {4 * n / t / 1e12:.2f} TB/s")
