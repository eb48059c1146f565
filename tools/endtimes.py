#!/usr/bin/env python3
"""Launch-level load balance of the fused rollout: per move wave start / end times.

Needs an engine built with -DSY_ENDTIMES3 (the pipeline's move waves leave both times, 100 MHz ticks, in the padding
bytes of the last recorded mask row; --rollout2: -DSY_ENDTIMES, padding words of the last record row):
    SY_HIPCC_FLAGS=-DSY_ENDTIMES python -m student_mechanism_design_amd.build   # or build to another path
    SY_ENGINE_LIB=<that .so> python tools/endtimes.py [T]
"""
import sys

import numpy as np
import torch

sys.path.insert(0, '.')
import student_mechanism_design_amd as sy  # noqa: E402

T = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 256
boards = sy.sample_board_pool(8, 200, 400, seed=0)
env = sy.BatchedScotlandYardEnv(4096, boards, 4, 20, np.full(11, 0.5), seed=1234, reveal_interval=5)
env.reset(seed=1)
out = env.alloc_rollout(T)
W = env.waves_per_block
for it in range(3):
    env.rollout(T, out=out)
    torch.cuda.synchronize()
    if "--rollout2" in sys.argv:       # round 1's kernel (-DSY_ENDTIMES -DSY_NO_PIPELINE): padding words of the last record row
        rec = out["record"][T - 1].cpu().numpy().astype(np.int64) & 0xffffffff   # [B, RW]
        st, en = rec[:, -2], rec[:, -1]
    else:                              # the pipeline (-DSY_ENDTIMES3): padding bytes of the last mask row of the last agent
        pad = out["mask"][T - 1, :, -1, 200:208].contiguous().cpu().numpy().view(np.uint32).astype(np.int64)   # [B, 2]
        st, en = pad[:, 0], pad[:, 1]
        cnt = out["mask"][T - 1, :, -2, 200:208].contiguous().cpu().numpy().view(np.uint32).astype(np.int64)   # [B, 2] restarts, conflicts
        if it == 2:
            w = slice(0, None, 2)          # one entry per move wave (both halves carry the same numbers)
            r = (en - st)[w] / 100.0
            X = np.stack([np.ones_like(r), cnt[w, 0], cnt[w, 1]], 1).astype(np.float64)
            coef, *_ = np.linalg.lstsq(X, r, rcond=None)
            res = r - X @ coef
            print(f"    per wave: restart steps mean {cnt[w,0].mean():.1f} sd {cnt[w,0].std():.1f}, conflict steps mean {cnt[w,1].mean():.1f} sd {cnt[w,1].std():.1f}")
            print(f"    run time = {coef[0]:.1f} us + {coef[1]*1e3:.0f} ns per restart step + {coef[2]*1e3:.0f} ns per conflict step; residual sd {res.std():.1f} us (of {r.std():.1f})")
            board = np.arange(r.shape[0]) // (r.shape[0] // 8)
            print("    mean run by board:", np.array([r[board == g].mean() for g in range(8)]).round(1))
            xcd = (np.arange(r.shape[0]) // 8) % 8
            print("    mean run by block % 8 (XCD):", np.array([r[xcd == g].mean() for g in range(8)]).round(1))
    t0 = st.min()
    run = (en - st) / 100.0
    end = (en - t0) / 100.0
    print(f"T={T} launch {it}: wave run mean {run.mean():.1f} min {run.min():.1f} max {run.max():.1f} std {run.std():.1f} us")
    print("    end percentiles 0/1/10/50/90/99/100:", np.percentile(end, [0, 1, 10, 50, 90, 99, 100]).round(1))
    blk = end.reshape(-1, W).max(1)
    print(f"    block end: mean {blk.mean():.1f} min {blk.min():.1f} max {blk.max():.1f}")
    print("    mean end by move wave of the block:", end[0::2].reshape(-1, W // 2).mean(0).round(1))
