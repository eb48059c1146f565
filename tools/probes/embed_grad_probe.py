#!/usr/bin/env python3
"""Which torch form of the first-layer weight gradient (a segment sum of gradient rows by node) is fastest on this GPU."""
import time
import torch
dev = "cuda"
R, N, H, P = 262144, 200, 256, 4
for R in (32768, 262144):
    g = torch.randn(R, H, device=dev)
    idx = torch.randint(0, N, (R, P), device=dev)
    oh = torch.zeros(R, N, device=dev).scatter_(1, idx, 1.0)

    def t(fn, n=10):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    ref = oh.t() @ g
    out = torch.zeros(N, H, device=dev)
    def ia():
        out.zero_()
        for j in range(P):
            out.index_add_(0, idx[:, j], g)
    print(R, "dense mm      %.3f ms" % t(lambda: oh.t() @ g))
    print(R, "index_add_ x4 %.3f ms" % t(ia), float((out - ref).abs().max()))
    flat_idx = idx.reshape(-1)
    gg = g.unsqueeze(1).expand(R, P, H).reshape(-1, H)
    def ia2():
        out.zero_()
        out.index_add_(0, flat_idx, gg)
    print(R, "index_add_ x1 %.3f ms" % t(ia2))
    # two-level: chunked dense (split K over 64 chunks as a bmm, then sum)
    C = 64
    def splitk():
        return torch.bmm(oh.view(C, R // C, N).transpose(1, 2), g.view(C, R // C, H)).sum(0)
    print(R, "split-K bmm   %.3f ms" % t(splitk), float((splitk() - ref).abs().max()))
    oh16, g16 = oh.half(), g.half()
    print(R, "fp16 mm       %.3f ms" % t(lambda: oh16.t() @ g16))
