#!/usr/bin/env python3
"""Weight-gradient product of the training path's rbf_proj (g^T a: [3H, K] x [K, R], K ~ 1e5 edges of a relation):
library GEMM vs batched K-chunks + sum (trainops.TallLinear), chunk sizes.  python tools/tall_gemm_bench.py"""
import time
import torch
dev = torch.device("cuda")
K, O, R = 128885, 384, 128
g, a = torch.randn(K, O, device=dev), torch.randn(K, R, device=dev)


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


ref = (g.double().t() @ a.double())
print("plain g.t() @ a            %8.1f us" % timed(lambda: g.t() @ a))
for C in (512, 1024, 2048, 4096, 8192, 16384):
    n = K // C

    def f():
        out = g[n * C:].t() @ a[n * C:]
        return out + torch.bmm(g[:n * C].view(n, C, -1).transpose(1, 2), a[:n * C].view(n, C, -1)).sum(0)
    err = float((f().double() - ref).abs().max() / ref.abs().max())
    print("chunks of %5d (%3d batches) %8.1f us   rel err %.1e" % (C, n, timed(f), err))
for C in (2048, 4096):
    n = K // C

    def f2():      # [n, R, C] x [n, C, O] -> the transposed product (other operand order)
        out = a[n * C:].t() @ g[n * C:]
        return (out + torch.bmm(a[:n * C].view(n, C, -1).transpose(1, 2), g[:n * C].view(n, C, -1)).sum(0)).t()
    print("transposed order, %5d      %8.1f us" % (C, timed(f2)))
