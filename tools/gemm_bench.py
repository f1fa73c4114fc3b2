#!/usr/bin/env python3
"""hermnet_node_gemm vs the library GEMMs (torch.bmm / addmm) on the node-level shapes of config 2."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from hermnet_amd import nodeops  # noqa: E402


def timeit(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3      # us


def main():
    dev = torch.device("cuda:0")
    T, B, H = 3, 3380, 128
    N = T * B
    gen = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=gen)
    shapes = [("x_proj.0   [N,H]x[TH,H]^T", N, T * H, H, 1), ("x_proj.2   T x [N,H]x[3H,H]^T", N, 3 * H, H, T),
              ("vec_proj   T x [3B,H]x[2H,H]^T", 3 * B, 2 * H, H, T), ("xvec.0     T x [B,2H]x[H,2H]^T", B, H, 2 * H, T),
              ("xvec.2     T x [B,H]x[3H,H]^T", B, 3 * H, H, T), ("bwd xvec.2 T x [B,3H]x[H,3H]^T", B, H, 3 * H, T),
              ("bwd x_proj.2 T x [N,3H]x[H,3H]^T", N, H, 3 * H, T), ("bwd x_proj.0 [N,TH]x[H,TH]^T", N, H, T * H, 1)]
    tot_mine = tot_lib = 0.0
    for name, M, Nn, K, bt in shapes:
        A, W = rnd(bt, M, K), rnd(bt, Nn, K) / K ** 0.5
        C = torch.empty(bt, M, Nn, device=dev)
        ref = torch.bmm(A, W.transpose(1, 2))
        nodeops.gemm(A, W, C, M, Nn, K, batch=bt, sA=M * K, sB=Nn * K, sC=M * Nn)
        err = float((C - ref).abs().max() / ref.abs().max())
        t_mine = timeit(lambda: nodeops.gemm(A, W, C, M, Nn, K, batch=bt, sA=M * K, sB=Nn * K, sC=M * Nn))
        Wt = W.transpose(1, 2).contiguous()
        t_lib = timeit(lambda: torch.bmm(A, Wt, out=C))
        fl = 2.0 * bt * M * Nn * K
        tot_mine += t_mine
        tot_lib += t_lib
        print("%-36s err %.1e  mine %6.1f us (%5.1f TF)  library %6.1f us (%5.1f TF)" %
              (name, err, t_mine, fl / t_mine / 1e6, t_lib, fl / t_lib / 1e6))
    print("sum: mine %.1f us, library %.1f us" % (tot_mine, tot_lib))
    # fused forms: prologue ScaledSiLU(+bias), epilogue bias / ScaledSiLU' / accumulate
    M, Nn, K, bt = 1000, 192, 128, 2
    A, W, pb, bias, E = rnd(bt, M, K), rnd(bt, Nn, K) / 11, rnd(bt, K), rnd(bt, Nn), rnd(bt, M, Nn)
    ss = lambda x: torch.nn.functional.silu(x) / 0.6
    C = torch.empty(bt, M, Nn, device=dev)
    nodeops.gemm(A, W, C, M, Nn, K, batch=bt, sA=M * K, sB=Nn * K, sC=M * Nn, prologue=1, pbias=pb, s_pbias=K,
                 epilogue=0, bias=bias, s_bias=Nn)
    ref = torch.bmm(ss(A + pb[:, None, :]), W.transpose(1, 2)) + bias[:, None, :]
    print("prologue ssilu + bias epilogue err %.1e" % float((C - ref).abs().max() / ref.abs().max()))
    nodeops.gemm(A, W, C, M, Nn, K, batch=bt, sA=M * K, sB=Nn * K, sC=M * Nn, epilogue=1, bias=bias, s_bias=Nn, E=E, lde=Nn,
                 sE=M * Nn)
    x = (E + bias[:, None, :]).double()
    s = torch.sigmoid(x)
    ref = torch.bmm(A, W.transpose(1, 2)) * (s * (1 + x * (1 - s)) / 0.6).float()
    print("ssilu' epilogue err %.1e" % float((C - ref).abs().max() / ref.abs().max()))
    C0 = rnd(bt, M, Nn)
    C = C0.clone()
    nodeops.gemm(A, W, C, M, Nn, K, batch=bt, sA=M * K, sB=Nn * K, sC=M * Nn, epilogue=2)
    ref = C0 + torch.bmm(A, W.transpose(1, 2))
    print("accumulate epilogue err %.1e" % float((C - ref).abs().max() / ref.abs().max()))


if __name__ == "__main__":
    main()
