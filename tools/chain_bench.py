#!/usr/bin/env python3
"""Times the four node chain kernels (csrc/node_chain.hip) on the row counts of BASELINE configs[1] (10,000 atoms, 3
relations, H = 128) or any other size:   python tools/chain_bench.py [atoms] [H] [T] [reps]
Prints microseconds per launch (HIP events around `reps` back-to-back launches) and the fp32 MFMA utilisation."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from hermnet_amd import nodeops  # noqa: E402
from hermnet_amd.layer import LayerWeights  # noqa: E402
from hermnet_amd.relations import RelationalGraph  # noqa: E402
from hermnet_amd.rmnet import PaiNNModule  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _opts import apply_option_env  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    H = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 50
    dev = torch.device("cuda")
    apply_option_env()
    torch.manual_seed(0)
    mods = [PaiNNModule(hidden_channels=H, num_rbf=16).to(dev) for _ in range(T)]
    w = LayerWeights(mods).refresh()
    zs = [13, 28, 29, 79, 47, 46][:T]
    z = torch.tensor(zs, device=dev)[torch.randint(0, T, (n,), device=dev)]
    src = torch.randint(0, n, (4 * n,), device=dev)
    tgt = torch.randint(0, n, (4 * n,), device=dev)
    g = RelationalGraph.build(z, torch.stack([src, tgt]), zs)
    N = g.N
    x, x1, vec1 = torch.randn(N, H, device=dev), torch.randn(N, H, device=dev), torch.randn(N, 3, H, device=dev)
    gxh, gxo, gvo = torch.randn(T, N, 3 * H, device=dev), torch.randn(N, H, device=dev), torch.randn(N, 3, H, device=dev)
    hb, xh, mean, rstd = nodeops.node_pre_fwd(x, w, T)
    xo, vo, vp, h2b, q23, nrm = nodeops.node_update_fwd(x1, vec1, w, g)
    cases = {
        "node_pre_fwd": (lambda: nodeops.node_pre_fwd(x, w, T), 2 * N * T * 4 * H * H),
        "node_pre_bwd": (lambda: nodeops.node_pre_bwd(gxh, hb, x, mean, rstd, w), 2 * N * T * 4 * H * H),
        "node_update_fwd": (lambda: nodeops.node_update_fwd(x1, vec1, w, g), 2 * N * 11 * H * H),
        "node_update_bwd": (lambda: nodeops.node_update_bwd(gxo, gvo, vp, h2b, q23, nrm, w, g), 2 * N * 11 * H * H),
    }
    if nodeops.fused_boundary_supported(g, H, w, w):        # round 5: the layer boundary as one launch each way (16-row tiles)
        gv_parts, gx1u, gvec1u = torch.randn(T, N, 3, H, device=dev), torch.randn(N, H, device=dev), torch.randn(N, 3, H, device=dev)
        bx, bv = torch.empty(N, H, device=dev), torch.empty(N, 3, H, device=dev)
        gn_parts = nodeops.node_pre_bwd16(gxh, hb, w)
        pend_parts = nodeops.PendingGrads(bx, bv, gn_parts, gv_parts, xo, mean, rstd, gx1u, gvec1u, 0)
        pend_chain = nodeops.PendingGrads(bx, bv, None, gv_parts, xo, mean, rstd, gx1u, gvec1u, 0, chain=(gxh, hb, w.w2tf16, w.w1tf16))
        cases.update({
            "pre_fwd16": (lambda: nodeops.node_pre_fwd16(x, w, T), 2 * N * T * 4 * H * H),
            "pre_bwd16": (lambda: nodeops.node_pre_bwd16(gxh, hb, w), 2 * N * T * 4 * H * H),
            "update_bwd(parts)": (lambda: nodeops.node_update_bwd(bx, bv, vp, h2b, q23, nrm, w, g, pending=pend_parts), 2 * N * 11 * H * H),
            "update_pre_fwd": (lambda: nodeops.node_update_pre_fwd(x1, vec1, w, g, w), 2 * N * (11 + 4 * T) * H * H),
            "pre_update_bwd": (lambda: nodeops.node_update_bwd(bx, bv, vp, h2b, q23, nrm, w, g, pending=pend_chain), 2 * N * (11 + 4 * T) * H * H),
        })
    # CHAIN_BENCH_THRASH=MB: between two timed launches a copy of that many MB runs (every launch then meets the caches the way
    # it does inside a step -- weights and inputs evicted from L2 by the message kernels' traffic -- instead of 200 back-to-back
    # launches on hot buffers); launches are timed one by one
    thrash = int(os.environ.get("CHAIN_BENCH_THRASH", "0"))
    if thrash:
        big_a = torch.empty(thrash * (1 << 20) // 4, device=dev)
        big_b = torch.empty_like(big_a)
    print("rows %d (atoms %d), H %d, T %d%s" % (N, n, H, T, ", %d MB copy between launches" % thrash if thrash else ""))
    for _ in range(3):                    # every case a few times before any is timed (allocator growth, code objects:
        for fn, _f in cases.values():     # a one-time stall otherwise lands in the first case's interval)
            fn()
    torch.cuda.synchronize()
    for name, (fn, flop) in cases.items():
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        if thrash:
            evs = []
            for _ in range(min(reps, 40)):
                big_b.copy_(big_a)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                fn()
                b.record()
                evs.append((a, b))
            torch.cuda.synchronize()
            ts = sorted(a.elapsed_time(b) for a, b in evs)
            us = ts[len(ts) // 2] * 1e3
        else:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                fn()
            b.record()
            torch.cuda.synchronize()
            us = a.elapsed_time(b) / reps * 1e3
        print("%-16s %8.1f us   %6.1f TFLOP/s  (%.2f of 155)" % (name, us, flop / us / 1e6, flop / us / 1e6 / 155))


if __name__ == "__main__":
    main()
