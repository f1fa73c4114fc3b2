#!/usr/bin/env python3
"""Phase anatomy of the node chain kernels from in-kernel clock stamps (diagnostic library built with -DHN_STAMPS):
    HERMNET_LIB_PATH=hermnet_amd/csrc/variants/lib_stamps.so python tools/chain_stamps.py [atoms]
Prints, per kernel, the mean / max shader cycles between consecutive stamps of wave 0 over all workgroups."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hermnet_amd import _lib, nodeops  # noqa: E402
from hermnet_amd.layer import LayerWeights  # noqa: E402
from hermnet_amd.relations import RelationalGraph  # noqa: E402
from hermnet_amd.rmnet import PaiNNModule  # noqa: E402

NAMES = {"node_pre_fwd": ["LN", "barrier", "GEMM1", "preload+barrier", "epi1", "barrier", "GEMM2", "epi2"],
         "node_update_fwd": ["tile0+barrier", "GEMM V0", "epi V0", "st+barrier", "GEMM V1", "epi V1", "st+barrier", "GEMM V2",
                             "epi V2", "norm+barriers", "GEMM X", "epiX+barrier+prefetch", "GEMM Q", "epi Q"]}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    H, T = 128, 3
    dev = torch.device("cuda")
    torch.manual_seed(0)
    mods = [PaiNNModule(hidden_channels=H, num_rbf=16).to(dev) for _ in range(T)]
    w = LayerWeights(mods).refresh()
    zs = [13, 28, 29]
    z = torch.tensor(zs, device=dev)[torch.randint(0, T, (n,), device=dev)]
    g = RelationalGraph.build(z, torch.stack([torch.randint(0, n, (4 * n,), device=dev), torch.randint(0, n, (4 * n,), device=dev)]), zs)
    N = g.N
    x, x1, vec1 = torch.randn(N, H, device=dev), torch.randn(N, H, device=dev), torch.randn(N, 3, H, device=dev)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    lib.hermnet_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
    for name, fn, blocks in (("node_pre_fwd", lambda: nodeops.node_pre_fwd(x, w, T), ((N + 63) // 64) * T),
                             ("node_update_fwd", lambda: nodeops.node_update_fwd(x1, vec1, w, g), (N + 31) // 32)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        buf = np.zeros(8192 * 16, dtype=np.uint64)
        assert lib.hermnet_debug_stamps(buf.ctypes.data, buf.size) == 0
        st = buf.reshape(8192, 16)[:min(blocks, 8192)].astype(np.int64)
        ns = len(NAMES[name]) + 1
        d = np.diff(st[:, :ns], axis=1)
        tot = st[:, ns - 1] - st[:, 0]
        span = st[:, :ns].max() - st[:, :ns].min()
        print("%s: %d workgroups, wave-0 life mean %d max %d cycles, whole grid %d cycles" % (name, len(st), tot.mean(), tot.max(), span))
        hw = st[:, 15].astype(np.uint64)
        xcc, raw = (hw >> np.uint64(32)) & np.uint64(0xf), hw & np.uint64(0xffffffff)
        cu = (raw >> np.uint64(8)) & np.uint64(0xf)
        sh = (raw >> np.uint64(12)) & np.uint64(0x1)
        se = (raw >> np.uint64(13)) & np.uint64(0x7)
        key = ((xcc * np.uint64(8) + se) * np.uint64(2) + sh) * np.uint64(16) + cu
        uniq, cnt = np.unique(key, return_counts=True)
        print("   placement: %d distinct CUs hold the %d workgroups; workgroups per used CU: %s" % (
            len(uniq), len(st), dict(zip(*np.unique(cnt, return_counts=True)))))
        for k, nm in enumerate(NAMES[name]):
            print("   %-24s mean %7d  max %7d" % (nm, d[:, k].mean(), d[:, k].max()))
        # co-resident pairs: do the two workgroups of a CU walk through their matrix phases in lockstep?
        gemm = [k for k, nm in enumerate(NAMES[name]) if nm.startswith("GEMM")]
        slot = (raw & np.uint64(0xf)).astype(np.int64)
        offs, share, slots = [], [], []
        for u in uniq[cnt == 2]:
            a_, b_ = np.nonzero(key == u)[0]
            if st[a_, 0] > st[b_, 0]:
                a_, b_ = b_, a_
            offs.append(st[b_, 0] - st[a_, 0])
            ia = [(st[a_, k], st[a_, k + 1]) for k in gemm]
            ib = [(st[b_, k], st[b_, k + 1]) for k in gemm]
            both = sum(max(0, min(x1_, y1_) - max(x0_, y0_)) for x0_, x1_ in ia for y0_, y1_ in ib)
            share.append(both / max(1, sum(x1_ - x0_ for x0_, x1_ in ia)))
            slots.append((int(slot[a_]), int(slot[b_])))
        if offs:
            print("   CUs with two workgroups: start offset mean %d max %d cycles; share of the first one's matrix phases during "
                  "which the second is in a matrix phase too: mean %.2f; wave slots of wave 0 (first, second): %s" % (
                      np.mean(offs), np.max(offs), np.mean(share), dict(zip(*np.unique(np.array(slots), axis=0, return_counts=True))) if False else
                      sorted(set(slots))[:6]))


if __name__ == "__main__":
    main()
