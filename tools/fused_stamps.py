#!/usr/bin/env python3
"""Phase anatomy of the fused layer-boundary kernels (csrc/node_chain16.hip) from in-kernel clock stamps:
    bash tools/build_chain_variant.sh stamps -DHN_STAMPS
    HERMNET_LIB_PATH=hermnet_amd/csrc/variants/libhermnet_stamps.so python tools/fused_stamps.py [rows]
Per kernel: mean shader cycles between consecutive stamps of wave 0 over all workgroups, the workgroups' lives, the span
of the whole grid, and how the workgroups were placed on the CUs (a diagnostic build: read shares, not lengths)."""
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

NAMES = {"update_pre_fwd": ["vp d0 (+load)", "vp d1", "vp d2", "norm+barriers", "GEMM h2", "epi h2+barrier+req", "GEMM pqr",
                            "epi update (stores)", "barrier", "LayerNorm+barrier", "proj t0", "proj t1", "proj t2"],
         "pre_update_bwd": ["load gxh t0+barrier", "proj bwd t0", "proj bwd t1", "proj bwd t2", None, "sums+LN bwd+gvec sums",
                            "gq staging+barrier", "GEMM ga2", "gh2+requests+barrier", "GEMM gxin", "gvec d0", "gvec d1", "gvec d2"]}
SLOTS = {"update_pre_fwd": list(range(0, 14)), "pre_update_bwd": [0, 1, 2, 3, 4, 4, 6, 7, 8, 9, 10, 11, 12, 13]}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
    H, T = 128, 3
    dev = torch.device("cuda")
    torch.manual_seed(0)
    mk = lambda: LayerWeights([PaiNNModule(hidden_channels=H, num_rbf=16).to(dev) for _ in range(T)]).refresh()
    w, wn = mk(), mk()
    zs = [13, 28, 29]
    z = torch.tensor(zs, device=dev)[torch.randint(0, T, (n,), device=dev)]
    g = RelationalGraph.build(z, torch.stack([torch.randint(0, n, (4 * n,), device=dev), torch.randint(0, n, (4 * n,), device=dev)]), zs)
    N = g.N
    assert nodeops.fused_boundary_supported(g, H, w, wn), "this row count does not take 16-row tiles"
    r = lambda *s: torch.randn(*s, device=dev)
    x1, vec1 = r(N, H), r(N, 3, H)
    out = nodeops.node_update_pre_fwd(x1, vec1, w, g, wn)
    xo, vo, vp, h2b, q23, nrm, (hb, xh, mean, rstd) = out
    gxh, gv_parts, gx1u, gvec1u = r(T, N, 3 * H) * 0.3, r(T, N, 3, H), r(N, H), r(N, 3, H)

    def bwd():
        bx, bv = torch.empty(N, H, device=dev), torch.empty(N, 3, H, device=dev)
        pend = nodeops.PendingGrads(bx, bv, None, gv_parts, xo, mean, rstd, gx1u, gvec1u, 0, chain=(gxh, hb, wn.w2tf16, wn.w1tf16))
        return nodeops.node_update_bwd(bx, bv, vp, h2b, q23, nrm, w, g, pending=pend)

    lib = ctypes.CDLL(_lib.LIB_PATH)
    lib.hermnet_debug_stamps16.argtypes = [ctypes.c_void_p, ctypes.c_int]
    blocks = (N + 15) // 16
    for name, fn in (("update_pre_fwd", lambda: nodeops.node_update_pre_fwd(x1, vec1, w, g, wn)), ("pre_update_bwd", bwd)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(20):
            fn()
        ev1.record()
        torch.cuda.synchronize()
        buf = np.zeros(8192 * 16, dtype=np.uint64)
        assert lib.hermnet_debug_stamps16(buf.ctypes.data, buf.size) == 0
        st = buf.reshape(8192, 16)[:min(blocks, 8192)].astype(np.int64)
        sl = SLOTS[name]
        seq = st[:, sl]
        d = np.diff(seq, axis=1)
        life = seq[:, -1] - seq[:, 0]
        span = seq.max() - seq[:, 0].min()
        print("%s: %d workgroups, %.1f us per launch (stamped build); wave-0 life mean %d max %d cycles, whole grid %d cycles"
              % (name, len(st), ev0.elapsed_time(ev1) / 20 * 1e3, life.mean(), life.max(), span))
        hw = st[:, 15].astype(np.uint64)
        xcc, raw = (hw >> np.uint64(32)) & np.uint64(0xf), hw & np.uint64(0xffffffff)
        cu = (raw >> np.uint64(8)) & np.uint64(0xf)
        sh = (raw >> np.uint64(12)) & np.uint64(0x1)
        se = (raw >> np.uint64(13)) & np.uint64(0x7)
        key = ((xcc * np.uint64(8) + se) * np.uint64(2) + sh) * np.uint64(16) + cu
        uniq, cnt = np.unique(key, return_counts=True)
        print("   placement: %d distinct CUs; workgroups per used CU: %s" % (len(uniq), dict(zip(*np.unique(cnt, return_counts=True)))))
        t0 = seq[:, 0].min()
        for c in sorted(set(cnt)):
            ks = uniq[cnt == c]
            ends = [max(seq[key == u, -1]) - t0 for u in ks]
            starts = [sorted(seq[key == u, 0] - t0) for u in ks]
            print("   CUs with %d workgroups: last end mean %d max %d; start times of their workgroups (mean): %s"
                  % (c, np.mean(ends), np.max(ends), np.mean(np.array(starts), axis=0).astype(int).tolist()))
        for k, nm in enumerate(NAMES[name]):
            if nm is not None:
                print("   %-28s mean %7d  max %7d  (%.1f %% of a life)" % (nm, d[:, k].mean(), d[:, k].max(), 100.0 * d[:, k].mean() / life.mean()))


if __name__ == "__main__":
    main()
