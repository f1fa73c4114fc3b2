#!/usr/bin/env python3
"""Micro-benchmark of the two message kernels on the config-2 graph (10k atoms, E=431,340).
Library options (include/hermnet_hip.h: HN_OPT_*) from HN_OPTIONS="fwd_rows=17,bwd_lanes16=1,..." (tools/_opts.py).
Prints avg ms and algorithmic GB/s per kernel."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import hermnet_amd as hn  # noqa: E402
from hermnet_amd import synth, _lib  # noqa: E402
from hermnet_amd.ops import EdgeGeometry, _stream  # noqa: E402
from hermnet_amd.relations import RelationalGraph  # noqa: E402
from bench import algorithmic_bytes  # noqa: E402
from _opts import apply_option_env  # noqa: E402


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda:0")
    lib = _lib.load()
    apply_option_env()
    d = synth.fcc_alloy().to(dev)
    # KBENCH_ORDER=x|y|z: renumber the atoms along one axis first (row order = locality of the gathers)
    axis = os.environ.get("KBENCH_ORDER", "")
    if axis:
        perm = torch.argsort(d.pos[:, "xyz".index(axis)], stable=True)
        inv = torch.empty_like(perm)
        inv[perm] = torch.arange(perm.numel(), device=dev)
        d.pos, d.atomic_number, d.edge_index = d.pos[perm].contiguous(), d.atomic_number[perm].contiguous(), inv[d.edge_index]
    g = RelationalGraph.build(d.atomic_number, d.edge_index, [13, 28, 29], d.edge_shift, d.batch)
    # KBENCH_LOCAL=K: fold every gather index into K rows (WRONG results; the run time with every gather an L2 hit)
    fold = int(os.environ.get("KBENCH_LOCAL", "0"))
    if fold:
        g.csc_tgt.remainder_(fold)
        g.csr_src.remainder_(fold)
    # KBENCH_LOCAL_REC=K: every backward edge reads one of K radial records (WRONG results; the run time with the
    # scalar-path record loads served by the scalar cache)
    fold_rec = int(os.environ.get("KBENCH_LOCAL_REC", "0"))
    if fold_rec:
        g.csc_pos.remainder_(fold_rec)
    model = hn.HVNet(["Al", "Ni", "Cu"], rc=5.0, num_layers=1, hidden_channels=128, num_rbf=128).to(dev)
    rbf = model.radial_basis.descriptor()
    N, E, T, H, R = g.N, g.E, g.T, 128, 128
    edge = EdgeGeometry.apply(d.pos, d.cell, g)
    gen = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=gen)
    xh, x, vec = rnd(T, N, 3 * H), rnd(N, H), rnd(N, 3, H)
    wt, brbf = rnd(T, R, 3 * H) / 11.3, rnd(T, 3 * H) * 0.1
    xb = rnd(T, 3 * H) * 0.1 if os.environ.get("KBENCH_XH_BIAS", "1") != "0" else None
    x1, vec1 = torch.empty_like(x), torch.empty_like(vec)
    gx1, gvec1 = rnd(N, H), rnd(N, 3, H)
    split = 0
    gxh, gx = torch.empty_like(xh), torch.empty_like(x)
    gvec = torch.empty(T, N, 3, H, device=dev) if split else torch.empty_like(vec)
    gedge = torch.zeros(H // 64, E, 4, device=dev)
    gs, rs = g.as_struct(), rbf.struct()
    P = _lib.ptr

    def fwd(v):
        return lib.hermnet_message_scatter_fwd(ctypes.byref(gs), ctypes.byref(rs), H, P(xh), P(xb), P(v), P(x), P(wt), P(brbf),
                                               P(edge), P(x1), P(vec1), None, 1, 0, _stream())

    # KBENCH_TABLE=0: without the per-edge radial table the backward takes its 16-lanes-per-edge (VW) form
    from hermnet_amd.ops import edge_radial_table
    table = edge_radial_table(g, rbf, edge) if os.environ.get("KBENCH_TABLE", "1") != "0" else None

    part = torch.empty(T, N, 3, H, device=dev) if table is not None else None

    def bwd(v):
        return lib.hermnet_message_scatter_bwd(ctypes.byref(gs), ctypes.byref(rs), H, P(xh), P(xb), P(v), P(wt), P(brbf), P(edge),
                                               P(gx1), P(gvec1), P(gxh), P(gvec if v is not None else None), P(gx),
                                               P(gedge), split, P(table), P(part), None, None, 0, _stream())

    ab = algorithmic_bytes(E, N, H, T)
    res = {}
    for name, fn, arg in [("message_scatter_fwd", fwd, vec), ("message_scatter_fwd_l0", fwd, None),
                          ("message_scatter_bwd", bwd, vec), ("message_scatter_bwd_l0", bwd, None)]:
        for _ in range(3):
            assert fn(arg) == 0
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(iters):
            fn(arg)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / iters
        res[name] = ms
        print("%-24s %8.3f ms  %7.1f GB/s (algorithmic)" % (name, ms, ab[name] / 1e6 / ms))
        if hasattr(lib, "hermnet_debug_stamps_cl") and "bwd" in name and table is not None:
            buf = (ctypes.c_ulonglong * 12)()
            lib.hermnet_debug_stamps_cl(buf)
            tot, stage, pro, rec, con, alg, red, epi, edges, segs, waves = [float(v) for v in buf[:11]]
            if tot > 0:
                print("   stamps(cl): per wave %.0f cycles; staging %.1f%% | seg prologue %.1f%% | record wait %.1f%% | contraction "
                      "%.1f%% | algebra %.1f%% | reduce+store %.1f%% | epilogue+tail %.1f%%;  per edge: record %.0f, "
                      "contraction %.0f, algebra %.0f, reduce %.0f; per segment: prologue %.0f, epilogue %.0f; edges/seg %.1f"
                      % (tot / waves, 100 * stage / tot, 100 * pro / tot, 100 * rec / tot, 100 * con / tot, 100 * alg / tot,
                         100 * red / tot, 100 * epi / tot, rec / edges, con / edges, alg / edges, red / edges, pro / segs,
                         epi / segs, edges / segs))
        elif hasattr(lib, "hermnet_debug_stamps") and "bwd" in name:      # diagnostic build (-DHN_STAMPS)
            buf = (ctypes.c_ulonglong * 8)()
            lib.hermnet_debug_stamps(buf)
            tot, stage, pro, it, epi, segs, waves = [float(v) for v in buf[:7]]
            if tot > 0:
                print("   stamps: per wave %.0f cycles; staging %.1f%% | segment prologue %.1f%% | iterations %.1f%% | "
                      "epilogue %.1f%% | other %.1f%%; per segment: prologue %.0f, iterations %.0f, epilogue %.0f cycles"
                      % (tot / waves, 100 * stage / tot, 100 * pro / tot, 100 * it / tot, 100 * epi / tot,
                         100 * (tot - stage - pro - it - epi) / tot, pro / segs, it / segs, epi / segs))
    knobs = {k: v for k, v in os.environ.items() if k.startswith("HERMNET_") or k.startswith("KBENCH_") or k == "HN_OPTIONS"}
    print("knobs", knobs, "checksum", float(x1.sum() + vec1.sum()), float(gxh.sum() + gvec.sum() + gedge.sum()))


if __name__ == "__main__":
    main()
