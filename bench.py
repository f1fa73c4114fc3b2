#!/usr/bin/env python3
"""Headline benchmark: atom-steps/sec (energy + forces) of HVNet, fp32, synthetic data, random-init (seeded)
weights; model of BASELINE.json configs[1]: rc=5 A, hidden=128, num_rbf=128, 5 layers, 3 elements.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config auto|c2|c4|weak] [--no-cpu-baseline]

A "step" = HVNet.forward(data) + autograd.grad(E, pos) on a prebuilt `Data` (neighbour list excluded, as in
SURVEY.md section 8(d)); the relation-ordered graph build (the replacement of the reference's per-layer
`in_subgraph`) IS inside the step.  Inputs are resident in HBM before the timed region.

Workloads (`--config`; `auto` = c2 on one GPU, c4 on several):
  c2    BASELINE configs[1]: the 10,000-atom fcc alloy cell (10 x 10 x 25), the configuration the metric is quoted on.
  c4    BASELINE configs[3]: the FIXED 100,000-atom cell (fcc 10 x 10 x 250, 36 x 36 x 900 A), sharded by atom into N
        slabs with a one-hop halo: STRONG scaling (total work fixed).  On one GPU: the same cell unsharded.
  weak  one cell of N x 10k atoms (fcc 10 x 10 x 25N), ~10k owned atoms per rank: weak scaling (round 1's mode).
N > 1: one process per GPU.  `python bench.py --gpus N` on its own starts the N ranks itself (fresh child processes via
`python -m torch.distributed.run`, before this process makes any GPU call) and relays rank 0's line; inside a torchrun job
(WORLD_SIZE set, the driver's command line) it is one of the ranks.  Every rank plans its slab on the device from the coordinates alone
(`sharding.SlabStepper`: owners, geometric halo of rc + skin kept while valid, neighbour search over owned + halo atoms only); per layer one
RCCL all-to-all moves the halo rows, the energy is one scalar all-reduce.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

# read by the HSA runtime when it initialises (first HIP call): must be in the environment before that
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# Atom-sharded runs: HIP multiplexes its streams over GPU_MAX_HW_QUEUES hardware queues (default 4) and two streams that share
# a queue run in submission order.  With 4 queues the process group's internal stream shared its queue with the default
# stream -- the "asynchronous" all-to-all then sits in FRONT of the kernels it is meant to hide behind (kernel trace of the
# self-peer step and tools/overlap_probe.py: overlap 0.05 with 4 queues, 0.57 with 8; profiles/r06_overlap_probe.json).
if int(os.environ.get("WORLD_SIZE", "1")) > 1 or any(a in sys.argv for a in ("--self-peer", "--shard-anyway")):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8 TB/s (the guide measured 6.29 TB/s for a float4 copy;
                           # this pool's boxes measure ~5.2 TB/s: `roofline.measured_copy_GBps`)
FP32_MFMA_PEAK_TF = 155.0  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 / 16x16x4_f32, measured (157.3 spec)


def algorithmic_bytes(E, N, H, T):
    """Algorithmic HBM bytes per launch of the message kernels (DESIGN.md section "Kernels").
    fwd: per edge 2 int32 + (rhat,d) 16 B + xh_j row 12H (+ vec_j row 12H when layer>0);
                       per target atom: x1,vec1 write 16H.
    bwd (this build):  per edge 2 int32 + edge 16 B + gx1,gvec1 rows of the target 16H + gD write 16*(H/64);
                       per source atom and relation: xh row 12H read + gxh row 12H write; per atom: vec 12H read,
                       gvec,gx 16H write, gx1/gvec1 identity 16H read."""
    fwd_l0 = E * (8 + 16 + 12 * H) + N * 16 * H
    fwd = E * (8 + 16 + 24 * H) + N * 16 * H
    bwd = E * (8 + 16 + 16 * H + 16 * (H // 64)) + N * (T * 24 * H + 12 * H + 32 * H)
    return {"message_scatter_fwd_l0": fwd_l0, "message_scatter_fwd": fwd,
            "message_scatter_bwd": bwd, "message_scatter_bwd_l0": bwd - N * 12 * H}


def host_cores():
    """Cores this process may actually use: affinity, capped by the cgroup CPU quota and by 16 (the
    GPU box's per-GPU CPU share; oversubscribing OpenMP threads makes the CPU path far slower)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(model_kw, elems, seed):
    """The reference CPU path (oracle, mode="faithful": same op sequence incl. the O(N*E)
    in_subgraph loop) timed on a bounded sample of the same workload: 1,000-, 2,500- and (when the
    host is fast enough to stay within about a minute) the full 10,000-atom cell; the largest one
    that ran is reported."""
    from hermnet_amd import synth
    import hermnet_amd as hn
    from oracle import hermnet_oracle as orc
    m = hn.HVNet(elems, **model_kw)
    sd = synth.synth_state_dict(m.state_dict(), seed)
    kw = dict(rc=model_kw["rc"], num_layers=model_kw["num_layers"], hidden_channels=model_kw["hidden_channels"],
              num_rbf=model_kw["num_rbf"])
    cores = host_cores()
    torch.set_num_threads(cores)
    orc.energy_and_forces(sd, elems, synth.fcc_alloy(reps=(3, 3, 3)), mode="faithful", **kw)   # warm-up
    best = None
    for reps in [(5, 5, 10), (5, 5, 25), (10, 10, 25)]:      # 1,000 / 2,500 / the full 10,000 atoms
        sample = synth.fcc_alloy(reps=reps)
        t0 = time.time()
        orc.energy_and_forces(sd, elems, sample, mode="faithful", **kw)
        dt = time.time() - t0
        n = sample.pos.size(0)
        best = (n, dt, reps)
        if dt > 4.0:          # the next sample costs 4-16x (the in_subgraph loop is O(N*E)): stay within ~1 min
            break
    n, dt_probe, reps = best
    # The probe above was the warm-up at this size; ONE timed step follows.  (Rounds 1-5 timed two faithful and three vectorised
    # steps: ~85 s of CPU work, most of the driver's wall time for this benchmark; the contract asks for a bounded sample of
    # about 10-30 s.  A step is ~15 s on 16 cores and repeats to within 1 %: profiles/r05_final_bench.json.)
    sample = synth.fcc_alloy(reps=reps)
    t0 = time.time()
    orc.energy_and_forces(sd, elems, sample, mode="faithful", **kw)
    dt = time.time() - t0
    # the same sample through the oracle's vectorised mode (one mask per relation instead of the reference's
    # O(N*E) in_subgraph loop): the GPU/CPU ratio is not meant to be inflated by that loop.
    # Warm-up on a 2,500-atom slice (allocator, threads), then one timed step at full size.
    orc.energy_and_forces(sd, elems, synth.fcc_alloy(reps=(5, 5, 25)), mode="vectorised", **kw)
    t0 = time.time()
    orc.energy_and_forces(sd, elems, sample, mode="vectorised", **kw)
    dt_vec = time.time() - t0
    return {"value": n / dt, "unit": "atom-steps/s", "cores": cores, "kind": "port",
            "vectorised_value": n / dt_vec, "vectorised_timed_steps_s": [round(dt_vec, 2)],
            "timed_steps_s": [round(dt, 2)], "warmup_step_s": round(dt_probe, 2),
            "sample": "oracle mode=faithful (reference op sequence incl. in_subgraph loop), energy+forces steps on a "
                      "%d-atom slice (fcc %dx%dx%d) of the same alloy/model: one warm-up step at this size, then one timed "
                      "step: %.1f s on %d threads (about %.0f s of CPU work in all, the vectorised figure included)"
                      % (n, reps[0], reps[1], reps[2], dt, cores, dt_probe + dt + dt_vec + 3)}


def measured_copy_bandwidth(dev, nbytes=1 << 30, reps=10):
    """Device copy bandwidth of this box, GB/s (read + written bytes per second of a 1 GiB buffer copy): the library's
    float4 stream kernel (`hermnet_stream_copy`: 16-byte loads and stores, the method behind MI355X_MICROARCH.md's
    6.29 TB/s; best of a few grid sizes) and, beside it, what `Tensor.copy_` gets.  -> (stream kernel, copy_)."""
    from hermnet_amd import _lib
    lib = _lib.load()
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    stream = torch.cuda.current_stream().cuda_stream

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s0.record()
        for _ in range(reps):
            fn()
        s1.record()
        torch.cuda.synchronize()
        return 2.0 * nbytes * reps / (s0.elapsed_time(s1) * 1e-3) / 1e9

    best = 0.0
    for wgs in (256 * 4, 256 * 8, 256 * 16, 256 * 32):
        best = max(best, timed(lambda: _lib.check(lib.hermnet_stream_copy(a.data_ptr(), b.data_ptr(), a.numel(), wgs, stream),
                                                   "hermnet_stream_copy")))
    assert torch.equal(a[:4096], b[:4096]) and torch.equal(a[-4096:], b[-4096:])
    return best, timed(lambda: b.copy_(a))


def other_configs_secondary(hn, synth, dev, model_kw, steps=5, skip_c4=False):
    """configs[3] on ONE GPU (100k-atom cell, no sharding) and configs[4] (1024-molecule batch): energy + forces
    per step incl. the relation build, neighbour list prebuilt -- the same step definition as the headline."""
    res = {}
    cases = [("configs[2] 10k-atom 3-element cell, HTNet (18 triadic relations; build-defined model, the reference's "
              "class is a stub)", ["Al", "Ni", "Cu"], lambda: synth.fcc_alloy(reps=(10, 10, 25), seed=0, device=dev), hn.HTNet),
             ("configs[3] 100k-atom 3-element cell on 1 GPU", ["Al", "Ni", "Cu"],
              lambda: synth.fcc_alloy(reps=(10, 10, 250), seed=0, device=dev), hn.HVNet),
             ("configs[4] 1024-molecule batch", ["H", "C", "O"], lambda: synth.molecule_batch(num_graphs=1024).to(dev),
              hn.HVNet)]
    for name, elems, make, cls in cases:
        if skip_c4 and name.startswith("configs[3]"):
            continue
        d = make()
        model = cls(elems, **model_kw).eval()
        model.load_state_dict(synth.synth_state_dict(model.state_dict(), 10))
        model = model.to(dev)
        for p_ in model.parameters():
            p_.requires_grad_(False)

        def one():
            d.pos.requires_grad_(True)
            en = model(d)
            return en, -torch.autograd.grad(en.sum(), d.pos)[0]

        # >= 5 warm-up steps after the model is built (first-use costs: library solution loading for the read-out
        # GEMM shapes, allocator growth), then `steps` steps timed ONE BY ONE: min and median are reported, the
        # median is the figure
        for _ in range(5):
            one()
        torch.cuda.synchronize()
        ts = []
        for _ in range(max(steps, 10)):
            t0 = time.perf_counter()
            en, f = one()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        dt = ts[len(ts) // 2]
        res[name] = {"atoms": d.pos.size(0), "edges": d.edge_index.size(1), "graphs": int(en.numel()),
                     "ms_per_step": dt * 1e3, "ms_per_step_min": ts[0] * 1e3, "timed_steps": len(ts),
                     "atom_steps_per_s": d.pos.size(0) / dt}
        del d, model
    return res


def count_launches(fn):
    """Kernel launches of one call of `fn`, counted by the profiler (every device kernel, torch's included) ->
    (all, this library's own)."""
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        fn()
        torch.cuda.synchronize()
    names = [ev.name for ev in prof.events() if ev.device_type is not None and str(ev.device_type).endswith("CUDA")]
    own = sum(1 for n_ in names if "anonymous namespace" in n_ or n_.startswith("void (anonymous"))
    return len(names), own


def reference_default_width_secondary(hn, synth, dev, data, model_kw, steps=10):
    """The headline cell at the REFERENCE'S DEFAULT width, hidden_channels = 512 (hermnet.py:86; the width its examples
    train at): the node chains run on csrc/node_chain_wide.hip, the message kernels on 8 column blocks."""
    kw = dict(model_kw, hidden_channels=512)
    model = hn.HVNet(["Al", "Ni", "Cu"], **kw).eval()
    model.load_state_dict(synth.synth_state_dict(model.state_dict(), 10))
    model = model.to(dev)
    for p_ in model.parameters():
        p_.requires_grad_(False)
    d = hn.Data(**{k: v for k, v in data if not k.startswith("_hn")})
    d.pos = d.pos.detach()

    def one():
        d.pos.requires_grad_(True)
        en = model(d)
        return en, -torch.autograd.grad(en.sum(), d.pos)[0]

    for _ in range(3):
        one()
    torch.cuda.synchronize()
    from hermnet_amd import ops
    gt = ops.KernelTimer(prefix=("gemm", "node_", "message_"))
    ops.set_kernel_timer(gt)
    one()
    torch.cuda.synchronize()
    ops.set_kernel_timer(None)
    ts = []
    for _ in range(steps):
        t0 = time.perf_counter()
        en, f = one()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    ts.sort()
    dt = ts[len(ts) // 2]
    n_all, n_own = count_launches(one)
    N, H, T, L = d.pos.size(0), 512, 3, kw["num_layers"]
    gs = gt.summary()
    node_ms = sum(c * ms for k, (c, ms) in gs.items() if k.startswith("node_") or k == "gemm")
    gflop = gemm_flops_per_step(N, N, H, T, L) / 1e9
    return {"workload": "configs[1] cell (%d atoms, %d edges), HVNet hidden=512 num_rbf=%d layers=%d" %
                        (N, d.edge_index.size(1), kw["num_rbf"], L),
            "ms_per_step": dt * 1e3, "ms_per_step_min": ts[0] * 1e3, "atom_steps_per_s": N / dt,
            "launches_per_step": n_all, "launches_per_step_own_kernels": n_own,
            "node_chain_ms_per_step": node_ms, "node_gflop_per_step": gflop,
            "node_mfma_util": (gflop / node_ms / FP32_MFMA_PEAK_TF) if node_ms > 0 else None,
            "kernels_ms": {k: {"launches": c, "avg_ms": ms} for k, (c, ms) in gs.items()},
            "energy": float(en.detach()[0])}


def decomposition_secondary(synth, dev, rc, world=8, skin=1.0):
    """What the planners give at `world` ranks for configs[3]'s cell and for SURVEY 8(d) C4's near-cubic stress variant
    (~100k atoms, 104 A cube): owned atoms, halo atoms, the share of owned atoms whose messages run while the halo
    exchange is in flight ("interior"), peers per rank -- slabs against boxes (`sharding.plan_blocks`).  Planning only
    (no multi-GPU run): the figures the exchange volume and the hidden share follow from."""
    import numpy as np
    from hermnet_amd.sharding import block_grid, plan_blocks, plan_slab
    out = {}
    for name, reps in (("configs[3] cell 36x36x900 A (fcc 10x10x250)", (10, 10, 250)),
                       ("near-cubic stress variant 104 A cube (fcc 29x29x29)", (29, 29, 29))):
        pos_np, cell_np, z_np = synth.fcc_alloy_atoms(reps=reps, seed=0)
        pos = torch.from_numpy(pos_np.astype(np.float32)).to(dev)
        cell = torch.from_numpy(cell_np.astype(np.float32)).to(dev)
        z = torch.from_numpy(z_np).to(dev)
        res = {"atoms": int(pos.size(0))}
        for kind, grid in (("slabs", None), ("boxes", block_grid(world, cell))):
            owned, halo, interior, peers = [], [], [], []
            for r in range(world):
                pl = (plan_slab(pos, z, cell, rc, r, world, skin=skin) if grid is None
                      else plan_blocks(pos, z, cell, rc, r, world, grid=grid, skin=skin))
                owned.append(pl.n_owned)
                halo.append(int(pl.halo_global.numel()))
                interior.append(int((~pl.late_local).sum()))
                peers.append(sum(1 for c in pl.atom_plan.recv_counts if c))
            res[kind] = {"grid": list(grid) if grid else [1, 1, world], "owned_max": max(owned), "halo_max": max(halo),
                         "halo_over_owned": max(halo) / max(max(owned), 1),
                         "interior_share_min": min(i / max(o, 1) for i, o in zip(interior, owned)),
                         "peers_max": max(peers), "local_rows_max": max(o + h for o, h in zip(owned, halo))}
        out[name] = res
    out["note"] = "rc %.1f A + skin %.1f A; halo = geometric reach along every cut axis; one rank's figures are the maxima" % (rc, skin)
    return out


def self_peer_secondary(timeout=240.0):
    """The load of ONE rank of the 8-slab plan of configs[3], halo exchange included, on this one GPU (VERDICT r5
    item 1): 12,400 owned atoms (fcc 10 x 10 x 31) + the ~1,400 ghost rows of one slab across the periodic boundary, the
    per-layer exchange running over RCCL with this rank as its own peer (`sharding.plan_self_peer`, `bench.py --self-peer 1`).
    A CHILD process (it needs a process group and more hardware queues than the headline run: see the top of this file); its
    line is condensed here.  What a one-GPU box can say about the 8-GPU curve: `projected_speedup_at_8` = the unsharded 100k
    cell's step / this step -- compute, launches and stream hand-offs are real, the links are not (a local copy)."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--self-peer", "1", "--config", "c2", "--reps", "10,10,31", "--steps", "30",
           "--warmup", "5", "--no-cpu-baseline"]
    env = dict(os.environ, HERMNET_BENCH_CHILD="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    env["MASTER_PORT"] = "29583"
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    if r.returncode != 0 or not lines:
        return {"error": "child exited with %d: %s" % (r.returncode, r.stderr[-400:])}
    d = json.loads(lines[-1])
    c, sec = d.get("comm", {}), d.get("secondary", {})
    same = sec.get("single_gpu_same_cell", {})
    out = {"workload": d["config"]["workload"], "parallelism": d["config"]["parallelism"],
           "owned_atoms": d["config"]["atoms_owned_rank0"], "ghost_rows": d["config"]["halo_atoms_rank0"],
           "ms_per_step": d["ms_per_step"], "unsharded_same_cell_ms_per_step": same.get("ms_per_step"),
           "over_unsharded": (d["ms_per_step"] / same["ms_per_step"]) if same.get("ms_per_step") else None,
           "ms_per_step_incl_planning": sec.get("ms_per_step_incl_planning"),
           "exchanges_per_step": c.get("exchanges_per_step"), "bytes_per_row": c.get("bytes_per_row"),
           "rows_per_exchange": c.get("rows_sent_per_exchange", {}).get("max"),
           "stream_wait_ms_per_exchange": c.get("stream_wait_ms_per_exchange"),
           "all_to_all_ms_isolated": c.get("all_to_all_ms_isolated", {}).get("max"), "hidden_fraction": c.get("hidden_fraction"),
           "energy": d.get("energy")}
    return out


def graph_replay_secondary(hn, synth, dev, model, data, model_kw, steps=20):
    """ms per step of whole-step hipGraph replay: the headline workload and the 1024-molecule batch (launch-bound
    when enqueued eagerly)."""
    from hermnet_amd.graph import GraphedStep
    res = {}
    cases = [("headline workload", model, hn.Data(**{k: v for k, v in data if not k.startswith("_hn")}))]
    mol = synth.molecule_batch(num_graphs=1024).to(dev)
    m2 = hn.HVNet(["H", "C", "O"], **model_kw).eval()
    m2.load_state_dict(synth.synth_state_dict(m2.state_dict(), 10))
    m2 = m2.to(dev)
    for p_ in m2.parameters():
        p_.requires_grad_(False)
    cases.append(("configs[4] 1024-molecule batch", m2, mol))
    # a launch-bound case: configs[0]'s 64-atom Si cell (2 layers) -- eager enqueueing, not the GPU, sets its step time
    m3 = hn.HVNet(["Si"], rc=5.0, num_layers=2, hidden_channels=128, num_rbf=128).eval()
    m3.load_state_dict(synth.synth_state_dict(m3.state_dict(), 1))
    m3 = m3.to(dev)
    for p_ in m3.parameters():
        p_.requires_grad_(False)
    cases.append(("configs[0] 64-atom Si cell", m3, synth.si_diamond().to(dev)))
    for _, _, d in cases:
        d.pos = d.pos.detach()
    import torch.cuda.tunable as tunable
    was = tunable.is_enabled()
    tunable.enable(False)        # TunableOp's per-call bookkeeping is not capturable; library default GEMM choices here
    try:
        steps_ = [(name, GraphedStep(m, d), d) for name, m, d in cases]
    finally:
        tunable.enable(was)
    for name, step, d in steps_:
        t = {}
        for kind, fn in (("graph", step), ("eager", step._eager)):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                fn()
            torch.cuda.synchronize()
            t[kind] = (time.perf_counter() - t0) / steps
        res[name] = {"ms_per_step": t["graph"] * 1e3, "atom_steps_per_s": d.pos.size(0) / t["graph"],
                     "eager_ms_per_step_same_settings": t["eager"] * 1e3}
    res["note"] = ("one hipGraph launch per step (relation build + forward + force backward captured once); valid while "
                   "the neighbour list is unchanged; library-default GEMM solutions")
    # the MD-style step -- neighbour search INCLUDED -- as one graph launch (graph.GraphedMDStep: padded list, no host read;
    # the same graph stays valid when the list changes)
    try:
        from hermnet_amd.graph import GraphedMDStep
        md = {}
        for name, m, d in cases:
            if d.get("cell") is None:
                continue
            cell = d.cell.reshape(3, 3)
            g = GraphedMDStep(m, d.atomic_number, cell, d.pos.detach())
            jitter = 0.01 * torch.randn_like(d.pos.detach())
            for _ in range(3):
                g()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(steps):
                g(d.pos.detach() + (jitter if k & 1 else -jitter))       # (moving coordinates: the list is rebuilt anyway)
            torch.cuda.synchronize()
            dt_g = (time.perf_counter() - t0) / steps
            ok, n_edges = g.check()
            md[name] = {"ms_per_step": dt_g * 1e3, "atom_steps_per_s": d.pos.size(0) / dt_g, "capacity": g.capacity,
                        "edges_found": n_edges, "list_complete": bool(ok)}
        res["md_step_incl_neighbour_search_as_one_graph"] = md
    except Exception as ex:
        res["md_step_incl_neighbour_search_as_one_graph"] = {"error": repr(ex)}
    return res


def training_secondary(hn, synth, dev, model_kw, num_graphs=1024, steps=3):
    import torch.nn.functional as F
    d = synth.molecule_batch(num_graphs=num_graphs).to(dev)
    model = hn.HVNet(["H", "C", "O"], **model_kw)
    model.load_state_dict(synth.synth_state_dict(model.state_dict(), 12))
    model = model.to(dev).train()
    opt = torch.optim.Adam(model.parameters(), lr=3e-4)
    gen = torch.Generator().manual_seed(0)
    y = torch.randn(num_graphs, generator=gen).to(dev)
    ftgt = (0.5 * torch.randn(d.pos.shape, generator=gen)).to(dev)

    def train_step():
        opt.zero_grad()
        d.pos.requires_grad_(True)
        pred_e = model(d)
        pred_f = -torch.autograd.grad(pred_e.sum(), d.pos, create_graph=True)[0]
        loss = 0.2 * F.mse_loss(pred_e, y) + 0.8 * F.mse_loss(pred_f, ftgt)
        loss.backward()
        opt.step()
        return loss

    def timed():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss_ = train_step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps, float(loss_)

    train_step()
    dt_default, loss = timed()
    # The step's ~9 ms of node-level library GEMMs with the library's DEFAULT solution choice above; below the same step after
    # PyTorch's TunableOp has timed the candidates of its ~35 GEMM shapes in one untimed warm-up step (what a training run
    # does once: hermnet_amd.utils.enable_tuned_gemms(online=True) ... freeze_gemm_tuning()).  Same arithmetic (fp32 GEMMs), other
    # tiles.  Nothing is tuned inside the timed steps; the selection is switched off again for the rest of the benchmark.
    dt, tuned_note = dt_default, "library default solutions (online tuning unavailable)"
    try:
        import tempfile
        import torch.cuda.tunable as tunable
        was_on = tunable.is_enabled()
        tunable.enable(True)
        tunable.set_filename(os.path.join(tempfile.gettempdir(), "hermnet_tunableop_train_%d.csv" % os.getpid()))
        tunable.set_max_tuning_iterations(8)          # (the products take 30-80 us: a handful of runs ranks the candidates)
        tunable.set_max_tuning_duration(4)            # ms per candidate at most
        tunable.tuning_enable(True)
        t0 = time.perf_counter()
        train_step()
        torch.cuda.synchronize()
        tune_s = time.perf_counter() - t0
        tunable.tuning_enable(False)
        train_step()
        dt_tuned, _ = timed()
        if dt_tuned <= dt_default:
            dt = dt_tuned
            tuned_note = ("TunableOp: candidates of every GEMM shape timed in ONE untimed warm-up step (%.1f s), frozen for the "
                          "timed steps" % tune_s)
        else:       # (a table tuned on a busy box can lose to the defaults: a deployment would not keep it either)
            tuned_note = ("library default solutions: the TunableOp table of this run (tuned in %.1f s) was slower, %.2f ms per "
                          "step, and is not used" % (tune_s, dt_tuned * 1e3))
        if not was_on:
            tunable.enable(False)
    except Exception as ex:
        tuned_note = "library default solutions (%r)" % (ex,)
    res = {"workload": "configs[4]: %d molecules, %d atoms, %d edges; loss = 0.2 MSE(E) + 0.8 MSE(F), Adam"
                       % (num_graphs, d.pos.size(0), d.edge_index.size(1)),
           "ms_per_step": dt * 1e3, "graphs_per_s": num_graphs / dt, "atom_steps_per_s": d.pos.size(0) / dt,
           "ms_per_step_default_gemm_solutions": dt_default * 1e3, "gemm_selection": tuned_note,
           "loss": float(loss)}

    def no_optimizer():                     # what tools/train_profile.py counts (forward + force pass + backward)
        opt.zero_grad()
        d.pos.requires_grad_(True)
        pred_e = model(d)
        pred_f = -torch.autograd.grad(pred_e.sum(), d.pos, create_graph=True)[0]
        (0.2 * F.mse_loss(pred_e, y) + 0.8 * F.mse_loss(pred_f, ftgt)).backward()
    try:
        res["launches_per_step"] = count_launches(train_step)[0]
        res["launches_per_step_without_optimizer"] = count_launches(no_optimizer)[0]
    except Exception as ex:
        res["launches_per_step"] = {"error": repr(ex)}
    return res


def gemm_flops_per_step(N, nk, H, T, layers):
    """FLOPs of the dense feature-mixing linears of one energy+forces step (forward + hand-written backward; every
    GEMM has one backward GEMM of the same size w.r.t. its input: parameters are constants):
      pre-message (rmnet.py:52) on all N rows:  LN -> [H -> T*H] -> SSiLU -> per relation [H -> 3H]
      update (rmnet.py:94-107) on the nk target rows:  vec_proj 3 x [H -> 2H], [2H -> H], [H -> 3H]."""
    pre = 2 * N * H * (T * H) + 2 * T * N * H * 3 * H
    upd = 2 * 3 * nk * H * 2 * H + 2 * nk * 2 * H * H + 2 * nk * H * 3 * H
    return 2 * layers * (pre + upd)


def chain_kernel_bounds(gsum, N, nk, H, T, graph=None):
    """The node chain kernels against their OWN bounds (csrc/node_chain.hip): a launch is a few hundred workgroups, each a fixed
    number of MFMAs on the four SIMDs of one CU, so the time cannot go below (most workgroups any CU gets) x (a workgroup's
    matrix-pipe time).  Since round 5 an fp32 product runs as six bf16 partial products (v_mfma_f32_32x32x16_bf16, 4096
    FLOP/clk per CU): 6 x the FLOPs at 16 x the rate of the fp32 MFMA (256 FLOP/clk per CU) = 0.375 of its time.  `bound_us` is
    that quantised bound, `fp32_pipe_bound_us` what the fp32 MFMAs would have needed for the same grid (the kernels now run
    BELOW it where frac_fp32_pipe > 1), `hbm_bound_us` the launch's algorithmic bytes at 8 TB/s -- the bound the update kernels
    are closest to now.  `clock` 2.4 GHz as the peaks assume."""
    cus, flop_per_clk_cu = 256, 256.0
    io_floats_per_row = {"node_pre_fwd": 5.0 * T * H, "node_pre_bwd": 5.0 * T * H, "node_update_fwd": 24.0 * H,
                         "node_update_bwd": (27.0 + 4.0 * T) * H}
    rows_of = {"node_pre_fwd": N, "node_pre_bwd": N, "node_update_fwd": nk, "node_update_bwd": nk}
    from hermnet_amd import nodeops
    tr_pre, tr_upd = nodeops.chain_tile_rows(H), nodeops.chain_tile_rows(H, update=True)
    if graph is not None:          # the update kernels' tile height is picked per row layout (16-row form on small grids)
        tr_upd = nodeops.update_tile_rows(graph, H) or tr_upd
    grids = {"node_pre_fwd": ((N + tr_pre - 1) // tr_pre * T, 8.0 * tr_pre * H * H),
             "node_pre_bwd": ((N + tr_pre - 1) // tr_pre * T, 8.0 * tr_pre * H * H),
             "node_update_fwd": ((nk + tr_upd - 1) // tr_upd, 22.0 * tr_upd * H * H),
             "node_update_bwd": ((nk + tr_upd - 1) // tr_upd, 22.0 * tr_upd * H * H)}
    out = {}
    for name, (grid, flops_wg) in grids.items():
        if name not in gsum or grid == 0:
            continue
        ms = gsum[name][1]
        per_cu = (grid + cus - 1) // cus
        fp32_us = per_cu * flops_wg / flop_per_clk_cu / 2.4e9 * 1e6
        bound_us = fp32_us * 6.0 / 16.0
        hbm_us = io_floats_per_row[name] * 4.0 * rows_of[name] / 8e12 * 1e6
        out[name] = {"workgroups": grid, "tile_rows": tr_upd if "update" in name else tr_pre, "most_per_cu": per_cu,
                     "gflop_per_launch": grid * flops_wg / 1e9,
                     "bound_us": bound_us, "measured_us": ms * 1e3, "frac": bound_us / (ms * 1e3),
                     "fp32_pipe_bound_us": fp32_us, "frac_fp32_pipe": fp32_us / (ms * 1e3),
                     "hbm_bound_us": hbm_us, "frac_hbm": hbm_us / (ms * 1e3)}
    return out


def self_launch(ngpus, argv, timeout=None):
    """Start the `ngpus` ranks of this benchmark as FRESH child processes (`python -m torch.distributed.run`, one
    process per GPU, rendezvous on 127.0.0.1 at a free port) and relay their output.  Called before this process has
    touched the GPU (a process that has initialised HIP must not fork/exec ranks).  Returns the exit code."""
    import socket
    import subprocess
    tools = [v for k in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_TOOL_LIBRARIES", "HSA_TOOLS_LIB",
                       "HERMNET_PROFILER_HINT")
             for v in [os.environ.get(k, "")] if "rocprof" in v.lower()]
    if tools:
        # under rocprofv3 this process would be an idle relay and the ranks unprofiled children: refuse
        print("bench.py: --gpus %d would start the ranks as child processes, which the profiler attached to THIS process "
              "(%s) does not see.  Profile one rank instead, with NO launcher between `--` and the program (the "
              "profiler's preloaded library has initialised the GPU by then, and a launcher would fork / exec the rank "
              "from such a process): MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 "
              "rocprofv3 ... -- python3 bench.py --shard-anyway (tools/profile_shard1.sh)." % (ngpus, tools[0]),
              file=sys.stderr)
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HERMNET_BENCH_CHILD="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("GPU_MAX_HW_QUEUES", "8")      # (see the top of this file)
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // ngpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ngpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    import signal
    import threading
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1, start_new_session=True)
    timed_out = []

    def expire():          # a hung rank (rendezvous, RCCL) must not block forever: end the whole child job
        timed_out.append(True)
        victims = []
        try:      # the elastic agent starts every rank in a session of its own: collect the whole tree first
            import psutil
            victims = [c.pid for c in psutil.Process(proc.pid).children(recursive=True)]
        except Exception:
            pass
        for pid in victims + [proc.pid]:           # exactly the processes this launcher started, by pid
            try:
                os.kill(pid, signal.SIGKILL)
            except OSError:
                pass

    watchdog = threading.Timer(timeout, expire) if timeout and timeout > 0 else None
    if watchdog is not None:
        watchdog.daemon = True
        watchdog.start()
    line = None
    for ln in proc.stdout:                 # relayed line by line, as the ranks write
        ln = ln.rstrip("\n")
        if ln.startswith("{") and '"metric"' in ln:
            line = ln                      # rank 0's JSON line
        else:
            print(ln, file=sys.stderr, flush=True)     # anything else the ranks wrote to stdout is not the result
    rc = proc.wait()
    if watchdog is not None:
        watchdog.cancel()
    if line is not None:
        print(line, flush=True)
    if timed_out:
        print("bench.py: the %d-rank child job did not finish within --launch-timeout %.0f s and was killed"
              % (ngpus, timeout), file=sys.stderr)
        return 124
    if rc != 0:
        print("bench.py: the %d-rank child job exited with code %d" % (ngpus, rc), file=sys.stderr)
        return rc
    return 0 if line is not None else 1


def rank_fence(sharded, backend, dev):
    """barrier + device synchronisation on both sides (the contract's bracket of the timed region)."""
    def fence():
        if dev is not None:
            torch.cuda.synchronize()
        if sharded:
            if backend == "nccl":
                dist.barrier(device_ids=[dev.index])
            else:
                dist.barrier()
        if dev is not None:
            torch.cuda.synchronize()
    return fence


def launch_rehearsal(args, world, rank):
    """`--launch-rehearsal`: the launcher, rendezvous, fences and the max-over-ranks timing around an EMPTY step --
    no model, no GPU, nothing measured (value null).  What the CPU suite can check of the N > 1 command line."""
    if world > 1:
        dist.init_process_group("gloo")
    fence = rank_fence(world > 1, "gloo", None)
    fence()
    if os.environ.get("HERMNET_REHEARSAL_HANG"):      # tests: a rank that never comes back (launcher watchdog)
        time.sleep(float(os.environ["HERMNET_REHEARSAL_HANG"]))
    t0 = time.perf_counter()
    fence()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    ranks = torch.ones(1, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(ranks)
    if rank == 0:
        print(json.dumps({"metric": "launch rehearsal (no workload)", "value": None, "unit": "atom-steps/s",
                          "n_gpus": world, "ranks_seen": int(ranks.item()), "steps": 0, "warmup": 0,
                          "ms_per_step": None, "scaling": "strong", "rehearsal": True}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def comm_block(step, plan, data, dev, backend, world, H, layers, kernel_ms_per_step, step_ms_local, T=3):
    """What the halo exchanges of the sharded step cost, per rank and over the ranks: every
    rank times its stream waits with HIP events (sharding.CommProbe) in a short pass behind the timed region, times the same
    all-to-all in isolation, and the per-rank figures are gathered to rank 0.  hidden fraction = 1 - (time the compute
    stream waited for an exchange) / (time the exchange takes by itself)."""
    from hermnet_amd import sharding
    probe = sharding.CommProbe()
    sharding.set_comm_probe(probe)
    nprobe = 5
    try:
        for _ in range(nprobe):
            step()
        rec = probe.summary()
    finally:
        sharding.set_comm_probe(None)
    ap = plan.atom_plan
    n_send, n_recv = sum(ap.send_counts), sum(ap.recv_counts)
    # floats per packed row: the "proj" form of the exchange (the default where it applies: layer.FusedRelationalLayer) moves
    # xh[t] of the T relations + vec = (T + 1) 3H floats per halo atom, the rows form x | vec = 4H
    proj = os.environ.get("HERMNET_HALO_OVERLAP", "2") not in ("0", "1") and H == 128
    row_floats = (T + 1) * 3 * H if proj else 4 * H
    bytes_row = row_floats * 4
    # the layer exchange by itself: the same packed rows, nothing else on the GPU
    iso = float("nan")
    try:
        buf = torch.zeros(n_send, row_floats, device=dev)
        for _ in range(3):
            sharding._all_to_all_rows(buf, ap.send_counts, ap.recv_counts, ap.group)
        torch.cuda.synchronize()
        dist.barrier(**({"device_ids": [dev.index]} if backend == "nccl" else {}))
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            sharding._all_to_all_rows(buf, ap.send_counts, ap.recv_counts, ap.group)
        b.record()
        torch.cuda.synchronize()
        iso = a.elapsed_time(b) / 20
    except Exception:
        pass
    g = lambda tag, key: rec.get(tag, {}).get(key, 0)
    per_step = lambda tag: g(tag, "ms_total") / nprobe
    vals = [kernel_ms_per_step, step_ms_local, per_step("fwd"), per_step("bwd"), per_step("blocking"),
            g("fwd", "count") / nprobe, g("bwd", "count") / nprobe, g("blocking", "count") / nprobe, iso,
            float(n_send), float(n_recv), float(sum(1 for c in ap.send_counts if c)), float(sum(1 for c in ap.recv_counts if c)),
            float(plan.halo_global.numel()), float(plan.owned_global.numel()), float(data.edge_index.size(1))]
    names = ["kernel_ms_per_step", "step_ms_local", "wait_fwd_ms_per_step", "wait_bwd_ms_per_step", "blocking_ms_per_step",
             "async_exchanges_fwd_per_step", "async_exchanges_bwd_per_step", "blocking_exchanges_per_step",
             "all_to_all_ms_isolated", "rows_sent_per_exchange", "rows_received_per_exchange", "peers_sent_to", "peers_received_from",
             "halo_atoms", "owned_atoms", "edges"]
    t = torch.tensor(vals, dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    allv = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(allv, t)
    m = torch.stack(allv).cpu()
    col = {n: m[:, i] for i, n in enumerate(names)}
    mm = lambda n: {"max": float(col[n].max()), "min": float(col[n].min()), "mean": float(col[n].mean())}
    nf, nb = float(col["async_exchanges_fwd_per_step"].max()), float(col["async_exchanges_bwd_per_step"].max())
    iso_max = float(col["all_to_all_ms_isolated"].max())
    out = {
        "what": "halo exchange of the sharded step: per layer ONE variable-size all_to_all_single of packed rows each way "
                "(proj form: the owner's projections xh[t] of every relation | vec forward, the partial sums of their gradients "
                "backward; rows form: x | vec), started asynchronously and waited for by the compute STREAM behind the messages "
                "into / out of the rows that read no halo row",
        "exchange_form": "proj" if proj else "rows",
        "exchanges_per_step": {"features_forward": layers - 1, "gradients_backward": layers - 1, "position_gradient_return": 1,
                               "scalar_all_reduces": 2, "asynchronous_forward": nf, "asynchronous_backward": nb,
                               "blocking": float(col["blocking_exchanges_per_step"].max())},
        "bytes_per_row": bytes_row,
        "rows_sent_per_exchange": mm("rows_sent_per_exchange"), "rows_received_per_exchange": mm("rows_received_per_exchange"),
        "bytes_sent_per_exchange": {k: v * bytes_row for k, v in mm("rows_sent_per_exchange").items()},
        "peers": {"sent_to": mm("peers_sent_to"), "received_from": mm("peers_received_from")},
        "bytes_per_exchange_per_peer_mean": float((col["rows_sent_per_exchange"] / col["peers_sent_to"].clamp(min=1)).mean()) * bytes_row,
        "all_to_all_ms_isolated": mm("all_to_all_ms_isolated"),
        "stream_wait_ms_per_exchange": {
            "forward": {k: v / max(nf, 1.0) for k, v in mm("wait_fwd_ms_per_step").items()},
            "backward": {k: v / max(nb, 1.0) for k, v in mm("wait_bwd_ms_per_step").items()}},
        "hidden_fraction": {
            "forward": None if not (nf and iso_max > 0) else 1.0 - float(col["wait_fwd_ms_per_step"].max()) / nf / iso_max,
            "backward": None if not (nb and iso_max > 0) else 1.0 - float(col["wait_bwd_ms_per_step"].max()) / nb / iso_max,
            "note": "1 - (slowest rank's stream wait per exchange) / (slowest rank's isolated all-to-all); <= 0: nothing hidden"},
        "wait_ms_per_step": {k: v for k, v in zip(("max", "min", "mean"), (
            float((col["wait_fwd_ms_per_step"] + col["wait_bwd_ms_per_step"] + col["blocking_ms_per_step"]).max()),
            float((col["wait_fwd_ms_per_step"] + col["wait_bwd_ms_per_step"] + col["blocking_ms_per_step"]).min()),
            float((col["wait_fwd_ms_per_step"] + col["wait_bwd_ms_per_step"] + col["blocking_ms_per_step"]).mean())))},
        "kernel_ms_per_step": mm("kernel_ms_per_step"), "step_ms_local": mm("step_ms_local"),
        "halo_atoms": mm("halo_atoms"), "owned_atoms": mm("owned_atoms"), "edges": mm("edges"),
        "per_rank": {n: [float(v) for v in col[n]] for n in ("step_ms_local", "kernel_ms_per_step", "wait_fwd_ms_per_step",
                                                              "wait_bwd_ms_per_step", "blocking_ms_per_step",
                                                              "all_to_all_ms_isolated", "rows_sent_per_exchange", "halo_atoms")},
    }
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="auto", choices=["auto", "c2", "c4", "weak"],
                    help="workload (see the module docstring); auto = c2 on one GPU, c4 (strong scaling) on several")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="headline step only (profiling runs)")
    ap.add_argument("--shard-anyway", action="store_true",
                    help="take the sharded (process-group) code path even with one rank: rehearsal of the RCCL calls "
                         "on a single GPU (launch through torch.distributed.run --nproc-per-node 1)")
    ap.add_argument("--self-peer", type=int, default=0, metavar="V",
                    help="one rank whose halo peer is itself (sharding.plan_self_peer, V virtual slabs; V = 1: one slab across "
                         "the periodic boundary = the load of one rank of an 8-slab plan): the production halo exchange with "
                         "real rows on ONE GPU over RCCL.  Implies --shard-anyway; world size 1 only")
    ap.add_argument("--reps", default=None, help="fcc cells nx,ny,nz of the synthetic alloy (default: the config's)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = rehearsal with several ranks sharing one GPU (exchange staged through the host)")
    ap.add_argument("--launch-rehearsal", action="store_true",
                    help="launcher check without a GPU: every rank joins the process group, runs the fences and the "
                         "max-over-ranks reduction of the timed region around an empty step, rank 0 prints a line with "
                         "value null (tests/test_bench_launcher.py)")
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="plain `--gpus N` only: seconds after which the child job is killed (a hung rank must not block "
                         "forever); 0 = none")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        if "WORLD_SIZE" in os.environ or os.environ.get("HERMNET_BENCH_CHILD"):
            raise SystemExit("bench.py --gpus %d inside a %d-rank job: launch with python -m torch.distributed.run "
                             "--nproc-per-node %d bench.py --gpus %d" % (args.gpus, world, args.gpus, args.gpus))
        # plain `python bench.py --gpus N`: this process has made no GPU call yet and never will -- it starts the N
        # ranks as fresh children, relays rank 0's line and exits with their code
        raise SystemExit(self_launch(args.gpus, sys.argv[1:], timeout=args.launch_timeout))
    if args.self_peer and world != 1:
        raise SystemExit("bench.py --self-peer is a one-rank plan (WORLD_SIZE=%d)" % world)
    sharded = world > 1 or args.shard_anyway or bool(args.self_peer)
    if sharded and "RANK" not in os.environ:        # (--shard-anyway / --self-peer started plainly: a one-rank group)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29555")
        os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    if args.launch_rehearsal:
        return launch_rehearsal(args, world, rank)
    cfg = args.config if args.config != "auto" else ("c4" if world > 1 else "c2")
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", local_rank % max(ndev, 1))
    torch.cuda.set_device(dev)
    if sharded:
        if args.backend == "nccl":
            # (this pool exports NCCL_DEBUG=VERSION and RCCL prints its banner with printf: it would land on stdout in front
            # of the JSON line -- the communicator is created with file descriptor 1 pointing at stderr)
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group("nccl", device_id=dev)
                dist.barrier(device_ids=[dev.index])
                torch.cuda.synchronize()
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)
        else:
            # (gloo announces its peers with printf as well: "[Gloo] Rank 0 is connected to 1 peer ranks" would land on stdout in
            # front of the JSON line)
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                dist.init_process_group("gloo")
                dist.barrier()
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)

    import numpy as np
    import hermnet_amd as hn
    from hermnet_amd import synth, ops, _lib
    from hermnet_amd.utils import enable_tuned_gemms, freeze_gemm_tuning
    _lib.load()   # fail loudly if the HIP library is missing
    if os.environ.get("HN_OPTIONS") or os.environ.get("HN_SWITCHES"):      # (A/B loops of tools/*.sh: tools/_opts.py)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from _opts import apply_option_env
        apply_option_env()
    # No library GEMM is left on the energy/force path (csrc/node_chain.hip, the fused read-out), so there is nothing for
    # TunableOp to choose: round 2's recorded table / online tuning during the warm-up -- and with it the risk of ranks
    # timing candidates differently in a multi-GPU run -- are off unless asked for (HERMNET_BENCH_TUNED_GEMMS=1: only the
    # training secondary and widths outside {64, 128, 256} still call library GEMMs).
    tuned = online_tuning = False
    if os.environ.get("HERMNET_BENCH_TUNED_GEMMS", "0") != "0" and os.environ.get("PYTORCH_TUNABLEOP_ENABLED") is None:
        online_tuning = args.warmup > 0
        tuned = enable_tuned_gemms(online=online_tuning)

    elems = ["Al", "Ni", "Cu"]
    model_kw = dict(rc=5.0, num_layers=5, hidden_channels=128, num_rbf=128)
    seed = 10
    model = hn.HVNet(elems, **model_kw).eval()
    model.load_state_dict(synth.synth_state_dict(model.state_dict(), seed))
    model = model.to(dev)
    for p_ in model.parameters():      # energy/force evaluation: no parameter gradients
        p_.requires_grad_(False)
    reps = {"c2": (10, 10, 25), "c4": (10, 10, 250), "weak": (10, 10, 25 * world)}[cfg]
    if args.reps:
        reps = tuple(int(v) for v in args.reps.split(","))
    scaling = "weak" if cfg == "weak" else "strong"
    if sharded:
        # every rank holds the global coordinates (what a calculator is handed per MD step) and plans ITS slab on the
        # device: owners, geometric halo, neighbour search over owned + halo atoms only (outside the timed region,
        # like the neighbour list of the single-GPU headline; timed separately below)
        from hermnet_amd.sharding import SlabStepper, plan_slab
        pos_np, cell_np, z_np = synth.fcc_alloy_atoms(reps=reps, seed=0)
        gpos = torch.from_numpy(pos_np.astype(np.float32)).to(dev)
        gcell = torch.from_numpy(cell_np.astype(np.float32)).to(dev)
        gz = torch.from_numpy(z_np).to(dev)
        group = dist.group.WORLD

        # the slab plan (owners, halo = rc + skin, exchange lists) is kept while no atom has moved further than skin/2;
        # per step only the displacement check and the slab-local neighbour search run (sharding.SlabStepper)
        SKIN = 1.0
        # (deferred: no host read per step -- the displacement flag and the padded list's count stay on the device until ONE
        # check() behind the step; the first call of a plan searches exactly, which is also the Data of the timed region)
        stepper = SlabStepper(gz, gcell, model_kw["rc"], rank, world, skin=SKIN, group=group, deferred=True,
                              self_peer=args.self_peer)

        def plan_shard():
            return stepper(gpos)

        data, plan = plan_shard()
        halo = int(plan.halo_global.numel())
        N_global = int(gpos.size(0))
        e_cnt = torch.tensor([data.edge_index.size(1)], dtype=torch.float64,
                             device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(e_cnt)
        E_global = int(e_cnt.item())
    else:
        # the neighbour list comes from the device cell list (bit-identical to the host list,
        # tests/test_gpu_parity.py) and is outside the timed region
        data, halo = synth.fcc_alloy(reps=reps, seed=0, device=dev), 0
        N_global, E_global = data.pos.size(0), data.edge_index.size(1)
    N, E = data.pos.size(0), data.edge_index.size(1)                # local: owned + halo atoms, edges by owned target
    H, T = model_kw["hidden_channels"], len(elems)

    def step(d=None):
        d = data if d is None else d
        d.pos.requires_grad_(True)
        e = model(d)
        f = -torch.autograd.grad(e.sum(), d.pos)[0]
        return e, f

    fence = rank_fence(sharded, args.backend, dev)

    def max_over_ranks(v):
        if not sharded:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    for _ in range(args.warmup):
        step()
    if online_tuning:
        freeze_gemm_tuning()
    timer = ops.KernelTimer()
    ops.set_kernel_timer(timer)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        e, f = step()
    t_host = time.perf_counter() - t0          # host time to enqueue the K steps (the GPU may still be running)
    fence()
    dt = time.perf_counter() - t0
    ops.set_kernel_timer(None)
    dt_local = dt
    dt = max_over_ranks(dt)

    # ---- after the timed region: the dense linears, timed with HIP events in a short pass of their own (an event
    # pair costs ~15 us of host time; ~50 of them per step inside the timed region would distort a ~3 ms step)
    gtimer = ops.KernelTimer(prefix=("gemm", "node_"))
    ops.set_kernel_timer(gtimer)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ops.set_kernel_timer(None)

    # ---- sharded runs: what the halo exchanges cost (every rank takes part; rank 0 prints)
    comm = None
    if sharded:
        try:
            # this rank's own kernels per step: message kernels (timed region) + node chain kernels (the pass above)
            own_ms = (sum(c * ms for _, (c, ms) in timer.summary().items()) / args.steps
                      + sum(c * ms for _, (c, ms) in gtimer.summary().items()) / 3.0)
            comm = comm_block(step, plan, data, dev, args.backend, world, model_kw["hidden_channels"], model_kw["num_layers"],
                              own_ms, dt_local / args.steps * 1e3)
        except Exception as ex:      # (never lose the headline over diagnostics -- but every rank must fail alike: collectives)
            comm = {"error": repr(ex)}

    # ---- secondary, all ranks: the step INCLUDING planning (sharded: slab plan + slab-local neighbour search;
    # single GPU: the device neighbour search), i.e. what one MD step costs end to end
    md = None
    if not args.no_secondary:
        from hermnet_amd.neighbor import neighbor_search
        if sharded:
            md_valid = []

            def md_step():
                d, _ = plan_shard()
                out_ = step(d)
                return out_
        else:
            pos0, cell0 = data.pos.detach(), data.cell

            def md_step():
                ei, sh = neighbor_search(pos0, model_kw["rc"], cell0)
                d = hn.Data(pos=pos0.clone(), atomic_number=data.atomic_number, batch=data.batch,
                            cell=cell0, edge_index=ei, edge_shift=sh)
                return step(d)

        for _ in range(2):
            md_step()
        fence()
        t1 = time.perf_counter()
        nmd = max(3, min(args.steps, 10))
        for _ in range(nmd):
            md_step()
        fence()
        md = max_over_ranks(time.perf_counter() - t1) / nmd
        if sharded:
            md_valid.append(bool(stepper.check()))       # ONE host read for the last step (a calculator does it per step, when
                                                         # it copies the results to the host anyway)
        md_padded = None
        if not sharded:
            # the same MD-style step with the neighbour list built WITHOUT its host read: padded to a capacity, count and
            # flags left on the device (checked once behind the timed loop here; a calculator checks them when it copies
            # the results to the host)
            from hermnet_amd.neighbor import neighbor_search_padded, padded_capacity, padded_list_ok
            cap = padded_capacity(E)
            totals = []

            def md_step_padded():
                ei, sh, total = neighbor_search_padded(pos0, model_kw["rc"], cell0, cap)
                d = hn.Data(pos=pos0.clone(), atomic_number=data.atomic_number, batch=data.batch,
                            cell=cell0, edge_index=ei, edge_shift=sh)
                d._hn_edge_count = total
                totals.append(total)
                return step(d)

            for _ in range(2):
                e_p, _f = md_step_padded()
            fence()
            t1 = time.perf_counter()
            for _ in range(nmd):
                md_step_padded()
            fence()
            md_padded = (time.perf_counter() - t1) / nmd
            ok, n_found = padded_list_ok(totals[-1])
            md_padded = {"ms_per_step": md_padded * 1e3, "atom_steps_per_s": N_global / md_padded, "capacity": cap,
                         "edges_found": n_found, "list_complete": bool(ok),
                         "energy_equals_exact_list": bool(torch.equal(e_p.detach(), step()[0].detach())),
                         "over_step": md_padded / (dt / args.steps)}
        replan_ms = None
        if sharded:       # what a re-plan costs when the skin is used up (every ~10-50 MD steps)
            fence()
            t1 = time.perf_counter()
            for _ in range(3):
                plan_slab(gpos, gz, gcell, model_kw["rc"], rank, world, group=group, skin=SKIN)
            fence()
            replan_ms = max_over_ranks(time.perf_counter() - t1) / 3 * 1e3

    if rank == 0:
        ksum = timer.summary()            # name -> (launches, mean ms), HIP events on the launch stream
        ab = algorithmic_bytes(E, N, H, T)
        kernels = {}
        for name, (cnt, ms) in ksum.items():
            kernels[name] = {"launches_per_step": cnt / args.steps, "avg_ms": ms}
            if name in ab:
                kernels[name].update({"algorithmic_GB": ab[name] / 1e9, "GBps": ab[name] / 1e9 / (ms / 1e3),
                                      "hbm_frac": ab[name] / 1e9 / (ms / 1e3) / HBM_PEAK_GBS})
        # dominant kernel = largest share of the step
        dom = max(ab, key=lambda k: kernels[k]["avg_ms"] * kernels[k]["launches_per_step"] if k in kernels else 0.0)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(dom)
        # what actually limits each message kernel (profiles/README.md): the backward kernels issue
        # VALU work (banded 12-tap contraction) for longer than their HBM time; the forward kernels' gathers are served
        # from L2 / Infinity Cache faster than HBM could deliver the same bytes
        # (forward: with every gather folded into an L2-resident set it gains 7 % -- profiles/r02_kbench_gather_locality.log --
        # so it is the dependent load -> LDS -> FMA chain at 2 waves per SIMD, not the gather rate, that sets its time)
        limiter = {"message_scatter_fwd": "issue/latency at 2 waves per SIMD (gathers from L2/Infinity Cache)",
                   "message_scatter_fwd_l0": "issue/latency (gathers from L2/Infinity Cache)",
                   "message_scatter_bwd": "valu-issue", "message_scatter_bwd_l0": "valu-issue"}
        traffic_all = json.load(open(tpath)) if os.path.exists(tpath) else {}
        for k in kernels:
            if k in limiter:
                kernels[k]["limiter"] = limiter[k]
                tb = traffic_all.get(k)
                if tb:      # HBM-side bytes per launch from the PMC passes (profiles/traffic.json, tools/profile_round.sh)
                    kernels[k]["traffic_GB"] = tb / 1e9
                    kernels[k]["hbm_frac_measured"] = tb / 1e9 / (kernels[k]["avg_ms"] / 1e3) / HBM_PEAK_GBS
                    kernels[k]["traffic_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the builder's box " \
                                                   "(profiles/traffic.json), not measured in this run"
                if k.startswith("message_scatter_fwd"):
                    kernels[k]["ceiling"] = ("gather rows are served by L2 / Infinity Cache, not HBM: the algorithmic fraction "
                                             "(`hbm_frac`) can exceed 1; the guide's random-row gather ceiling from the Infinity "
                                             "Cache is 8.6 TB/s (MI355X_MICROARCH.md, 'Indexed rows'), `hbm_frac_measured` is "
                                             "the HBM-side share")
                    kernels[k]["frac_of_cache_gather_ceiling"] = kernels[k]["GBps"] / 8600.0
        gsum = gtimer.summary()
        gemm_ms = sum(cnt * ms for _, (cnt, ms) in gsum.items()) / 3.0          # per step
        nk = N if not sharded else N
        gflop = gemm_flops_per_step(N, nk, H, T, model_kw["num_layers"]) / 1e9
        mfma = {"what": "dense feature-mixing linears (node MLPs) of one step, forward + backward",
                "gflop_per_step": gflop, "ms_per_step": gemm_ms, "launches_per_step": sum(c for c, _ in gsum.values()) / 3.0,
                "achieved_TFLOPs": (gflop / gemm_ms) if gemm_ms > 0 else None, "peak_TFLOPs_fp32_mfma": FP32_MFMA_PEAK_TF,
                "mfma_util": (gflop / gemm_ms / FP32_MFMA_PEAK_TF) if gemm_ms > 0 else None,
                "arithmetic": "fp32 values and fp32 accumulation; every product runs as a three-way bf16 split of both operands, "
                              "the six largest partial products on v_mfma_f32_{32x32x16,16x16x32}_bf16 (fp32-equivalent: the "
                              "dropped terms are <= 2^-24 of a product; 6 / 16 of the fp32 MFMA's pipe time). gflop_per_step and "
                              "mfma_util count the ALGORITHMIC fp32 FLOPs against the fp32 matrix peak",
                "bf16_pipe": {"executed_gflop_per_step": 6.0 * gflop, "peak_TFLOPs_bf16_dense": 2500.0,
                              "util": (6.0 * gflop / gemm_ms / 2500.0) if gemm_ms > 0 else None},
                "own_roofline": chain_kernel_bounds(gsum, N, nk, H, T, data.get("_hn_graph")),
                "own_roofline_note": "per chain kernel: bound_us = (most workgroups on one CU) x (a workgroup's bf16-pipe time at "
                                     "4096 FLOP/clk per CU, 2.4 GHz); fp32_pipe_bound_us = the same grid on the fp32 MFMAs; "
                                     "hbm_bound_us = algorithmic bytes of the launch at 8 TB/s; frac* = bound / measured"}
        from hermnet_amd import switches as _sw
        if H == 128 and _sw.defer_sums() and not sharded:
            mfma["note_pending_grads"] = ("node_update_bwd of layers 0..L-2 also forms its incoming gradients from the partial sums "
                                          "of the layer above (hn_pending_grads: what message_bwd_finish + layernorm_bwd_parts did "
                                          "in two launches outside this section): +12-15 us per launch of memory-bound work counted "
                                          "here, 1 % off the step; HERMNET_DEFER_SUMS=0 gives the separate launches")
        names = {"c2": "configs[1]: %d-atom 3-element fcc alloy (Al/Ni/Cu)",
                 "c4": "configs[3]: FIXED %d-atom 3-element fcc alloy cell (10x10x250, 36x36x900 A)",
                 "weak": "weak-scaling variant: %d-atom 3-element fcc alloy cell (10x10x25N)"}
        out = {
            "metric": "atom-steps/sec (energy+forces) on 10k-atom 3-element cell; HBM GB/s vs roofline",
            "value": N_global * args.steps / dt, "unit": "atom-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "host_enqueue_ms_per_step": t_host / args.steps * 1e3,
            "higher_is_better": True, "scaling": scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "dtype_note": "fp32 inputs, outputs, accumulation and elementwise math everywhere; the node-level matrix products run as "
                          "exact three-way bf16 splits of both fp32 operands (six partial products, fp32 accumulate: fp32-equivalent, "
                          "see mfma.arithmetic); parity tests at the reference tolerance 1e-5 unchanged",
            "config": {"workload": (names[cfg] % N_global) + ", HVNet rc=5.0 hidden=128 num_rbf=128 layers=5, E=%d "
                                   "directed edges, energy+forces per step" % E_global,
                       "config": cfg, "atoms_total": N_global, "atoms_owned_rank0": N - halo, "halo_atoms_rank0": halo,
                       "edges_rank0": E, "tuned_gemm_table": bool(tuned), "gemm_tuning_in_warmup": bool(online_tuning),
                       "parallelism": ("1 GPU, self-peer plan: %d virtual slab(s), the halo rows exchanged with this rank itself "
                                       "over %s (all_to_all_single, send_counts = recv_counts = [%d])"
                                       % (args.self_peer, args.backend, halo)) if args.self_peer else
                       "1 GPU" if world == 1 else
                       "atom-sharded x%d slabs (slab-local planning), one-hop halo all-to-all per layer over %s"
                       % (world, args.backend)},
            "roofline": {"bound": limiter[dom], "kernel": dom, "achieved": kernels[dom]["GBps"], "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": kernels[dom]["GBps"] / HBM_PEAK_GBS, "traffic": traffic,
                         "note": "achieved/peak/frac are ALGORITHMIC HBM bytes per launch / HIP-event time vs the 8 TB/s "
                                 "HBM peak; `bound` names what limits the kernel in practice"},
            "mfma": mfma,
            "kernels": kernels,
            "energy": float(e.detach()[0]),
            "library": _lib.build_info(),
        }
        if comm is not None:
            out["comm"] = comm
        if md is not None:
            out["secondary"] = {"atom_steps_per_s_incl_planning": N_global / md, "ms_per_step_incl_planning": md * 1e3,
                                "note": ("displacement check + slab-local device neighbour search (edges into owned "
                                         "atoms only) + relation build + energy + forces per step, max over ranks") if sharded else
                                        "device cell-list neighbour search + relation build + energy + forces per step"}
            out["secondary"]["incl_planning_over_step"] = md / (dt / args.steps)
            if md_padded is not None:
                out["secondary"]["incl_planning_padded_list"] = md_padded
            if sharded:
                out["secondary"].update({"skin_A": SKIN, "replan_ms": replan_ms, "replans_in_run": stepper.replans,
                                         "host_reads_per_step": 0, "last_step_valid": md_valid[-1] if md_valid else None,
                                         "steps_repeated": stepper.repeats,
                                         "planning_note": "plan reused under the Verlet skin (static coordinates here); a "
                                                          "re-plan costs replan_ms more on the step that needs it"})
        if sharded and not args.no_secondary and cfg != "weak":
            # the strong-scaling baseline inside the same run: the same cell, unsharded, on rank 0's GPU
            try:
                d1 = synth.fcc_alloy(reps=reps, seed=0, device=dev)
                for _ in range(2):
                    step(d1)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(5):
                    step(d1)
                torch.cuda.synchronize()
                one = (time.perf_counter() - t1) / 5
                out["secondary"]["single_gpu_same_cell"] = {"ms_per_step": one * 1e3, "atom_steps_per_s": N_global / one}
                out["secondary"]["speedup_vs_single_gpu"] = one / (dt / args.steps)
                # (top level too: `--gpus 1` runs configs[1], the cell the metric is quoted on, `--gpus N` this cell -- a ratio
                # of the two lines' `value`s would compare different workloads; this is the same cell on ONE GPU of this run)
                out["single_gpu_same_workload"] = {"value": N_global / one, "unit": "atom-steps/s", "ms_per_step": one * 1e3,
                                                   "where": "rank 0's GPU, unsharded, timed in this process"}
                out["speedup_vs_single_gpu_same_workload"] = one / (dt / args.steps)
                del d1
            except Exception as ex:
                out["secondary"]["single_gpu_same_cell"] = {"error": repr(ex)}
        if world == 1 and not sharded and not args.no_secondary:
            try:      # kernel launches of ONE step, counted by the profiler (every device kernel, torch's included)
                out["secondary"]["launches_per_step"], out["secondary"]["launches_per_step_own_kernels"] = count_launches(step)
            except Exception as ex:
                out["secondary"]["launches_per_step"] = {"error": repr(ex)}
            try:
                out["secondary"]["sharded_rank_of_8_self_peer"] = self_peer_secondary()
            except Exception as ex:
                out["secondary"]["sharded_rank_of_8_self_peer"] = {"error": repr(ex)}
            try:
                out["secondary"]["decomposition_at_8_ranks"] = decomposition_secondary(synth, dev, model_kw["rc"])
            except Exception as ex:
                out["secondary"]["decomposition_at_8_ranks"] = {"error": repr(ex)}
            if cfg == "c2":
                try:
                    out["secondary"]["reference_default_width_h512"] = reference_default_width_secondary(
                        hn, synth, dev, data, model_kw)
                except Exception as ex:
                    out["secondary"]["reference_default_width_h512"] = {"error": repr(ex)}
            # the same step captured once and replayed as ONE hipGraph launch (valid while the neighbour list is
            # unchanged, hermnet_amd/graph.py): what the GPU needs when the host is out of the loop
            try:
                out["secondary"]["graph_replay"] = graph_replay_secondary(hn, synth, dev, model, data, model_kw)
            except Exception as ex:
                out["secondary"]["graph_replay"] = {"error": repr(ex)}
            try:   # the box's own copy bandwidth next to the 8 TB/s the roofline is priced against
                cp_k, cp_t = measured_copy_bandwidth(dev)
                out["roofline"]["measured_copy_GBps"] = cp_k
                out["roofline"]["measured_copy_GBps_torch_copy"] = cp_t
                out["roofline"]["frac_of_measured_copy"] = out["roofline"]["achieved"] / cp_k
            except Exception as ex:
                out["roofline"]["measured_copy_GBps"] = None
            # secondary figures: the other single-GPU configurations of BASELINE.json at full size, same model
            try:
                out["secondary"]["other_configs"] = other_configs_secondary(hn, synth, dev, model_kw,
                                                                            skip_c4=(cfg == "c4"))
            except Exception as ex:
                out["secondary"]["other_configs"] = {"error": repr(ex)}
            # secondary figure: one optimisation step of `example/dist_train.py:86-99`
            # (energy + force loss with create_graph=True, backward to all parameters, Adam) on configs[4]'s
            # molecule batch; runs the differentiable device-op path of train() mode
            try:
                out["secondary"]["training"] = training_secondary(hn, synth, dev, model_kw)
            except Exception as ex:      # never lose the headline line over the secondary figure
                out["secondary"]["training"] = {"error": repr(ex)}
        if world == 1 and not sharded and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model_kw, elems, seed)
        print(json.dumps(out), flush=True)
    if sharded:
        fence()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
