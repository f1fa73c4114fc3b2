"""CPU: the C-ABI library loads without a GPU, exports every symbol include/hermnet_hip.h
declares, and its banded radial contraction (the formulation the kernels use) agrees with the
dense reference formula (rmnet.py:55,168-172) and with finite differences."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from hermnet_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "hermnet_hip.h")).read()
    declared = set(re.findall(r"\b(hermnet_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES.keys()), declared ^ set(_lib.SIGNATURES.keys())
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.hermnet_abi_version() == _lib.ABI_VERSION == 13
    assert b"gfx950" in lib.hermnet_build_info()


def _dense(offset, inv_rc, coeff, env_kind, p, wt, b, d):
    """float64 dense evaluation of rbf_proj(env * gaussians) and its d-derivative."""
    u = d * inv_rc
    off = offset.astype(np.float64)
    if env_kind == 0:
        a_, b_, c_ = -(p + 1) * (p + 2) / 2, p * (p + 2), -p * (p + 1) / 2
        env = 1 + a_ * u ** p + b_ * u ** (p + 1) + c_ * u ** (p + 2)
        denv = a_ * p * u ** (p - 1) + b_ * (p + 1) * u ** p + c_ * (p + 2) * u ** (p + 1)
    else:
        q = -(u * u) / ((1 - u) * (1 + u))
        env = np.exp(q)
        denv = -2 * u / ((1 - u) * (1 + u)) ** 2 * env
    if u >= 1:
        env = denv = 0.0
    gk = np.exp(coeff * (u - off) ** 2)
    dgk = gk * 2 * coeff * (u - off)
    rb = b + env * (gk @ wt)
    drb = inv_rc * (denv * (gk @ wt) + env * (dgk @ wt))
    return rb, drb


def ref_env(u, env_kind, p):
    u = float(u)
    if env_kind == 0:
        return 1 - (p + 1) * (p + 2) / 2 * u ** p + p * (p + 2) * u ** (p + 1) - p * (p + 1) / 2 * u ** (p + 2)
    return np.exp(-(u * u) / ((1 - u) * (1 + u)))


@pytest.mark.parametrize("R,env_kind,p", [(128, 0, 5), (32, 0, 5), (64, 1, 0), (16, 0, 3)])
def test_banded_rbf_row_matches_dense(R, env_kind, p):
    lib = _lib.load()
    rs = np.random.RandomState(R)
    C, rc = 24, 5.0
    offset = torch.linspace(0, 1, R).numpy().copy()
    coeff = -0.5 / float(offset[1] - offset[0]) ** 2
    wt = (rs.uniform(-1, 1, size=(R, C)) / np.sqrt(R)).astype(np.float32)
    b = rs.uniform(-0.1, 0.1, size=C).astype(np.float32)
    rb = np.zeros(C, np.float32)
    drb = np.zeros(C, np.float32)
    P = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    worst = worst_d = worst32 = 0.0
    for d in list(rs.uniform(0.05, rc * 0.999, size=200)) + [1e-6, 0.01, rc * 0.9999, rc, rc * 1.5, 33.7, 1e4]:
        rcode = lib.hermnet_host_rbf_row(P(offset), R, 1.0 / rc, coeff, env_kind, p, P(wt), P(b), C,
                                         float(np.float32(d)), P(rb), P(drb))
        assert rcode == 0
        ref, dref = _dense(offset, np.float32(1.0 / rc), coeff, env_kind, p, wt.astype(np.float64), b.astype(np.float64),
                           float(np.float32(d)))
        assert np.all(np.isfinite(rb)) and np.all(np.isfinite(drb))
        scale = np.abs(ref).max()
        worst = max(worst, np.abs(rb - ref).max() / scale)
        # same dense sum in fp32 with the reference's operation order isolates the band truncation
        u32 = np.float32(d) * np.float32(1.0 / rc)
        if u32 < 1:
            g32 = np.exp(np.float32(coeff) * (u32 - offset) ** 2, dtype=np.float32)
            ref32 = b.astype(np.float64) + float(np.float32(ref_env(u32, env_kind, p))) * (g32.astype(np.float64) @ wt.astype(np.float64))
            worst32 = max(worst32, np.abs(rb - ref32).max() / scale)
        worst_d = max(worst_d, np.abs(drb - dref).max() / max(np.abs(dref).max(), 0.1))   # abs error floor: derivative -> 0 at the cutoff
        if d >= rc:   # beyond the cutoff only the bias is left
            assert np.array_equal(rb, b) and not drb.any()
    # dropped taps are < exp(-18) of the largest; the rest is fp32 rounding of u - offset[k]
    assert worst32 < 1e-6, worst32    # band truncation + summation order only
    assert worst < 2e-5, worst        # + fp32 rounding of (u - offset[k]) * (R-1), shared with the reference
    assert worst_d < 2e-4, worst_d


def test_banded_rbf_derivative_vs_finite_difference():
    lib = _lib.load()
    R, C, rc = 128, 8, 5.0
    rs = np.random.RandomState(0)
    offset = torch.linspace(0, 1, R).numpy().copy()
    coeff = -0.5 / float(offset[1] - offset[0]) ** 2
    wt = (rs.uniform(-1, 1, size=(R, C)) / np.sqrt(R)).astype(np.float64)
    b = np.zeros(C)
    for d in [0.7, 2.345, 4.2]:
        h = 1e-6
        rp, _ = _dense(offset, 1 / rc, coeff, 0, 5, wt, b, d + h)
        rm, _ = _dense(offset, 1 / rc, coeff, 0, 5, wt, b, d - h)
        _, dr = _dense(offset, 1 / rc, coeff, 0, 5, wt, b, d)
        assert np.allclose((rp - rm) / (2 * h), dr, rtol=1e-5, atol=1e-7)


def test_default_message_kernels_fit_their_register_budget(tmp_path):
    """The channel-per-lane backward must stay at 4 waves per SIMD (<= 128 VGPRs at 1024 threads) without scratch:
    a spill or a fifth-wave-less build is a silent 2x.  Checked on the code hipcc generates for gfx950 (no GPU)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not found")
    src = os.path.join(ROOT, "hermnet_amd", "csrc", "message_bwd_cl.hip")
    out = str(tmp_path / "cl.s")
    subprocess.run([hipcc, "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-slp-vectorize", "-S", "--cuda-device-only",
                    src, "-o", out], check=True, capture_output=True, timeout=600)
    text = open(out).read()
    kernels = re.findall(r"\.name:\s+(\S*message_scatter_bwd_cl_kernel\S*)(.*?)\.wavefront_size", text, flags=re.S)
    assert len(kernels) == 4        # {with, without vec rows} x {whole tile, tap-row windows (num_rbf > 137)}
    for name, meta in kernels:
        get = lambda key: int(re.search(r"\.%s:\s+(\d+)" % key, meta).group(1))
        assert get("vgpr_count") <= 128, (name, get("vgpr_count"))
        assert get("vgpr_spill_count") == 0, name
        if "ELb0EEE" in name:       # the whole-tile instances (the ones every default-sized model runs)
            assert get("sgpr_spill_count") == 0 and get("private_segment_fixed_size") == 0, name


def test_argument_checks_of_the_round_4_entry_points_need_no_gpu():
    """The entry points added with ABI v8 refuse malformed calls before anything is launched (and treat empty inputs as
    done), so their argument contracts can be checked here: hermnet_train_node_op (operand counts per op), the *_rows edge
    kernels, hn_pending_grads of hermnet_node_update_bwd."""
    lib = _lib.load()
    HN_OK, BAD = 0, lib.hermnet_train_node_op(0, None, 0, None, 0, 0, 4, 0.0, 0.0, None)
    assert BAD != HN_OK
    buf = np.zeros(64, dtype=np.float32)
    ptr = buf.ctypes.data
    arr = lambda n: (ctypes.c_void_p * max(n, 1))(*([ptr] * n))
    need_in = {1: 2, 2: 3, 3: 2, 4: 3, 5: 5, 6: 6, 7: 6, 8: 11, 9: 2, 10: 3, 11: 5, 12: 3}
    need_out = {1: 1, 2: 2, 3: 2, 4: 2, 5: 3, 6: 2, 7: 5, 8: 5, 9: 1, 10: 2, 11: 2, 12: 2}
    for op in need_in:
        ni, no = need_in[op], need_out[op]
        assert lib.hermnet_train_node_op(op, arr(ni), ni, arr(no), no, 0, 128, 0.0, 0.0, None) == HN_OK        # no rows: done
        assert lib.hermnet_train_node_op(op, arr(ni), ni + 1, arr(no), no, 0, 128, 0.0, 0.0, None) == BAD     # operand count
        assert lib.hermnet_train_node_op(op, arr(ni), ni, arr(no), no, 0, 130, 0.0, 0.0, None) == BAD         # width % 4
    assert lib.hermnet_train_node_op(13, arr(2), 2, arr(1), 1, 0, 128, 0.0, 0.0, None) == BAD
    # the row-sum edge kernels: a negative group count is refused, no groups = nothing to do
    assert lib.hermnet_edge_message_fwd_rows(ptr, ptr, None, ptr, 0, 128, None, None, None, ptr, None, -1, ptr, ptr, None) == BAD
    assert lib.hermnet_edge_message_fwd_rows(ptr, ptr, None, ptr, 0, 128, None, None, None, ptr, None, 0, ptr, ptr, None) == HN_OK
    assert lib.hermnet_edge_message_fwd_rows(ptr, ptr, None, ptr, 0, 130, None, None, None, ptr, None, 0, ptr, ptr, None) == BAD
    # hn_pending_grads: every pointer of the record is required
    rp = (ctypes.c_int * 2)(0, 5)
    pend = _lib.PendingGrads(ptr, None, ptr, ptr, ptr, ptr, ptr, 1, 0)
    assert lib.hermnet_node_update_bwd(ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, None, ptr, rp, ptr, ptr, 5, 1, 128, 0,
                                       ctypes.byref(pend), None) == BAD
    assert lib.hermnet_node_update_bwd(ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, ptr, None, ptr, rp, ptr, ptr, 0, 1, 128, 0,
                                       None, None) == HN_OK                                                 # no rows


def test_argument_checks_of_the_projected_halo_entry_points_need_no_gpu():
    """hermnet_halo_proj_rows / hermnet_halo_proj_accumulate (ABI v12): malformed calls are refused before a launch, an empty
    row list is done."""
    lib = _lib.load()
    buf = np.zeros(64, dtype=np.float32)
    ptr = buf.ctypes.data
    HN_OK, BAD = 0, 1
    rows = lambda mode, nseg, nsum, n, width, a=ptr: lib.hermnet_halo_proj_rows(mode, a, 0, nseg, ptr, 0, nsum, ptr, n, width, ptr, None)
    assert rows(0, 3, 1, 0, 384) == HN_OK                       # no rows: done (whatever the pointers)
    assert rows(0, 3, 1, 0, 384, a=None) == HN_OK
    assert rows(0, 3, 1, -1, 384) == BAD                        # negative count
    assert rows(0, 3, 1, 0, 386) == BAD and rows(0, 3, 1, 0, 0) == BAD          # width % 4, width <= 0
    assert rows(0, 0, 1, 0, 384) == BAD and rows(1, 3, 0, 0, 384) == BAD        # at least one block and one slice
    assert rows(2, 3, 3, 0, 384) == BAD                         # unpack takes b without a slice axis
    assert rows(3, 3, 1, 4, 384) == BAD                         # unknown mode (checked after the pointers, before a launch)
    assert rows(0, 3, 1, 4, 384, a=None) == BAD                 # rows but no source
    acc = lambda nseg, nu, width, a=ptr: lib.hermnet_halo_proj_accumulate(a, 0, nseg, ptr, ptr, ptr, ptr, nu, width, ptr, None)
    assert acc(3, 0, 384) == HN_OK and acc(3, 0, 384, a=None) == HN_OK
    assert acc(3, -1, 384) == BAD and acc(0, 0, 384) == BAD and acc(3, 0, 382) == BAD
    assert acc(3, 2, 384, a=None) == BAD


def test_argument_checks_of_the_training_entry_points_of_abi_13_need_no_gpu():
    """hermnet_band_product / _grad_a / _grad_b / _grads, hermnet_basis_window, hermnet_edge_unit: malformed calls are refused
    before a launch; nothing to do (no chunks, no edges) is done."""
    lib = _lib.load()
    buf = np.zeros(64, dtype=np.float32)
    ptr = buf.ctypes.data
    OK, BAD = 0, 1
    assert lib.hermnet_band_product_supported(1024, 384) == 1 and lib.hermnet_band_product_supported(64, 96) == 1
    assert lib.hermnet_band_product_supported(1000, 384) == 0 and lib.hermnet_band_product_supported(1024, 100) == 0
    assert lib.hermnet_band_product_supported(1024, 512) == 0                     # wider than the weight window's LDS image
    prod = lambda nc, C, N, a=ptr: lib.hermnet_band_product(a, ptr, None, nc, C, N, ptr, None)
    assert prod(0, 1024, 384) == OK and prod(0, 1024, 384, a=None) == OK
    assert prod(-1, 1024, 384) == BAD and prod(0, 1000, 384) == BAD and prod(0, 1024, 100) == BAD and prod(2, 1024, 384, a=None) == BAD
    ga = lambda nc, C, N, g=ptr: lib.hermnet_band_product_grad_a(g, None, ptr, nc, C, N, ptr, None)
    assert ga(0, 64, 96) == OK and ga(0, 60, 96) == BAD and ga(3, 64, 96, g=None) == BAD
    gb = lambda nc, C, N, a=ptr: lib.hermnet_band_product_grad_b(a, ptr, None, nc, C, N, ptr, None, None)
    assert gb(0, 64, 96) == OK and gb(0, 64, 98) == BAD and gb(3, 64, 96, a=None) == BAD
    gs = lambda nc, C, N, a=ptr: lib.hermnet_band_product_grads(a, ptr, ptr, None, nc, C, N, ptr, ptr, None, None)
    assert gs(0, 1024, 384) == OK and gs(0, 1024, 385) == BAD and gs(1, 1024, 384, a=None) == BAD
    bw = lambda order, nc, p, u=ptr: lib.hermnet_basis_window(order, u, ptr, 5, ptr, ptr, nc, 1024, -1.0, p, ptr, ptr, ptr, ptr, None)
    assert bw(0, 0, 5) == OK and bw(3, 0, 5) == BAD and bw(0, 0, 0) == BAD and bw(0, -1, 5) == BAD and bw(1, 2, 5, u=None) == BAD
    eu = lambda order, E, D=ptr: lib.hermnet_edge_unit(order, D, None, None, ptr, E, ptr, ptr, ptr, None)
    assert eu(0, 0) == OK and eu(3, 0) == BAD and eu(0, -1) == BAD and eu(0, 4, D=None) == BAD


@pytest.mark.parametrize("out_f,in_f", [(128, 128), (384, 128), (128, 256), (64, 64), (32, 96)])
def test_weight_fragments_through_the_c_abi_equal_the_host_codes(out_f, in_f):
    """a binder of the C seam must not re-implement the three-plane weight stream.  `hermnet_weight_fragments`
    (host pointers in and out) gives frag(W) (tile_rows 32) and frag16(W) (16) bit for bit as `nodeops.weight_fragments` /
    `weight_fragments16` build them in torch ops -- values that are bf16 ties, denormals, huge, tiny and negative zeros
    included -- and the planes sum back to the weight exactly."""
    from hermnet_amd import nodeops
    lib = _lib.load()
    gen = torch.Generator().manual_seed(out_f * 1000 + in_f)
    w = torch.randn(out_f, in_f, generator=gen) * (10.0 ** torch.randint(-6, 7, (out_f, 1), generator=gen).float())
    w[0, :8] = torch.tensor([1.00390625, -1.00390625, 1.01171875, 3.0e38, -3.0e38, 1.0e-38, 1.0e-41, -0.0])   # ties, extremes, a denormal
    w = w.contiguous()
    for tile_rows, ref_fn in ((32, nodeops.weight_fragments), (16, nodeops.weight_fragments16)):
        if out_f % tile_rows or in_f % (16 if tile_rows == 32 else 32):
            out = np.zeros(out_f * in_f * 3, dtype=np.uint16)
            assert lib.hermnet_weight_fragments(w.data_ptr(), out_f, in_f, tile_rows, out.ctypes.data) == 1     # HN_ERR_BAD_ARG
            continue
        out = np.zeros(out_f * in_f * 3, dtype=np.uint16)
        assert lib.hermnet_weight_fragments(w.data_ptr(), out_f, in_f, tile_rows, out.ctypes.data) == 0
        ref = ref_fn(w).view(torch.int16).numpy().view(np.uint16)
        assert np.array_equal(out, ref), (tile_rows, int((out != ref).sum()))
    # the planes are an exact split: p0 + p1 + p2 == w (fp32 arithmetic, largest first), barring under / overflow of a residual
    planes = [p.float() for p in nodeops._bf16_planes(w)]          # smallest first
    normal = w.abs() >= 1.0e-30                                    # (a denormal's residual planes underflow: documented)
    assert torch.equal(((planes[2] + planes[1]) + planes[0])[normal], w[normal])
    assert lib.hermnet_weight_fragments(None, 32, 32, 32, out.ctypes.data) == 1


def test_library_options_round_trip():
    """hermnet_set_option / hermnet_get_option (ABI v12: they replace the environment variables the library read once per
    process): defaults as documented, a value set is the value read, unknown options are refused."""
    lib = _lib.load()
    assert _lib.get_option("fwd_variant") == 8420 and _lib.get_option("fwd_variant_l0") == 16420
    assert _lib.get_option("update_tile16") == 2 and _lib.get_option("bwd_lanes16") == 0
    with _lib.options(fwd_rows=17, node_chain_wide=1):
        assert _lib.get_option("fwd_rows") == 17 and _lib.get_option("node_chain_wide") == 1
    assert _lib.get_option("fwd_rows") == 0 and _lib.get_option("node_chain_wide") == 0
    assert lib.hermnet_set_option(99, 1) == 1 and lib.hermnet_set_option(-1, 1) == 1
    hdr = open(os.path.join(ROOT, "include", "hermnet_hip.h")).read()
    opts = dict((n.lower(), int(v)) for n, v in re.findall(r"#define HN_OPT_([A-Z0-9_]+) (\d+)", hdr))
    assert opts == _lib.OPTIONS and int(re.search(r"#define HN_NUM_OPTIONS (\d+)", hdr).group(1)) == len(opts)
