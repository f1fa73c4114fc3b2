"""Thin Python wrappers (allocate outputs, pass pointers) around the node-level fused
kernels of `csrc/node_kernels.hip`.  One call = one kernel launch on the current stream."""

import ctypes

import torch

from . import _lib
from .ops import _launch, _stream

P = _lib.ptr


def layernorm_fwd(x, eps=1e-5, h_real=0):
    """LayerNorm without affine over the last axis -> (n, mean [rows], rstd [rows]).  `h_real` (0 = all): width of
    the statistics when the rows are zero-padded (padded outputs are zero)."""
    rows, H = x.shape
    n = torch.empty_like(x)
    mean = torch.empty(rows, dtype=x.dtype, device=x.device)
    rstd = torch.empty(rows, dtype=x.dtype, device=x.device)
    _lib.check(_launch("layernorm_fwd", lambda: _lib.load().hermnet_layernorm_fwd(
        P(x), P(n), P(mean), P(rstd), rows, H, h_real, eps, _stream())), "hermnet_layernorm_fwd")
    return n, mean, rstd


def layernorm_bwd(g, x, mean, rstd, add=None, h_real=0):
    """Gradient of layernorm_fwd w.r.t. x, plus `add`."""
    rows, H = x.shape
    gx = torch.empty_like(x)
    _lib.check(_launch("layernorm_bwd", lambda: _lib.load().hermnet_layernorm_bwd(
        P(g), P(x), P(mean), P(rstd), P(add), P(gx), rows, H, h_real, _stream())), "hermnet_layernorm_bwd")
    return gx


def ssilu_fwd(h, bias=None, rows_per_bias=0):
    """a = ScaledSiLU(h + bias); h [rows, cols]; bias [groups, cols] (group = row // rows_per_bias) or None."""
    a = torch.empty_like(h)
    cols = h.size(-1)
    _lib.check(_launch("ssilu_fwd", lambda: _lib.load().hermnet_ssilu_fwd(
        P(h), P(bias), rows_per_bias, P(a), h.numel() // cols, cols, _stream())), "hermnet_ssilu_fwd")
    return a


def ssilu_bwd(g, h, N, T, C, gs_n, gs_t, bias=None, rows_per_bias=0):
    gh = torch.empty(N, T * C, dtype=h.dtype, device=h.device)
    _lib.check(_launch("ssilu_bwd", lambda: _lib.load().hermnet_ssilu_bwd(
        P(g), P(h), P(bias), rows_per_bias, P(gh), N, T, C, gs_n, gs_t, _stream())), "hermnet_ssilu_bwd")
    return gh


def update_mid(vp, x1, rows, H):
    vdot = torch.empty(x1.size(0), H, dtype=x1.dtype, device=x1.device)
    xin = torch.empty(x1.size(0), 2 * H, dtype=x1.dtype, device=x1.device)
    _lib.check(_launch("update_mid", lambda: _lib.load().hermnet_update_mid(P(vp), P(x1), P(vdot), P(xin), rows, H,
                                                                            _stream())), "hermnet_update_mid")
    return vdot, xin


def update_out(q, vdot, vp, x1, vec1, mask, N, nk, H, qbias=None, rows_per_bias=0):
    xo = torch.empty(N, H, dtype=x1.dtype, device=x1.device)
    vo = torch.empty(N, 3, H, dtype=x1.dtype, device=x1.device)
    _lib.check(_launch("update_out", lambda: _lib.load().hermnet_update_out(
        P(q), P(qbias), rows_per_bias, P(vdot), P(vp), P(x1), P(vec1), P(mask), P(xo), P(vo), N, nk, H, _stream())),
        "hermnet_update_out")
    return xo, vo


def update_out_bwd(gxo, gvo, q, vdot, vp, mask, N, nk, H, qbias=None, rows_per_bias=0):
    dev, dt = gxo.device, gxo.dtype
    gq = torch.empty(N, 3 * H, dtype=dt, device=dev)
    gvdot = torch.empty(N, H, dtype=dt, device=dev)
    gvp = torch.empty(N, 3, 2 * H, dtype=dt, device=dev)
    gx1 = torch.empty(N, H, dtype=dt, device=dev)
    gvec1 = torch.empty(N, 3, H, dtype=dt, device=dev)
    _lib.check(_launch("update_out_bwd", lambda: _lib.load().hermnet_update_out_bwd(
        P(gxo), P(gvo), P(q), P(qbias), rows_per_bias, P(vdot), P(vp), P(mask), P(gq), P(gvdot), P(gvp), P(gx1), P(gvec1),
        N, nk, H,
        _stream())), "hermnet_update_out_bwd")
    return gq, gvdot, gvp, gx1, gvec1


def update_mid_bwd(gvdot, gxin, vp, xin, gvp, gx1, rows, H):
    _lib.check(_launch("update_mid_bwd", lambda: _lib.load().hermnet_update_mid_bwd(
        P(gvdot), P(gxin), P(vp), P(xin), P(gvp), P(gx1), rows, H, _stream())), "hermnet_update_mid_bwd")


def energy_head_fwd(h, w, b, mask=None):
    """e[n] = (sum_c ScaledSiLU(h[n,c]) w[c] + b[0]) * mask[n]  (b: 1-element device tensor or None)."""
    rows, C = h.shape
    e = torch.empty(rows, dtype=h.dtype, device=h.device)
    _lib.check(_launch("energy_head_fwd", lambda: _lib.load().hermnet_energy_head_fwd(
        P(h), P(w), P(b), P(mask), P(e), rows, C, _stream())), "hermnet_energy_head_fwd")
    return e


def energy_head_bwd(ge, h, w, mask=None):
    rows, C = h.shape
    gh = torch.empty_like(h)
    _lib.check(_launch("energy_head_bwd", lambda: _lib.load().hermnet_energy_head_bwd(
        P(ge), P(h), P(w), P(mask), P(gh), rows, C, _stream())), "hermnet_energy_head_bwd")
    return gh


def head_fused_supported(H, C):
    return C in (64, 128, 256) and H in (64, 128, 256) and H * C * 4 <= 65536


def energy_head_fused_fwd(x, w0t, b0, w2, b2, mask=None):
    """x [rows,H] -> (h [rows,C] pre-activation, e [rows]) in one launch (no library GEMM); w0t = out_energy[0].weight^T."""
    rows, H = x.shape
    C = w0t.size(1)
    h = torch.empty(rows, C, dtype=x.dtype, device=x.device)
    e = torch.empty(rows, dtype=x.dtype, device=x.device)
    _lib.check(_launch("energy_head_fused_fwd", lambda: _lib.load().hermnet_energy_head_fused_fwd(
        P(x), P(w0t), P(b0), P(w2), P(b2), P(mask), P(h), P(e), rows, H, C, _stream())), "hermnet_energy_head_fused_fwd")
    return h, e


def energy_head_fused_bwd(ge, h, w0, w2, mask=None):
    rows, C = h.shape
    H = w0.size(1)
    gx = torch.empty(rows, H, dtype=h.dtype, device=h.device)
    _lib.check(_launch("energy_head_fused_bwd", lambda: _lib.load().hermnet_energy_head_fused_bwd(
        P(ge), P(h), P(w0), P(w2), P(mask), P(gx), rows, H, C, _stream())), "hermnet_energy_head_fused_bwd")
    return gx


def head16_supported(H, C):
    return bool(_lib.load().hermnet_energy_head16_supported(int(H), int(C)))


def energy_head16_fwd(x, w0f16, b0, w2, b2, mask=None):
    """The read-out with its H -> C product on the matrix pipe (16-row tiles, H = 128, C = 64): (h [rows,C], e [rows])."""
    rows, H = x.shape
    C = b0.numel()
    h = torch.empty(rows, C, dtype=x.dtype, device=x.device)
    e = torch.empty(rows, dtype=x.dtype, device=x.device)
    _lib.check(_launch("energy_head16_fwd", lambda: _lib.load().hermnet_energy_head16_fwd(
        P(x), P(w0f16), P(b0), P(w2), P(b2), P(mask), P(h), P(e), rows, H, C, _stream())), "hermnet_energy_head16_fwd")
    return h, e


def energy_head16_bwd(ge, h, w0tf16, w2, H, mask=None):
    rows, C = h.shape
    gx = torch.empty(rows, H, dtype=h.dtype, device=h.device)
    _lib.check(_launch("energy_head16_bwd", lambda: _lib.load().hermnet_energy_head16_bwd(
        P(ge), P(h), P(w0tf16), P(w2), P(mask), P(gx), rows, H, C, _stream())), "hermnet_energy_head16_bwd")
    return gx


def pair_mean(x, vec, Te, P_, B, rows_out, backward=False):
    """HTNet: mean over a centre atom's P virtual target rows (backward=False: [Te*P*B, ...] -> [rows_out, ...], zero
    rows behind Te*B) or its gradient (backward=True: [rows_out, ...] -> [Te*P*B, ...])."""
    H = x.size(1)
    rows = Te * P_ * B if backward else rows_out
    xo = torch.empty(rows, H, dtype=x.dtype, device=x.device)
    vo = torch.empty(rows, 3, H, dtype=x.dtype, device=x.device)
    _lib.check(_launch("pair_mean", lambda: _lib.load().hermnet_pair_mean(
        1 if backward else 0, P(x), P(vec), P(xo), P(vo), Te, P_, B, rows_out, H, 1.0 / P_, 1.0 / P_, None, 0, _stream())),
        "hermnet_pair_mean")
    return xo, vo


def pair_sum_accumulate(x, vec, x_acc, vec_acc, Te, P_, B, scale_x, scale_vec, ranges=None):
    """x_acc[c*B + i] += scale_x * sum_k x[(c*P + k)*B + i] (and vec likewise): the residual's gradient of HTNet's virtual
    target rows, one launch.  `ranges` = (device [k,2] int32, host list): only these rows of x_acc / vec_acc."""
    H = x.size(1)
    rd, nr = (None, 0) if ranges is None else (ranges[0], len(ranges[1]))
    _lib.check(_launch("pair_sum", lambda: _lib.load().hermnet_pair_mean(
        2, P(x), P(vec), P(x_acc), P(vec_acc), Te, P_, B, x_acc.size(0), H, scale_x, scale_vec, P(rd), nr, _stream())),
        "hermnet_pair_mean")


def halo_rows(mode, x, vec, idx, buf=None):
    """Packed halo rows [n, 4H] = [x | vec] of the rows `idx` (int64): mode 0 pack, 1 pack-and-clear, 2 unpack
    (see include/hermnet_hip.h).  Returns `buf` (allocated for the packing modes)."""
    n, H = int(idx.numel()), x.size(1)
    if buf is None:
        buf = torch.empty(n, 4 * H, dtype=x.dtype, device=x.device)
    _lib.check(_launch("halo_rows", lambda: _lib.load().hermnet_halo_rows(
        mode, P(x), P(vec), P(idx), n, H, P(buf), _stream())), "hermnet_halo_rows")
    return buf


def halo_accumulate(x, vec, plan, buf):
    """rows[plan.acc_rows[u]] += sum of the packed rows `buf[plan.acc_pos[q]]` of segment u, in list order
    (ordered sums: no float atomics anywhere on the path)."""
    rows, ptr, pos = plan.accumulate_lists()
    _lib.check(_launch("halo_accumulate", lambda: _lib.load().hermnet_halo_accumulate(
        P(x), P(vec), P(rows), P(ptr), P(pos), int(rows.numel()), x.size(1), P(buf), _stream())), "hermnet_halo_accumulate")


def halo_proj_rows(mode, a, b, idx, buf=None):
    """The exchange of PROJECTED halo rows (csrc/node_kernels.hip: hermnet_halo_proj_rows): a packed row holds the S = a.size(0)
    blocks a[j][r] and one block  sum_s b[s][r]  (or b[r] when b has no slice axis), each W = a.size(2) floats.  Forward:
    a = xh [T, N, 3H], b = vec [N, 3, H]; backward: a = gxh, b = the per-relation partial sums of gvec [T, N, 3, H].
    mode 0 pack, 1 pack and clear every source, 2 unpack (b without a slice axis).  Returns `buf` [n, (S + 1) W]."""
    S, N, W = a.shape
    sliced = b.dim() == 4
    nsum = b.size(0) if sliced else 1
    n = int(idx.numel())
    if buf is None:
        buf = torch.empty(n, (S + 1) * W, dtype=a.dtype, device=a.device)
    _lib.check(_launch("halo_rows", lambda: _lib.load().hermnet_halo_proj_rows(
        mode, P(a), N * W, S, P(b), N * W, nsum, P(idx), n, W, P(buf), _stream())), "hermnet_halo_proj_rows")
    return buf


def halo_proj_accumulate(a, b, plan, buf):
    """a[j][rows[u]] += the sum of block j of the returned rows of segment u, b[0][rows[u]] (or b[rows[u]]) += the sum of their
    last blocks, in list order (`plan.accumulate_lists()`): the owner's side of the gradient return, no atomics."""
    rows, ptr, pos = plan.accumulate_lists()
    S, N, W = a.shape
    _lib.check(_launch("halo_accumulate", lambda: _lib.load().hermnet_halo_proj_accumulate(
        P(a), N * W, S, P(b), P(rows), P(ptr), P(pos), int(rows.numel()), W, P(buf), _stream())), "hermnet_halo_proj_accumulate")


# ---- node chain kernels (csrc/node_chain.hip): one launch per chain on the fp32 matrix pipe ---------------------------
def chain_supported(H):
    """Widths the chain kernels are instantiated for: every multiple of 64 up to 512 (csrc/node_chain.hip for 64 / 128 /
    256, csrc/node_chain_wide.hip for the rest).  Anything else -- i.e. padded widths beyond 512 -- takes library GEMMs +
    the stage kernels above."""
    return bool(_lib.load().hermnet_node_chain_supported(int(H)))


def chain_tile_rows(H, update=False):
    """Rows per tile of the pre (or update) chain kernels at width H: the granularity of the row windows."""
    return int(_lib.load().hermnet_node_chain_tile_rows(int(H), 1 if update else 0))


def _bf16_planes(w):
    """w = p0 + p1 + p2 exactly (bf16 planes, each the round-to-nearest-even of what the planes before it leave): the three-way
    split behind the chain kernels' fp32 products on the bf16 matrix pipe (csrc/node_chain_common.h: split8).  Returned smallest
    first -- the order the kernels' weight streams hold them in."""
    w32 = w.float()
    p0 = w32.to(torch.bfloat16)
    r1 = w32 - p0.float()
    p1 = r1.to(torch.bfloat16)
    p2 = (r1 - p1.float()).to(torch.bfloat16)
    return [p2, p1, p0]


def weight_fragments(w):
    """nn.Linear weight [out, in] (or a stack [T, out, in]) -> the chain kernels' weight stream (include/hermnet_hip.h: frag(W);
    csrc/node_chain_common.h: mma_panel): per 32-row block cb of W and 16-deep k-group Q the three bf16 planes of the weights,
    smallest first, as v_mfma_f32_32x32x16_bf16 operands:
        frag(W)[((cb * K/16 + Q) * 3 + s) * 64 + l] = 8 bf16  W_(2-s)[32 cb + (l & 31)][16 Q + 8 (l >> 5) .. +7]
    out % 32 == 0, in % 16 == 0 -> float32 [..., out * in * 3 / 2] (6 bytes per weight, opaque)."""
    lead = w.shape[:-2]
    o, k = w.shape[-2:]
    n = len(lead)
    f = torch.stack(_bf16_planes(w), dim=n).reshape(*lead, 3, o // 32, 32, k // 16, 2, 8)     # [s, cb, m, Q, g, e]
    f = f.permute(*range(n), n + 1, n + 3, n, n + 4, n + 2, n + 5).contiguous()                # [cb, Q, s, g, m, e]
    return f.reshape(*lead, o * k * 3).view(torch.float32)


def weight_fragments16(w):
    """The same for the 16-row kernels (csrc/node_chain16.hip: mma16_panel; include/hermnet_hip.h: frag16(W)): per 16-row block b
    and 32-deep k-group Q the three planes as v_mfma_f32_16x16x32_bf16 operands:
        frag16(W)[((b * K/32 + Q) * 3 + s) * 64 + l] = 8 bf16  W_(2-s)[16 b + (l & 15)][32 Q + 8 (l >> 4) .. +7]
    out % 16 == 0, in % 32 == 0 -> float32 [..., out * in * 3 / 2]."""
    lead = w.shape[:-2]
    o, k = w.shape[-2:]
    n = len(lead)
    f = torch.stack(_bf16_planes(w), dim=n).reshape(*lead, 3, o // 16, 16, k // 32, 4, 8)     # [s, b, m, Q, g, e]
    f = f.permute(*range(n), n + 1, n + 3, n, n + 4, n + 2, n + 5).contiguous()                # [b, Q, s, g, m, e]
    return f.reshape(*lead, o * k * 3).view(torch.float32)


def update_tile_rows(graph, H):
    """16 or the default tile height: which form of the update kernels shortens the launch for this row layout
    (cached on the graph: it depends on the row counts only)."""
    tr = getattr(graph, "_upd_tile", None)
    if tr is None or tr[0] != H:
        tr = (H, int(_lib.load().hermnet_node_update_tile_rows(_rowptr_host(graph), graph.N, graph.T, H)))
        try:
            graph._upd_tile = tr
        except AttributeError:
            pass
    return tr[1]


def _rowptr_host(graph):
    import ctypes
    c = getattr(graph, "_rowptr_c", None)
    if c is None:
        vals = list(graph.type_rowptr_host)
        c = (ctypes.c_int * len(vals))(*vals)
        try:
            graph._rowptr_c = c
        except AttributeError:
            pass
    return c


def node_pre_fwd(x, w, T, src_ranges=None, windows=None, mode=0, out=None):
    """x [Ns,H] -> (hb [T,Ns,H], xh [T,Ns,3H] incl. bias, mean [Ns], rstd [Ns])  (rmnet.py:52 for every relation).
    `src_ranges` [T,4] int32 (HTNet): the two source-row ranges a relation gathers from; other rows are skipped.
    `windows` [W,2] int32 + `mode` (atom shards): 1 = only the row tiles that touch a window, 2 = only the others;
    the second of the two calls passes the first one's result as `out`."""
    Ns, H = x.shape
    dev, dt = x.device, x.dtype
    if out is None:
        hb = torch.empty(T, Ns, H, dtype=dt, device=dev)
        xh = torch.empty(T, Ns, 3 * H, dtype=dt, device=dev)
        alloc = torch.empty if src_ranges is None else torch.zeros      # rows no relation wants keep (0, 0): finite
        mean = alloc(Ns, dtype=dt, device=dev)
        rstd = alloc(Ns, dtype=dt, device=dev)
    else:
        hb, xh, mean, rstd = out
    nwin = 0 if windows is None else int(windows.size(0))
    _lib.check(_launch("node_pre_fwd", lambda: _lib.load().hermnet_node_pre_fwd(
        P(x), P(w.w1f), P(w.b1cat), P(w.w2f), P(w.b2), P(hb), P(xh), P(mean), P(rstd), P(src_ranges), Ns, T, H, w.h_real,
        1e-5, P(windows), nwin, mode if windows is not None else 0, _stream())), "hermnet_node_pre_fwd")
    return hb, xh, mean, rstd


def node_pre_bwd(gxh, hb, x, mean, rstd, w, add=None, src_ranges=None, windows=None, mode=0, out=None, parts_only=False):
    """Gradient of node_pre_fwd w.r.t. x (+ add).  `windows` / `mode` / `out` as in node_pre_fwd (`out` = (gx, parts) of
    the first call).  `parts_only`: no LayerNorm-backward launch -- returns the per-relation partial sums [T,Ns,H] for a
    consumer that runs it itself (node_update_bwd's `pending`)."""
    T, Ns, H = hb.shape
    if parts_only:
        parts = torch.empty(T, Ns, H, dtype=x.dtype, device=x.device)
        _lib.check(_launch("node_pre_bwd", lambda: _lib.load().hermnet_node_pre_bwd(
            P(gxh), P(hb), P(w.w2tf), P(w.w1tf), P(parts), None, None, None, None, None, P(src_ranges), Ns, T, H, w.h_real,
            None, 0, 0, _stream())), "hermnet_node_pre_bwd")
        return parts
    if out is None:
        parts = torch.empty(T, Ns, H, dtype=x.dtype, device=x.device)
        gx = torch.empty_like(x)
    else:
        gx, parts = out
    nwin = 0 if windows is None else int(windows.size(0))
    _lib.check(_launch("node_pre_bwd", lambda: _lib.load().hermnet_node_pre_bwd(
        P(gxh), P(hb), P(w.w2tf), P(w.w1tf), P(parts), P(x), P(mean), P(rstd), P(add), P(gx), P(src_ranges), Ns, T, H, w.h_real,
        P(windows), nwin, mode if windows is not None else 0, _stream())), "hermnet_node_pre_bwd")
    return gx if out is None and windows is None else (gx, parts)


def node_update_fwd(x1, vec1, w, graph):
    """(x1, vec1) -> (x_out, vec_out) and the saved (vp [N,3,2H], h2b [N,H], q23 [N,2H], nrm [N,H])
    (rmnet.py:94-107, 29-31)."""
    N, H = x1.shape
    dev, dt = x1.device, x1.dtype
    t16 = update_tile_rows(graph, H) == 16 and getattr(w, "wvf16", None) is not None
    wvf, wx0f, wx2f = (w.wvf16, w.wx0f16, w.wx2f16) if t16 else (w.wvf, w.wx0f, w.wx2f)
    vp = torch.empty(N, 3, 2 * H, dtype=dt, device=dev)
    h2b = torch.empty(N, H, dtype=dt, device=dev)
    q23 = torch.empty(N, 2 * H, dtype=dt, device=dev)
    nrm = torch.empty(N, H, dtype=dt, device=dev)
    xo = torch.empty(N, H, dtype=dt, device=dev)
    vo = torch.empty(N, 3, H, dtype=dt, device=dev)
    _lib.check(_launch("node_update_fwd", lambda: _lib.load().hermnet_node_update_fwd(
        P(x1), P(vec1), P(wvf), P(wx0f), P(w.bx0_s), P(wx2f), P(w.bx2_s), P(graph.row_active), P(graph.type_rowptr),
        _rowptr_host(graph), P(vp), P(h2b), P(q23), P(nrm), P(xo), P(vo), N, graph.T, H, 16 if t16 else 0, _stream())),
        "hermnet_node_update_fwd")
    return xo, vo, vp, h2b, q23, nrm


class PendingGrads(object):
    """The gradients a layer's backward hands DOWN while they still sit in partial sums (hn_pending_grads): `gx` / `gvec`
    are the buffers the consumer -- the update backward of the layer below -- fills before it reads them.
    `chain` = (gxh, hb, w2tf16, w1tf16) (round 5, the fused form): the layer above has not even run its node_pre_bwd --
    the consumer runs that chain on its own tiles first; `gn_parts` is then None."""

    def __init__(self, gx, gvec, gn_parts, gvec_parts, x, mean, rstd, gx1, gvec1, h_real, chain=None):
        self.gx, self.gvec = gx, gvec
        self.tensors = (gn_parts, gvec_parts, x, mean, rstd, gx1, gvec1)        # (kept alive until consumed)
        self.chain = chain
        nparts = gvec_parts.size(0)
        extra = [None] * 4 if chain is None else [P(t) for t in chain]
        self.struct = _lib.PendingGrads(*[P(t) for t in self.tensors], nparts, h_real, *extra)


def fused_boundary_supported(graph, H, w, w_next=None):
    """The layer boundary as one node launch each way (csrc/node_chain16.hip): width 128, HVNet rows, and a row layout whose
    update kernels run on 16-row tiles (small grids: nodeops.update_tile_rows).  `w_next`: the next layer's weights (forward
    fusion: same width and relation count)."""
    if H != 128 or getattr(w, "w1f16", None) is None or graph.num_src or getattr(graph, "res_row", None) is not None:
        return False
    if w_next is not None and (getattr(w_next, "w1f16", None) is None or w_next.b1cat.numel() != graph.T * H):
        return False
    return update_tile_rows(graph, H) == 16


def node_update_pre_fwd(x1, vec1, w, graph, w_next):
    """node_update_fwd of this layer + node_pre_fwd of the NEXT layer (weights `w_next`) on the rows it produces, ONE launch:
    returns (x_out, vec_out, vp, h2b, q23, nrm, (hb, xh, mean, rstd))."""
    N, H = x1.shape
    T = graph.T
    dev, dt = x1.device, x1.dtype
    vp = torch.empty(N, 3, 2 * H, dtype=dt, device=dev)
    h2b = torch.empty(N, H, dtype=dt, device=dev)
    q23 = torch.empty(N, 2 * H, dtype=dt, device=dev)
    nrm = torch.empty(N, H, dtype=dt, device=dev)
    xo = torch.empty(N, H, dtype=dt, device=dev)
    vo = torch.empty(N, 3, H, dtype=dt, device=dev)
    hb = torch.empty(T, N, H, dtype=dt, device=dev)
    xh = torch.empty(T, N, 3 * H, dtype=dt, device=dev)
    mean = torch.empty(N, dtype=dt, device=dev)
    rstd = torch.empty(N, dtype=dt, device=dev)
    _lib.check(_launch("node_update_pre_fwd", lambda: _lib.load().hermnet_node_update_pre_fwd(
        P(x1), P(vec1), P(w.wvf16), P(w.wx0f16), P(w.bx0_s), P(w.wx2f16), P(w.bx2_s), P(graph.row_active), P(graph.type_rowptr),
        _rowptr_host(graph), P(vp), P(h2b), P(q23), P(nrm), P(xo), P(vo), N, T, H,
        P(w_next.w1f16), P(w_next.b1cat), P(w_next.w2f16), P(w_next.b2), P(hb), P(xh), P(mean), P(rstd), T, w_next.h_real,
        1e-5, _stream())), "hermnet_node_update_pre_fwd")
    return xo, vo, vp, h2b, q23, nrm, (hb, xh, mean, rstd)


def node_pre_fwd16(x, w, T):
    """node_pre_fwd on 16-row tiles (the second half of node_update_pre_fwd as a launch of its own: A/B, bit-for-bit check)."""
    Ns, H = x.shape
    dev, dt = x.device, x.dtype
    hb = torch.empty(T, Ns, H, dtype=dt, device=dev)
    xh = torch.empty(T, Ns, 3 * H, dtype=dt, device=dev)
    mean = torch.empty(Ns, dtype=dt, device=dev)
    rstd = torch.empty(Ns, dtype=dt, device=dev)
    _lib.check(_launch("node_pre_fwd16", lambda: _lib.load().hermnet_node_pre_fwd16(
        P(x), P(w.w1f16), P(w.b1cat), P(w.w2f16), P(w.b2), P(hb), P(xh), P(mean), P(rstd), Ns, T, H, w.h_real, 1e-5,
        _stream())), "hermnet_node_pre_fwd16")
    return hb, xh, mean, rstd


def node_pre_bwd16(gxh, hb, w):
    """Per-relation partial sums gn [T,Ns,H] of node_pre_bwd on 16-row tiles (the first half of the fused update backward as
    a launch of its own)."""
    T, Ns, H = hb.shape
    parts = torch.empty(T, Ns, H, dtype=hb.dtype, device=hb.device)
    _lib.check(_launch("node_pre_bwd16", lambda: _lib.load().hermnet_node_pre_bwd16(
        P(gxh), P(hb), P(w.w2tf16), P(w.w1tf16), P(parts), Ns, T, H, _stream())), "hermnet_node_pre_bwd16")
    return parts


def node_update_bwd(gxo, gvo, vp, h2b, q23, nrm, w, graph, pending=None):
    """Gradient of node_update_fwd w.r.t. (x1, vec1).  `pending` (PendingGrads whose gx / gvec ARE gxo / gvo): the kernel
    forms the incoming gradients from the partial sums of the layer above first."""
    N, H = gxo.shape
    t16 = update_tile_rows(graph, H) == 16 and getattr(w, "wvf16", None) is not None
    wx2tf, wx0tf, wvtf = (w.wx2tf16, w.wx0tf16, w.wvtf16) if t16 else (w.wx2tf, w.wx0tf, w.wvtf)
    gx1 = torch.empty_like(gxo)
    gvec1 = torch.empty_like(gvo)
    _lib.check(_launch("node_update_bwd", lambda: _lib.load().hermnet_node_update_bwd(
        P(gxo), P(gvo), P(vp), P(h2b), P(q23), P(nrm), P(wx2tf), P(wx0tf), P(wvtf), P(graph.row_active), P(graph.type_rowptr),
        _rowptr_host(graph), P(gx1), P(gvec1), N, graph.T, H, 16 if t16 else 0,
        None if pending is None else ctypes.byref(pending.struct), _stream())), "hermnet_node_update_bwd")
    return gx1, gvec1
