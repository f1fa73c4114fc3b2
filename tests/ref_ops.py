"""Plain-PyTorch fp32/fp64 restatement of the two fused operators, in relation (row) order,
with DENSE radial basis (rmnet.py:168-172) -- used to localise kernel bugs op by op."""
import math

import torch


def geometry_ref(pos, graph, cell):
    j, i = graph.src_id.long(), graph.tgt_id.long()
    D = pos[j] - pos[i]
    if graph.shift is not None and cell is not None:
        c = cell.reshape(-1, 3, 3).to(pos.dtype)
        D = D + torch.einsum('ni,nij->nj', graph.shift.to(pos.dtype), c[graph.batch32.long()[j]])
    d = D.norm(dim=-1)
    d = torch.where(d.abs() <= 1e-6, torch.full_like(d, 1e-6), d)
    return torch.cat([D / d[:, None], d[:, None]], dim=1)


def envelope_ref(u, kind, p):
    if kind == 0:
        a = -(p + 1) * (p + 2) / 2
        b = p * (p + 2)
        c = -p * (p + 1) / 2
        val = 1 + a * u ** p + b * u ** (p + 1) + c * u ** (p + 2)
    else:
        val = torch.exp(-(u ** 2) / ((1 - u) * (1 + u)))
    return torch.where(u < 1, val, torch.zeros_like(u))


def message_scatter_ref(xh, vec, x, edge, wt, brbf, graph, rbf):
    """Same contract as hermnet_message_scatter_fwd.  xh [T,Ns,3H]; wt [T,R,3H]; outputs in TARGET rows
    (graph.N; = source rows unless graph.num_src / graph.res_row are set: HTNet's virtual target rows)."""
    T, _, H3 = xh.shape
    N = graph.N
    H = H3 // 3
    dt = x.dtype
    rowptr = graph.csr_rowptr.long()
    tgt_row = torch.repeat_interleave(torch.arange(N, device=x.device), rowptr[1:] - rowptr[:-1])
    src = graph.csr_src.long()
    trp = graph.type_rowptr.long()
    rel_row = torch.bucketize(torch.arange(N, device=x.device), trp[1:], right=True)  # T for unknown rows
    rel_e = rel_row[tgt_row]
    known = rel_e < T
    rhat, d = edge[:, :3], edge[:, 3]
    u = d * rbf.inv_rc
    env = envelope_ref(u, rbf.env_kind, rbf.env_p)
    off = rbf.offset.to(dt)
    coeff = -0.5 / float(off[1] - off[0]) ** 2
    emb = env[:, None] * torch.exp(coeff * (u[:, None] - off[None, :]) ** 2)        # [E,R]
    re = rel_e.clamp(max=T - 1)
    rb = torch.einsum('er,erc->ec', emb, wt[re]) + brbf[re]                          # [E,3H]
    m = xh[re, src] * rb
    s, a, b = m[:, :H], m[:, H:2 * H], m[:, 2 * H:]
    mv = b[:, None, :] * rhat[:, :, None]
    if vec is not None:
        mv = mv + vec[src] * (a * (1 / math.sqrt(3.0)))[:, None, :]
    mv = mv * (1 / math.sqrt(H))
    kf = known.to(dt)
    dx = torch.zeros(N, H, dtype=dt, device=x.device).index_add_(0, tgt_row, s * kf[:, None])
    dv = torch.zeros(N, 3, H, dtype=dt, device=x.device).index_add_(0, tgt_row, mv * kf[:, None, None])
    rk = (rel_row < T).to(dt)
    res = getattr(graph, "res_row", None)
    xr = x if res is None else x[res.long()]
    x1 = (xr + dx) * (1 / math.sqrt(2.0)) * rk[:, None]
    v0 = torch.zeros_like(dv) if vec is None else (vec if res is None else vec[res.long()])
    vec1 = (v0 + dv) * rk[:, None, None]
    return x1, vec1


# ---------------------------------------------------------------------------------------------
# Plain-PyTorch versions of the node-level fused kernels (same contracts as hermnet_amd.nodeops)
# and of the message kernels' forward/backward entry points used by hermnet_amd.layer.
# ---------------------------------------------------------------------------------------------
def layernorm_fwd(x, eps=1e-5, h_real=0):
    Hr = h_real or x.size(-1)
    n, mean, rstd = torch.native_layer_norm(x[:, :Hr].contiguous(), [Hr], None, None, eps)
    if Hr < x.size(-1):
        n = torch.nn.functional.pad(n, (0, x.size(-1) - Hr))
    return n, mean.reshape(-1), rstd.reshape(-1)


def layernorm_bwd(g, x, mean, rstd, add=None, h_real=0):
    Hr = h_real or x.size(-1)
    gx = torch.ops.aten.native_layer_norm_backward(g[:, :Hr].contiguous(), x[:, :Hr].contiguous(), [Hr], mean.reshape(-1, 1),
                                                   rstd.reshape(-1, 1), None, None, [True, False, False])[0]
    if Hr < x.size(-1):
        gx = torch.nn.functional.pad(gx, (0, x.size(-1) - Hr))
    return gx if add is None else gx + add


def energy_head_fwd(h, w, b, mask=None):
    e = (torch.nn.functional.silu(h) / 0.6) @ w + (0 if b is None else b[0])
    return e if mask is None else e * mask


def energy_head_bwd(ge, h, w, mask=None):
    g = ge if mask is None else ge * mask
    return g[:, None] * w[None, :] * _dssilu(h)


def _with_bias(h, bias, rows_per_bias):
    """h [rows, cols] + bias [groups, cols] (group = row // rows_per_bias; <= 0: one row): the kernels'
    "bias convention" (include/hermnet_hip.h)."""
    if bias is None:
        return h
    b = bias.reshape(-1, h.size(-1))
    if rows_per_bias <= 0:
        return h + b[0]
    grp = torch.arange(h.size(0)) // rows_per_bias
    return h + b[grp]


def ssilu_fwd(h, bias=None, rows_per_bias=0):
    return torch.nn.functional.silu(_with_bias(h, bias, rows_per_bias)) / 0.6


def _dssilu(h):
    s = torch.sigmoid(h)
    return s * (1 + h * (1 - s)) / 0.6


def ssilu_bwd(g, h, N, T, C, gs_n, gs_t, bias=None, rows_per_bias=0):
    gg = torch.as_strided(g, (N, T, C), (gs_n, gs_t, 1))
    hb = _with_bias(h.reshape(-1)[:N * T * C].view(N, T * C), bias, rows_per_bias)
    return (gg * _dssilu(hb.view(N, T, C))).reshape(N, T * C)


def update_mid(vp, x1, rows, H):
    v1, v2 = vp[..., :H], vp[..., H:]
    vdot = torch.zeros(x1.size(0), H, dtype=x1.dtype)
    xin = torch.zeros(x1.size(0), 2 * H, dtype=x1.dtype)
    vdot[:rows] = (v1[:rows] * v2[:rows]).sum(1) / math.sqrt(H)
    xin[:rows, :H] = x1[:rows]
    xin[:rows, H:] = torch.sqrt((v2[:rows] ** 2).sum(1) + 1e-8)
    return vdot, xin


def _on(mask, N, nk):
    on = torch.arange(N) < nk
    if mask is not None:
        on = on & (mask != 0)
    return on


def _q_with_bias(q, qbias, rows_per_bias, nk):
    if qbias is None:
        return q
    q = q.clone()
    q[:nk] = _with_bias(q[:nk], qbias, rows_per_bias)
    return q


def update_out(q, vdot, vp, x1, vec1, mask, N, nk, H, qbias=None, rows_per_bias=0):
    on = _on(mask, N, nk)
    q = _q_with_bias(q, qbias, rows_per_bias, nk)
    q1, q2, q3 = q[:, :H], q[:, H:2 * H], q[:, 2 * H:]
    xo = x1 + (q1 + q2 * vdot) / math.sqrt(2.0)
    vo = vec1 + q3[:, None, :] * vp[..., :H]
    z = torch.zeros(())
    return torch.where(on[:, None], xo, z), torch.where(on[:, None, None], vo, z)


def update_out_bwd(gxo, gvo, q, vdot, vp, mask, N, nk, H, qbias=None, rows_per_bias=0):
    on = _on(mask, N, nk)
    q = _q_with_bias(q, qbias, rows_per_bias, nk)
    z = torch.zeros(())
    gx = torch.where(on[:, None], gxo, z)
    gv = torch.where(on[:, None, None], gvo, z)
    q2, q3 = q[:, H:2 * H], q[:, 2 * H:]
    s = 1 / math.sqrt(2.0)
    gq = torch.cat([gx * s, gx * vdot * s, (gv * vp[..., :H]).sum(1)], 1)
    gq = torch.where(on[:, None], gq, z)
    gvdot = torch.where(on[:, None], gx * q2 * s, z)
    gvp = torch.zeros(N, 3, 2 * H, dtype=gxo.dtype)
    gvp[..., :H] = torch.where(on[:, None, None], gv * q3[:, None, :], z)
    gvp[..., H:] = float("nan")          # the kernel leaves this half for update_mid_bwd
    gvp[nk:] = float("nan")
    return gq, gvdot, gvp, gx.clone(), gv.clone()


def update_mid_bwd(gvdot, gxin, vp, xin, gvp, gx1, rows, H):
    s = 1 / math.sqrt(H)
    v1, v2 = vp[:rows, :, :H], vp[:rows, :, H:]
    gd = gvdot[:rows, None, :]
    gnn = (gxin[:rows, H:] / xin[:rows, H:])[:, None, :]
    gx1[:rows] += gxin[:rows, :H]
    p = gvp[:rows, :, :H].clone()
    gvp[:rows, :, :H] = p + gd * s * v2
    gvp[:rows, :, H:] = gd * s * v1 + gnn * v2


# ---- node chain kernels (hermnet_amd.nodeops.node_*; csrc/node_chain.hip): same contracts, plain weights of `w` --------
def _tile_rows(Ns, H, windows, mode, device):
    """Rows of the tiles a windowed launch of the pre kernels runs (csrc/node_chain.hip: tile_selected)."""
    from hermnet_amd import nodeops
    TR = nodeops.chain_tile_rows(H)
    row0 = torch.arange(Ns, device=device) // TR * TR
    inside = torch.zeros(Ns, dtype=torch.bool, device=device)
    for lo, hi in windows.tolist():
        if hi > lo:
            inside |= (row0 < hi) & (row0 + TR > lo)
    return inside if mode == 1 else ~inside


def node_pre_fwd(x, w, T, src_ranges=None, windows=None, mode=0, out=None):
    Ns, H = x.shape
    dt = x.dtype
    n, mean, rstd = layernorm_fwd(x, 1e-5, h_real=w.h_real)
    hb = (n @ w.w1cat.to(dt).t() + w.b1cat.to(dt)).view(Ns, T, H).transpose(0, 1).contiguous()      # [T, Ns, H]
    a = torch.nn.functional.silu(hb) / 0.6
    xh = torch.bmm(a, w.w2t.to(dt)) + w.b2.to(dt)                                                    # [T, Ns, 3H]
    if windows is None or mode == 0:
        return hb, xh, mean, rstd
    # a windowed launch writes the rows of its tiles only (the rest: poison until the other launch fills it)
    rows = _tile_rows(Ns, H, windows, mode, x.device)
    if out is None:
        out = tuple(torch.full_like(t_, float("nan")) for t_ in (hb, xh, mean, rstd))
    out[0][:, rows], out[1][:, rows], out[2][rows], out[3][rows] = hb[:, rows], xh[:, rows], mean[rows], rstd[rows]
    return out


def node_pre_bwd(gxh, hb, x, mean, rstd, w, add=None, src_ranges=None, windows=None, mode=0, out=None, parts_only=False):
    T, Ns, H = hb.shape
    dt = x.dtype
    gh = torch.bmm(gxh, w.w2.to(dt)) * _dssilu(hb)                                                   # [T, Ns, H]
    gn = torch.bmm(gh, w.w1cat.to(dt).view(T, H, H))
    if parts_only:            # (hermnet_node_pre_bwd with gx = NULL: the per-relation partial sums only)
        return gn
    gn = gn.sum(0)
    gx = layernorm_bwd(gn, x, mean, rstd, add=add, h_real=w.h_real)
    if windows is None or mode == 0:
        return gx
    rows = _tile_rows(Ns, H, windows, mode, x.device)
    if out is None:
        out = (torch.full_like(gx, float("nan")), None)
    out[0][rows] = gx[rows]
    return out


def halo_rows(mode, x, vec, idx, buf=None):
    """csrc/node_kernels.hip: hermnet_halo_rows -- 0 pack, 1 pack-and-clear, 2 unpack (in place)."""
    H = x.size(1)
    if mode in (0, 1):
        buf = torch.cat([x.index_select(0, idx), vec.index_select(0, idx).reshape(-1, 3 * H)], dim=1)
        if mode == 1:
            x.index_fill_(0, idx, 0)
            vec.index_fill_(0, idx, 0)
        return buf
    assert mode == 2
    x.index_copy_(0, idx, buf[:, :H])
    vec.index_copy_(0, idx, buf[:, H:].reshape(-1, 3, H))
    return buf


def halo_proj_rows(mode, a, b, idx, buf=None):
    """csrc/node_kernels.hip: hermnet_halo_proj_rows -- packed row = [a[0][r] | ... | a[S-1][r] | sum_s b[s][r]]."""
    S, N, W = a.shape
    sliced = b.dim() == 4
    bv = b.reshape((b.size(0) if sliced else 1), N, W)
    if mode in (0, 1):
        buf = torch.cat([a.index_select(1, idx).permute(1, 0, 2).reshape(idx.numel(), S * W),
                         bv.index_select(1, idx).sum(0)], dim=1)
        if mode == 1:
            a.index_fill_(1, idx, 0)
            bv.index_fill_(1, idx, 0)
        return buf
    assert mode == 2 and not sliced
    a.index_copy_(1, idx, buf[:, :S * W].reshape(-1, S, W).permute(1, 0, 2))
    bv[0].index_copy_(0, idx, buf[:, S * W:])
    return buf


def halo_proj_accumulate(a, b, plan, buf):
    S, N, W = a.shape
    bv = b.reshape(-1, N, W)
    a.index_add_(1, plan.send_idx, buf[:, :S * W].reshape(-1, S, W).permute(1, 0, 2))
    bv[0].index_add_(0, plan.send_idx, buf[:, S * W:])


def halo_accumulate(x, vec, plan, buf):
    """hermnet_halo_accumulate: returned gradients summed per owned row in send-list order."""
    H = x.size(1)
    if buf.size(0) == 0:
        return
    rows, ptr, pos = plan.accumulate_lists()
    seg = torch.segment_reduce(buf.index_select(0, pos), "sum", lengths=ptr[1:] - ptr[:-1])
    x.index_add_(0, rows, seg[:, :H])
    vec.index_add_(0, rows, seg[:, H:].reshape(-1, 3, H))


def _update_parts(x1, vec1, w, graph):
    """Per relation block: (rows, v1, v2, vdot, norm, h2b, p, q, r) with autograd enabled inputs."""
    H = x1.size(1)
    dt = x1.dtype
    rp = graph.type_rowptr_host
    out = []
    for t in range(graph.T):
        lo, hi = rp[t], rp[t + 1]
        if hi <= lo:
            continue
        vp = vec1[lo:hi] @ w.wv[t].to(dt).t()                                   # [n, 3, 2H]
        v1, v2 = vp[..., :H], vp[..., H:]
        vdot = (v1 * v2).sum(1) / math.sqrt(H)
        norm = torch.sqrt((v2 ** 2).sum(1) + 1e-8)
        h2b = torch.cat([x1[lo:hi], norm], 1) @ w.wx0[t].to(dt).t() + w.bx0[t].to(dt)
        pqr = (torch.nn.functional.silu(h2b) / 0.6) @ w.wx2[t].to(dt).t() + w.bx2[t].to(dt)
        out.append((lo, hi, vp, v1, vdot, h2b, pqr))
    return out


def _node_update(x1, vec1, w, graph):
    N, H = x1.shape
    dt = x1.dtype
    act = torch.ones(N, dtype=dt) if graph.row_active is None else graph.row_active.to(dt)
    xo = torch.zeros(N, H, dtype=dt)
    vo = torch.zeros(N, 3, H, dtype=dt)
    vp_all = torch.zeros(N, 3, 2 * H, dtype=dt)
    h2b_all = torch.zeros(N, H, dtype=dt)
    q23 = torch.zeros(N, 2 * H, dtype=dt)
    nrm = torch.zeros(N, H, dtype=dt)
    xs, vs = [], []
    for lo, hi, vp, v1, vdot, h2b, pqr in _update_parts(x1, vec1, w, graph):
        p, q, r = pqr[:, :H], pqr[:, H:2 * H], pqr[:, 2 * H:]
        m = (act[lo:hi] != 0).to(dt)
        xs.append((lo, hi, (x1[lo:hi] + (p + q * vdot) / math.sqrt(2.0)) * m[:, None]))
        vs.append((lo, hi, (vec1[lo:hi] + r[:, None, :] * v1) * m[:, None, None]))
        vp_all[lo:hi], h2b_all[lo:hi], q23[lo:hi] = vp.detach(), h2b.detach(), pqr[:, H:].detach()
        nrm[lo:hi] = torch.sqrt((vp[..., H:].detach() ** 2).sum(1) + 1e-8)
    if xs:
        xo = torch.cat([torch.zeros(0, H, dtype=dt)] + _fill_rows(xs, N, (H,), dt))
        vo = torch.cat([torch.zeros(0, 3, H, dtype=dt)] + _fill_rows(vs, N, (3, H), dt))
    return xo, vo, vp_all, h2b_all, q23, nrm


def _fill_rows(parts, N, shape, dt):
    """Differentiable assembly of row blocks [(lo, hi, values)] into an N-row array (gaps = zero rows)."""
    out, at = [], 0
    for lo, hi, v in parts:
        if lo > at:
            out.append(torch.zeros((lo - at,) + shape, dtype=dt))
        out.append(v)
        at = hi
    if at < N:
        out.append(torch.zeros((N - at,) + shape, dtype=dt))
    return out


def node_update_fwd(x1, vec1, w, graph):
    with torch.no_grad():
        return _node_update(x1, vec1, w, graph)


def node_update_pre_fwd(x1, vec1, w, graph, w_next):
    """hermnet_node_update_pre_fwd: this layer's update and the NEXT layer's node projection of the rows it produces."""
    out = node_update_fwd(x1, vec1, w, graph)
    return tuple(out) + (node_pre_fwd(out[0], w_next, graph.T),)


def node_pre_fwd16(x, w, T):
    return node_pre_fwd(x, w, T)


def node_pre_bwd16(gxh, hb, w):
    T, Ns, H = hb.shape
    dt = hb.dtype
    gh = torch.bmm(gxh, w.w2.to(dt)) * _dssilu(hb)
    return torch.bmm(gh, w.w1cat.to(dt).view(T, H, H))


def node_update_bwd(gxo, gvo, vp, h2b, q23, nrm, w, graph, pending=None):
    """The kernel's backward formulas from the saved (vp, h2b, q23) (checked against autograd of the forward in
    tests/test_host_logic.py).  `pending` (hn_pending_grads): gxo / gvo are formed here first, from the partial sums of
    the layer above, and written into the caller's buffers."""
    N, H = gxo.shape
    dt = gxo.dtype
    if pending is not None:
        gn, gv, x, mean, rstd, gx1_up, gvec1_up = pending.tensors
        if pending.chain is not None:       # the fused form: the layer above's projection backward runs here, on gxh
            gn = node_pre_bwd16(pending.chain[0], pending.chain[1], pending.w_above)
        ident = (torch.arange(N) < graph.type_rowptr_host[-1]).to(dt)
        gxo.copy_(layernorm_bwd(gn.sum(0), x, mean, rstd, add=gx1_up * ident[:, None] / math.sqrt(2.0),
                                h_real=pending.struct.hidden_real))
        gvo.copy_(gv.sum(0) + gvec1_up * ident[:, None, None])
    act = torch.ones(N, dtype=dt) if graph.row_active is None else graph.row_active.to(dt)
    gx1 = torch.zeros(N, H, dtype=dt)
    gvec1 = torch.zeros(N, 3, H, dtype=dt)
    rp = graph.type_rowptr_host
    r2, rh = 1 / math.sqrt(2.0), 1 / math.sqrt(H)
    for t in range(graph.T):
        lo, hi = rp[t], rp[t + 1]
        if hi <= lo:
            continue
        m = (act[lo:hi] != 0).to(dt)
        gx, gv = gxo[lo:hi] * m[:, None], gvo[lo:hi] * m[:, None, None]
        v1, v2 = vp[lo:hi, :, :H], vp[lo:hi, :, H:]
        vdot = (v1 * v2).sum(1) * rh
        norm = nrm[lo:hi]
        q2, q3 = q23[lo:hi, :H], q23[lo:hi, H:]
        gq = torch.cat([gx * r2, gx * vdot * r2, (gv * v1).sum(1)], 1)
        gh2 = (gq @ w.wx2[t].to(dt)) * _dssilu(h2b[lo:hi])
        gxin = gh2 @ w.wx0[t].to(dt)
        gx1[lo:hi] = gx + gxin[:, :H]
        gnn = gxin[:, H:] / norm
        s_ = gx * q2 * (r2 * rh)
        gv1 = gv * q3[:, None, :] + s_[:, None, :] * v2
        gv2 = s_[:, None, :] * v1 + gnn[:, None, :] * v2
        gvec1[lo:hi] = gv + torch.cat([gv1, gv2], -1) @ w.wv[t].to(dt)
    return gx1, gvec1


def node_update_bwd_from_inputs(gxo, gvo, x1, vec1, w, graph):
    with torch.enable_grad():
        x1_ = x1.detach().requires_grad_(True)
        v_ = vec1.detach().requires_grad_(True)
        xo, vo = _node_update(x1_, v_, w, graph)[:2]
        return torch.autograd.grad([xo, vo], [x1_, v_], [gxo, gvo])


def _rows_of_ranges(ranges, n):
    sel = torch.zeros(n, dtype=torch.bool)
    for lo, hi in ranges:
        if hi > lo:
            sel[lo:hi] = True
    return sel


def msg_fwd(graph, rbf, H, xh, vec, x, w, edge, xh_bias=True, ranges=None, zero_unknown=True, out=None, range_rows=0):
    """`ranges` [T,2]: like the kernel, only these target rows are written (the rest keeps what `out` holds: poison
    when this call allocates it)."""
    edge = edge[0] if edge.dim() == 3 else edge
    x1, vec1 = message_scatter_ref(xh + w.b2 if xh_bias else xh, vec, x, edge, w.wt, w.brbf, graph, rbf)  # xh_bias = w.b2 [T,1,3H]
    if ranges is None:
        return x1, vec1
    sel = _rows_of_ranges(ranges.tolist(), x1.size(0))
    if zero_unknown:
        sel[int(graph.type_rowptr[-1]):] = True
    if out is None:
        out = (torch.full_like(x1, float("nan")), torch.full_like(vec1, float("nan")))
    out[0][sel], out[1][sel] = x1[sel], vec1[sel]
    return out


def msg_bwd(graph, rbf, H, xh, vec, w, edge, gx1, gvec1, gedge, xh_bias=True, ranges=None, out=None, finish=True):
    """Backward contract of hermnet_message_scatter_bwd via autograd of the dense restatement;
    the edge gradient is Cartesian (w.r.t. D = rhat * d).  `ranges` = (tensor, [(lo, hi)]): only these SOURCE rows (and
    the edges leaving them) are written, the call returns its buffers (gxh, gvec, gx, None) for the complementary call.
    `finish=False` (gx = NULL in the C call): (gxh, per-relation partial sums of gvec WITHOUT the residual's identity term)."""
    if not finish:
        T = xh.size(0)
        scratch = gedge if ranges is None else torch.zeros_like(gedge)
        gxh, gv, _ = msg_bwd(graph, rbf, H, xh, vec, w, edge, gx1, gvec1, scratch, xh_bias=xh_bias)
        part = None
        if gv is not None:
            ident = (torch.arange(graph.N) < int(graph.type_rowptr[T])).to(gv.dtype)
            part = torch.zeros((T,) + tuple(gv.shape), dtype=gv.dtype)
            part[0] = gv - gvec1 * ident[:, None, None]      # (the split over the relations is the kernel's own business)
        if ranges is None:
            return gxh, part
        # ranged launch without the finishing launch (the "proj" halo exchange): only these SOURCE rows are written
        sel = _rows_of_ranges(ranges[1], xh.size(1))
        if out is None:
            out = (torch.full_like(gxh, float("nan")), None if part is None else torch.full_like(part, float("nan")))
        out[0][:, sel] = gxh[:, sel]
        if part is not None:
            out[1][:, sel] = part[:, sel]
        esel = sel[graph.csr_src.long()]
        gedge[:, esel] = scratch[:, esel]
        return out
    with torch.enable_grad():
        xh_ = (xh + w.b2 if xh_bias else xh).detach().requires_grad_(True)
        x_ = torch.zeros(xh.size(1), H, dtype=xh.dtype).requires_grad_(True)
        v_ = vec.detach().requires_grad_(True) if vec is not None else None
        if edge.dim() == 3:          # per-layer handle of EdgeFanout: [H/64, E, 4] view of the same edge array
            edge = edge[0]
        D = (edge[:, :3] * edge[:, 3:4]).detach().requires_grad_(True)
        dn = D.norm(dim=-1)
        e_ = torch.cat([D / dn[:, None], dn[:, None]], 1)
        x1, vec1 = message_scatter_ref(xh_, v_, x_, e_, w.wt, w.brbf, graph, rbf)
        ins = [xh_, x_, D] + ([v_] if vec is not None else [])
        gr = torch.autograd.grad([x1, vec1], ins, [gx1, gvec1])
    gxh, gv, gx = gr[0], (gr[3] if vec is not None else None), gr[1]
    res = getattr(graph, "res_row", None)
    if res is not None:
        # kernel contract with virtual target rows: NO identity (residual) term in gx / gvec -- the host adds it
        T = xh.size(0)
        rk = (torch.arange(graph.N) < int(graph.type_rowptr[T])).to(xh.dtype)
        gx = gx - torch.zeros_like(gx).index_add_(0, res.long(), gx1 * rk[:, None] * (1 / math.sqrt(2.0)))
        if gv is not None:
            gv = gv - torch.zeros_like(gv).index_add_(0, res.long(), gvec1 * rk[:, None, None])
    if ranges is None:
        gedge.zero_()                    # (the caller may hand over uninitialised memory: the kernels write every slot)
        gedge[0, :, :3] = gr[2]          # all column blocks' contributions in slice 0, the others zero
        return gxh, gv, gx
    sel = _rows_of_ranges(ranges[1], xh.size(1))
    if out is None:
        nan = lambda t_: None if t_ is None else torch.full_like(t_, float("nan"))
        out = (nan(gxh), nan(gv), nan(gx), None)
    out[0][:, sel] = gxh[:, sel]
    if gv is not None:
        out[1][sel] = gv[sel]
    out[2][sel] = gx[sel]
    esel = sel[graph.csr_src.long()]                # the edges leaving the selected source rows
    gedge[:, esel] = 0
    gedge[0, esel, :3] = gr[2][esel]
    return out


class RefEdgeGeometry(torch.autograd.Function):
    """Same contract as hermnet_amd.ops.EdgeGeometry: the incoming gradient is Cartesian (dE/dD)."""

    @staticmethod
    def forward(ctx, pos, cell, graph):
        ctx.graph = graph
        ctx.cell_shape = None if cell is None else tuple(cell.shape)
        return geometry_ref(pos.detach(), graph, None if cell is None else cell.detach())

    @staticmethod
    def backward(ctx, gedge):
        g = ctx.graph
        gp = torch.zeros(g.num_atoms, 3, dtype=gedge.dtype)
        gp.index_add_(0, g.src_id.long(), gedge[:, :3])
        gp.index_add_(0, g.tgt_id.long(), -gedge[:, :3])
        gcell = None
        if ctx.needs_input_grad[1] and g.shift is not None and ctx.cell_shape is not None:
            # D = ... + shift @ cell[batch[src]]  =>  dE/dcell[b] = sum_e shift_e (x) gD_e   (ops.EdgeGeometry)
            outer = g.shift.to(gedge.dtype)[:, :, None] * gedge[:, None, :3]
            nb = 1
            for v in ctx.cell_shape[:-2]:
                nb *= v
            b = g.batch32.long()[g.src_id.long()]
            gcell = torch.zeros(nb, 3, 3, dtype=gedge.dtype).index_add_(0, b, outer).reshape(ctx.cell_shape)
        return gp, gcell, None
