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
    """Same contract as hermnet_message_scatter_fwd.  xh [T,N,3H]; wt [T,R,3H]."""
    T, N, H3 = xh.shape
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
    x1 = (x + dx) * (1 / math.sqrt(2.0)) * rk[:, None]
    v0 = vec if vec is not None else torch.zeros_like(dv)
    vec1 = (v0 + dv) * rk[:, None, None]
    return x1, vec1
