"""train() mode's edge part (`example/dist_train.py:86-99` differentiates the forces w.r.t.
the parameters, so every op here is differentiable TWICE): the autograd functions around the training kernels of
`csrc/train_kernels.hip` and the torch-op forms they fall back to on host tensors.

  GatherRows / SumRows        row gather and its adjoint (a segmented sum), closed under differentiation
  EdgeMessage(Grad)           the per-edge message algebra of rmnet.py:58-66 with per-edge operands
  MessageAlgebra(Grad)        the same with node-level inputs and outputs (gathers and row sums inside)
  BucketedBasis / TallLinear  rbf_proj (rmnet.py:55) on the distance-bucketed basis / on a dense basis
  message_scatter_generic     rbf_proj + gather + message + aggregation + residual of ALL relations of a layer
"""
import math
import os

import ctypes

import torch
import torch.nn.functional as F

from . import _lib


class _RowKey(object):
    """One edge -> row assignment `idx` [K] (values < n_rows) in both forms the pair below needs: the index vector for
    the gather, and (perm, lengths) -- the edges sorted by row and the run lengths of all n_rows rows -- for a sum
    without atomics.  perm = None: the edges already come sorted by row."""

    def __init__(self, idx, perm, lengths, n_rows):
        self.idx, self.perm, self.lengths, self.n_rows = idx, perm, lengths, int(n_rows)
        self.rowptr = torch.zeros(self.n_rows + 1, dtype=torch.long, device=lengths.device)
        self.rowptr[1:] = torch.cumsum(lengths, 0)


class GatherRows(torch.autograd.Function):
    """y = x[key.idx].  Gather and row-sum are each other's adjoint, so the pair is closed under differentiation to any
    order (create_graph=True): no zero-filled index_add_ with float atomics anywhere in the training step."""

    @staticmethod
    def forward(ctx, x, key):
        ctx.key = key
        return x.index_select(0, key.idx)

    @staticmethod
    def backward(ctx, g):
        return SumRows.apply(g, ctx.key), None


class SumRows(torch.autograd.Function):
    """y[r] = sum of x[k] over the edges k with key.idx[k] == r, in a fixed order (segmented sum over the sorted
    edges: deterministic, unlike index_add_)."""

    @staticmethod
    def forward(ctx, x, key):
        ctx.key = key
        if x.size(0) == 0:
            return x.new_zeros((key.n_rows,) + tuple(x.shape[1:]))
        width = x[0].numel()
        if x.is_cuda and x.dtype == torch.float32 and width % 4 == 0:
            # one pass: the rows are gathered inside the sum (csrc/train_kernels.hip: hermnet_segment_sum)
            return _segsum(x, key)
        xs = x if key.perm is None else x.index_select(0, key.perm)
        return torch.segment_reduce(xs.contiguous(), "sum", lengths=key.lengths, unsafe=True)

    @staticmethod
    def backward(ctx, g):
        return GatherRows.apply(g, ctx.key), None


class ToSlots(torch.autograd.Function):
    """y[r] = u[src[r]] for rows r that hold an edge (src[r] < len(u)), 0 for padding rows; `slot` is the inverse map on the
    edges (slot[e] = the row of edge e, a permutation onto the occupied rows).  The adjoint is therefore a GATHER,
    g_u[e] = g[slot[e]] -- index_select's own backward is an index_add_ with float atomics, and every padding row hits the one
    dummy entry (0.28 ms per pass at configs[4]'s batch, and not reproducible bit for bit)."""

    @staticmethod
    def forward(ctx, u, src, slot):
        ctx.maps = (src, slot)
        return torch.cat([u, u.new_zeros((1,) + tuple(u.shape[1:]))]).index_select(0, src)

    @staticmethod
    def backward(ctx, g):
        return FromSlots.apply(g, *ctx.maps), None, None


class FromSlots(torch.autograd.Function):
    """y[e] = g[slot[e]]: the adjoint of `ToSlots` (and ToSlots is its adjoint)."""

    @staticmethod
    def forward(ctx, g, src, slot):
        ctx.maps = (src, slot)
        return g.index_select(0, slot)

    @staticmethod
    def backward(ctx, c):
        return ToSlots.apply(c, *ctx.maps), None, None


class EdgeUnit(torch.autograd.Function):
    """D [E,3] -> (U = D / d [E,3], d = max(|D|, 1e-6) [E]): hermnet.py:144-152 (`hermnet_edge_unit`, csrc/train_kernels.hip), one
    launch per order of differentiation; twice differentiable (a third derivative raises)."""

    @staticmethod
    def forward(ctx, D):
        D = _c(D)
        ctx.save_for_backward(D)
        return _edge_unit(0, D, None, None, None)[:2]

    @staticmethod
    def backward(ctx, gU, gd):
        (D,) = ctx.saved_tensors
        return _EdgeUnitGrad.apply(gU, gd, D)


class _EdgeUnitGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gU, gd, D):
        gU, gd = _c(gU), _c(gd)
        ctx.save_for_backward(gU, gd, D)
        return _edge_unit(1, D, gU, gd, None)[0]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, C):
        gU, gd, D = ctx.saved_tensors
        c_gU, c_gd, c_D = _edge_unit(2, D, gU, gd, _c(C))
        return (c_gU if gU is not None else None), (c_gd if gd is not None else None), c_D


def _edge_unit(order, D, gU, gd, C):
    from .ops import _stream
    E = D.size(0)
    new = lambda *shape: torch.empty(*shape, dtype=D.dtype, device=D.device)
    o0, o1, o2 = new(E, 3), (new(E) if order != 1 else None), (new(E, 3) if order == 2 else None)
    P = _lib.ptr
    _lib.check(_lib.load().hermnet_edge_unit(order, P(D), P(gU), P(gd), P(C), E, P(o0), P(o1), P(o2), _stream()),
               "hermnet_edge_unit")
    return o0, o1, o2


class BasisWindow(torch.autograd.Function):
    """phi [nc,C,32] of the bucketed basis from the sorted distances u [nc*C] (`hermnet_basis_window`, csrc/band_product.hip):
    Gaussian window of the chunk's 32 centres x polynomial envelope, zero on padding rows (src == num_edges) and beyond the
    cutoff.  One launch per order of differentiation; differentiable twice (the step needs no more: a third derivative
    raises instead of returning something else)."""

    @staticmethod
    def forward(ctx, u, src, mu, w, num_edges, C, coeff, p):
        ctx.save_for_backward(u, src, mu, w)
        ctx.consts = (int(num_edges), int(C), float(coeff), int(p))
        return _basis_window(0, u, src, mu, w, ctx.consts, None, None)[0]

    @staticmethod
    def backward(ctx, g):
        u, src, mu, w = ctx.saved_tensors
        return (_BasisWindowGrad.apply(g, u, src, mu, w, ctx.consts),) + (None,) * 7


class _BasisWindowGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, u, src, mu, w, consts):
        g = _c(g)
        ctx.save_for_backward(g, u, src, mu, w)
        ctx.consts = consts
        return _basis_window(1, u, src, mu, w, consts, g, None)[0]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, cu):
        g, u, src, mu, w = ctx.saved_tensors
        d_g, d_u = _basis_window(2, u, src, mu, w, ctx.consts, g, _c(cu))
        return d_g, d_u, None, None, None, None


def _basis_window(order, u, src, mu, w, consts, g, cu):
    from .ops import _stream
    num_edges, C, coeff, p = consts
    nc = mu.size(0)
    new = lambda *shape: torch.empty(*shape, dtype=u.dtype, device=u.device)
    out0 = new(nc * C) if order == 1 else new(nc, C, 32)
    out1 = new(nc * C) if order == 2 else None
    P = _lib.ptr
    _lib.check(_lib.load().hermnet_basis_window(order, P(u), P(src), num_edges, P(mu), P(w), nc, C, coeff, p, P(g), P(cu), P(out0),
                                                P(out1), _stream()), "hermnet_basis_window")
    return out0, out1


class _EmbedRows(torch.autograd.Function):
    """y = W[index] (nn.Embedding, hermnet.py:123).  The gradient is a sum of ~N rows into a handful of table rows: formed as
    onehot(index)^T g through `_gram_over_rows` (a batched product over row chunks + a sum: a fixed order, the whole chip) --
    embedding's own backward sorts the indices with a multi-pass merge sort and scatters (~45 launches, 0.5 ms per step at
    configs[4]'s batch), a segmented sum leaves one lane group per table row (1.8 ms)."""

    @staticmethod
    def forward(ctx, W, index):
        ctx.index, ctx.n = index, W.size(0)
        return W.index_select(0, index)

    @staticmethod
    def backward(ctx, g):
        return _EmbedRowsT.apply(g, ctx.index, ctx.n), None


class _EmbedRowsT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g, index, n):
        ctx.index = index
        onehot = g.new_zeros(index.numel(), n).scatter_(1, index[:, None], 1.0)
        return _gram_over_rows(onehot[None], g.contiguous()[None])[0]

    @staticmethod
    def backward(ctx, c):
        return _EmbedRows.apply(c, ctx.index), None, None


def embedding_rows(weight, index):
    return _EmbedRows.apply(weight, index)


class EdgeDiff(torch.autograd.Function):
    """D[e] = pos[source(e)] - pos[target(e)] over the CSR edges (hermnet.py:135-139) and its adjoint `_EdgeDiffT`, each the
    other's backward (both are linear): differentiable to any order without `index_put_(accumulate=True)` -- the backward of
    `pos[idx]`, which sorts its 4e5 indices every time (0.27 ms per call here) -- and without float atomics.  The adjoint
    sums the edge gradients per TARGET row (the CSR segments) and per (relation, SOURCE row) (the CSC segments; edges into
    atoms of an unknown element are in none of them and carry no gradient), then reads each atom's row."""

    @staticmethod
    def forward(ctx, pos, graph):
        ctx.graph = graph
        s, t = _edge_atoms(graph)
        return pos.index_select(0, s) - pos.index_select(0, t)

    @staticmethod
    def backward(ctx, gD):
        return _EdgeDiffT.apply(gD, ctx.graph), None


class _EdgeDiffT(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gD, graph):
        ctx.graph = graph
        k_t, k_s, T, N = _edge_sum_keys(graph)
        g4 = F.pad(gD, (0, 1)).contiguous()                               # (the sum kernel takes float4 rows)
        rows = _segsum(g4, k_s).view(T, N, 4).sum(0) - _segsum(g4, k_t)     # per row: + as a source, - as a target
        return rows.index_select(0, graph.row_of_node)[:, :3]

    @staticmethod
    def backward(ctx, c):
        return EdgeDiff.apply(c.contiguous(), ctx.graph), None


def _edge_atoms(graph):
    """(source atom, target atom) of every CSR edge as int64, once per graph."""
    ea = getattr(graph, "_edge_atoms64", None)
    if ea is None:
        ea = graph._edge_atoms64 = (graph.src_id.long(), graph.tgt_id.long())
    return ea


def _edge_sum_keys(graph):
    """Row keys of the adjoint of EdgeDiff: every CSR edge by its target row; the CSC edges by (relation, source row)."""
    ks = getattr(graph, "_edge_sum_keys", None)
    if ks is None:
        T, N = graph.T, graph.N
        rp = graph.csr_rowptr.long()
        k_t = _RowKey(None, None, rp[1:] - rp[:-1], N)
        crp = graph.csc_rowptr.long()[:T * N + 1]
        k_s = _RowKey(None, graph.csc_pos.long(), crp[1:] - crp[:-1], T * N)
        ks = graph._edge_sum_keys = (k_t, k_s, T, N)
    return ks


class BucketedBasis(object):
    """The Gaussian basis of the first Ek edges, SORTED by (relation of the target, distance bucket) and cut to the 32
    centres of the edge's bucket: phi [nc, C, 32] (chunks of C rows; every (relation, bucket) group is padded to whole
    chunks with zero rows), group [nc] = relation * nb + bucket of each chunk, slot [Ek] = row of every edge in that
    order.  rbf_proj then is ONE batched [C,32] x [32,3H] product per layer instead of three dense [E_t,R] x [R,3H]
    GEMMs: a quarter of the FLOPs (a Gaussian is < 2e-8 of its peak six widths away, and a 32-centre window holds
    every centre within six widths of any distance of its bucket), still on the matrix pipe, still plain torch ops --
    differentiable to any order."""

    WIDTH, CHUNK = 32, 1024

    def __init__(self, phi, group, slot, nb, num_radial, pad=None, chunks_per_group=None):
        self.phi, self.group, self.slot, self.nb, self.num_radial = phi, group, slot, int(nb), int(num_radial)
        self.pad = pad        # [n_pad] rows of the sorted order that hold no edge
        # (chunks are listed group by group: the weight / bias rows of a chunk are a gather whose adjoint is an ORDERED sum
        # over the chunks of a group -- index_select's own backward adds them with float atomics, the one place where the
        # training step's gradients were not bit-reproducible from run to run)
        self.key_w = self.key_b = None
        self._unit = None
        if chunks_per_group is not None and chunks_per_group.numel() % self.nb == 0:
            cg = chunks_per_group.long()
            self.key_w = _RowKey(group, None, cg, cg.numel())
            self.key_b = _RowKey(group // self.nb, None, cg.view(-1, self.nb).sum(1), cg.numel() // self.nb)

    def unit_vectors(self, edge, Ek):
        """edge[:Ek, :3] (rhat of the known-target edges) as ONE contiguous tensor for all layers of the step: a slice per
        layer costs a copy forward and a zero-fill + copy + add per layer in each backward pass."""
        if self._unit is None or self._unit[0] is not edge:
            self._unit = (edge, edge[:Ek, :3].contiguous())
        return self._unit[1]

    def set_unit_vectors(self, edge, U, Ek):
        """The step's unit vectors when the caller holds them as an [E,3] tensor of their own (trainops.EdgeUnit)."""
        self._unit = (edge, U[:Ek])

    def project(self, w_rbf, b_rbf, scale):
        """rbf_proj of every relation (rmnet.py:55) on the bucketed basis -> (R, R) [nc * C, 3H] in the sorted edge order
        (two autograd outputs over one storage, see below); `scale` [3H] multiplies the output channels."""
        T, S = len(w_rbf), BucketedBasis.WIDTH - 12
        # (stack first, scale once: two launches where the per-relation form had 2 T)
        stacked = lambda ps: ps if torch.is_tensor(ps) else torch.stack(list(ps))
        wt = (stacked(w_rbf) * scale[None, :, None]).transpose(1, 2)                         # [T, R, 3H]
        need = (self.nb - 1) * S + BucketedBasis.WIDTH                                       # rows the windows reach
        wt = F.pad(wt, (0, 0, 5, max(need - 5 - wt.size(1), 0)))                             # centre k sits at row k + 5
        win = wt.unfold(1, BucketedBasis.WIDTH, S)[:, :self.nb]                              # [T, nb, 3H, 32]
        win = win.permute(0, 1, 3, 2).reshape(T * self.nb, BucketedBasis.WIDTH, -1)
        bias = stacked(b_rbf) * scale[None, :]                                               # [T, 3H]
        if self.key_w is not None:
            wc = GatherRows.apply(win.reshape(win.size(0), -1), self.key_w).view(-1, win.size(1), win.size(2))
            bc = GatherRows.apply(bias, self.key_b)
        else:
            wc = win.index_select(0, self.group)
            bc = bias.index_select(0, self.group // self.nb)
        # One product per chunk with the bias added in its epilogue (csrc/band_product.hip); R comes back TWICE -- the same
        # storage under two autograd outputs: the message algebra reads the first, its backward (a second consumer of R in
        # the graph) the second, so their two [E,3H] gradients reach BandP.backward separately and are added while the
        # gradient products read them -- not by an elementwise launch over 610 MB per layer.
        R, R2 = band_product(self.phi, wc, bc)
        return R.reshape(-1, wc.size(2)), R2.reshape(-1, wc.size(2))


# ---- batched products with a 32-wide side (csrc/band_product.hip): rbf_proj on the bucketed basis and its two derivatives ------
def _band_kernels(A, N):
    return (A.is_cuda and A.dtype == torch.float32 and A.size(2) == 32
            and bool(_lib.load().hermnet_band_product_supported(int(A.size(1)), int(N))) and A.size(1) % 16 == 0)


def _sum2(g1, g2):
    return g1 if g2 is None else g1 + g2


class BandP(torch.autograd.Function):
    """(A [nc,C,32], B [nc,32,N], bias [nc,N] | None, side1, side2) -> (out, out) with out[c] = A[c] B[c] + bias[c]:
    `hermnet_band_product`.  The two outputs share their storage; a caller with two consumers of `out` in the graph hands each
    its own output, and the gradients arrive here unsummed (g1, g2) -- BandQ / BandS add them while reading.  The backward
    forms the gradient of A and passes (g1, g2) on to the stand-ins of `_BandParams` (the parameter side, a node of its own:
    see `_TallBmmParams`).  P, Q, S are closed under differentiation, so create_graph=True differentiates the backward again."""

    @staticmethod
    def forward(ctx, A, B, bias, side1, side2, held):
        from .ops import _stream
        ctx.save_for_backward(B)
        ctx.side = side1 is not None
        ctx.held = held
        ctx.set_materialize_grads(False)
        nc, C, N = A.size(0), A.size(1), B.size(2)
        if _band_kernels(A, N):
            out = torch.empty(nc, C, N, dtype=A.dtype, device=A.device)
            P = _lib.ptr
            _lib.check(_lib.load().hermnet_band_product(P(A), P(B), P(bias), nc, C, N, P(out), _stream()), "hermnet_band_product")
        else:
            out = torch.bmm(A, B) if bias is None else torch.baddbmm(bias[:, None, :], A, B)
        return out, out.view_as(out)

    @staticmethod
    def backward(ctx, g1, g2):
        (B,) = ctx.saved_tensors
        if g1 is None:
            g1, g2 = g2, None
        if g1 is None:
            return None, None, None, None, None, None
        if ctx.side and ctx.needs_input_grad[0] and not torch.is_grad_enabled() and _band_kernels(ctx.held.t, B.size(2)):
            # the pass that differentiates the parameters (no graph is being recorded): gA and the parameter side's (gB, gbias)
            # from ONE pass over g1 (+ g2); `_BandParams.backward` picks its share up from the box it shares with this node
            from .ops import _stream
            A, g1, g2 = ctx.held.t, _c(g1), _c(g2)
            nc, C, N = g1.shape
            gA = torch.empty(nc, C, 32, dtype=A.dtype, device=A.device)
            gB = torch.empty(nc, 32, N, dtype=A.dtype, device=A.device)
            gb = torch.empty(nc, N, dtype=A.dtype, device=A.device)
            P = _lib.ptr
            _lib.check(_lib.load().hermnet_band_product_grads(P(A), P(B), P(g1), P(g2), nc, C, N, P(gA), P(gB), P(gb), _stream()),
                       "hermnet_band_product_grads")
            ctx.held.ready = (g1, g2, gB, gb)
            return gA, None, None, g1, g2, None
        gA = BandQ.apply(g1, g2, B) if ctx.needs_input_grad[0] else None
        return gA, None, None, (g1 if ctx.side else None), (g2 if ctx.side else None), None


class _BandParams(torch.autograd.Function):
    """(B, bias | None; A held) -> two stand-ins of A B; the backward forms (gB, gbias) = BandS(A, g1, g2)."""

    @staticmethod
    def forward(ctx, B, bias, held):
        ctx.held = held
        ctx.has_bias = bias is not None
        ctx.set_materialize_grads(False)
        A = held.t
        shape = (A.size(0), A.size(1), B.size(2))
        return _stand_in(shape, A), _stand_in(shape, A)

    @staticmethod
    def backward(ctx, g1, g2):
        A = ctx.held.t
        if g1 is None:
            g1, g2 = g2, None
        if g1 is None:
            return None, None, None
        ready, ctx.held.ready = ctx.held.ready, None
        if ready is not None and ready[0] is g1 and ready[1] is g2:          # formed by BandP.backward's single pass
            gB, gb = ready[2], ready[3]
        else:
            gB, gb = BandS.apply(A, g1, g2)
        return gB, (gb if ctx.has_bias else None), None


def band_product(A, B, bias):
    """(out, out): out[c] = A[c] B[c] + bias[c] (`BandP`), the parameter gradients in a node of their own."""
    A, B, bias = _c(A), _c(B), _c(bias)
    held = _Held(A)
    s1, s2 = _BandParams.apply(B, bias, held) if _wants_grad(B, bias) else (None, None)
    return BandP.apply(A, B, bias, s1, s2, held)


class BandQ(torch.autograd.Function):
    """(g1, g2 | None [nc,C,N], B [nc,32,N]) -> gA [nc,C,32] = (g1 + g2) B^T: `hermnet_band_product_grad_a`."""

    @staticmethod
    def forward(ctx, g1, g2, B):
        from .ops import _stream
        g1, g2, B = _c(g1), _c(g2), _c(B)
        ctx.save_for_backward(g1, g2, B)
        nc, C, N = g1.shape
        if _band_kernels(B.new_empty(0, C, 32), N):
            gA = torch.empty(nc, C, 32, dtype=g1.dtype, device=g1.device)
            P = _lib.ptr
            _lib.check(_lib.load().hermnet_band_product_grad_a(P(g1), P(g2), P(B), nc, C, N, P(gA), _stream()),
                       "hermnet_band_product_grad_a")
            return gA
        return torch.bmm(_sum2(g1, g2), B.transpose(1, 2))

    @staticmethod
    def backward(ctx, c):
        g1, g2, B = ctx.saved_tensors
        cg = band_product(c, B, None)[0] if (ctx.needs_input_grad[0] or ctx.needs_input_grad[1]) else None
        cB = BandS.apply(c, g1, g2)[0] if ctx.needs_input_grad[2] else None
        return (cg if ctx.needs_input_grad[0] else None), (cg if g2 is not None and ctx.needs_input_grad[1] else None), cB


class BandS(torch.autograd.Function):
    """(A [nc,C,32], g1, g2 | None [nc,C,N]) -> (gB [nc,32,N] = A^T (g1 + g2), gbias [nc,N] = column sums of g1 + g2):
    `hermnet_band_product_grad_b`."""

    @staticmethod
    def forward(ctx, A, g1, g2):
        from .ops import _stream
        A, g1, g2 = _c(A), _c(g1), _c(g2)
        ctx.save_for_backward(A, g1, g2)
        nc, C, N = g1.shape
        if _band_kernels(A, N):
            gB = torch.empty(nc, 32, N, dtype=A.dtype, device=A.device)
            gb = torch.empty(nc, N, dtype=A.dtype, device=A.device)
            P = _lib.ptr
            _lib.check(_lib.load().hermnet_band_product_grad_b(P(A), P(g1), P(g2), nc, C, N, P(gB), P(gb), _stream()),
                       "hermnet_band_product_grad_b")
            return gB, gb
        g = _sum2(g1, g2)
        return torch.bmm(A.transpose(1, 2), g), g.sum(1)

    @staticmethod
    def backward(ctx, cB, cb):
        A, g1, g2 = ctx.saved_tensors
        cA = BandQ.apply(g1, g2, cB) if ctx.needs_input_grad[0] else None
        cg = band_product(A, cB, cb)[0] if (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]) else None
        return cA, (cg if ctx.needs_input_grad[1] else None), (cg if g2 is not None and ctx.needs_input_grad[2] else None)


class TallLinear(torch.autograd.Function):
    """y = a @ w.T + b for a TALL `a` [K, R] (K = the edges of a relation, ~1e5) and a small w [O, R].  The forward and
    the input gradient are ordinary GEMMs; the WEIGHT gradient g.T @ a reduces over K into an [O, R] result -- three
    output tiles for the whole GPU when left to the library (0.55 ms per call at K = 129k, 23 TFLOP/s).  Written as a
    batched product over K-chunks plus a sum it fills the chip.  The backward is differentiable torch code, so
    create_graph=True differentiates it again."""

    CHUNK = 2048

    @staticmethod
    def forward(ctx, a, w, b):
        ctx.save_for_backward(a, w)
        return torch.addmm(b, a, w.t())

    @staticmethod
    def backward(ctx, g):
        a, w = ctx.saved_tensors
        ga = g @ w if ctx.needs_input_grad[0] else None
        gw = gb = None
        if ctx.needs_input_grad[1]:
            K, C = a.size(0), TallLinear.CHUNK
            n = K // C
            gw = g[n * C:].t() @ a[n * C:]
            if n > 0:
                gw = gw + torch.bmm(g[:n * C].view(n, C, -1).transpose(1, 2), a[:n * C].view(n, C, -1)).sum(0)
        if ctx.needs_input_grad[2]:
            gb = g.sum(0)
        return ga, gw, gb


_SPLIT_K_ROWS = 2048          # target rows per chunk (tests shrink it)


def _split_k_chunk(K):
    """A chunk length that divides K (so the chunked views are free), between an eighth of the target and the target."""
    for n in range(max(1, (K + _SPLIT_K_ROWS - 1) // _SPLIT_K_ROWS), K // max(1, _SPLIT_K_ROWS // 8) + 1):
        if K % n == 0:
            return K // n
    return 0


def _gram_over_rows(a, g):
    """a^T g per relation, [T,K,I] x [T,K,O] -> [T,I,O], as a batched product over K-chunks plus a sum (differentiable torch
    ops): the reduction over ~2e4 rows into a [128..384]^2 result is a handful of output tiles for the whole chip when left
    to the library as ONE product (14-24 TFLOP/s measured on the training step's node-level weight gradients)."""
    T, K, I = a.shape
    O = g.size(2)
    C = _split_k_chunk(K) if a.is_contiguous() and g.is_contiguous() else 0
    if C == 0 or C == K:
        return torch.bmm(a.transpose(1, 2), g)
    n = K // C
    return torch.bmm(a.view(T * n, C, I).transpose(1, 2), g.view(T * n, C, O)).view(T, n, I, O).sum(1)


class _ColSum(torch.autograd.Function):
    """g.sum(1) for a tall contiguous fp32 [T,K,O] on the device: `hermnet_col_sum` (one pass at streaming rate into a few
    partials per slice) + one small sum.  Its derivative is a broadcast."""

    ROWS = 64

    @staticmethod
    def forward(ctx, g):
        from .ops import _stream
        T, K, O = g.shape
        ctx.K = K
        nb = (K + _ColSum.ROWS - 1) // _ColSum.ROWS
        part = torch.empty(T, nb, O, dtype=g.dtype, device=g.device)
        _lib.check(_lib.load().hermnet_col_sum(_lib.ptr(g), T, K, O, _ColSum.ROWS, _lib.ptr(part), _stream()), "hermnet_col_sum")
        return part.sum(1)

    @staticmethod
    def backward(ctx, c):
        return c[:, None, :].expand(-1, ctx.K, -1)


def _col_sum_over_rows(g):
    """g.sum(1) for a tall [T,K,O]: two stages over row chunks (torch's single reduction over a non-innermost axis runs at
    0.75 TB/s on these shapes -- 122 us for [3,20172,384] -- the two-stage torch form at 35 us, the kernel at streaming rate)."""
    T, K, O = g.shape
    if g.is_cuda and g.dtype == torch.float32 and g.is_contiguous() and O % 4 == 0 and O <= 1024 and K >= 256:
        return _ColSum.apply(g)
    C = _split_k_chunk(K) if g.is_contiguous() else 0
    if C == 0 or C == K:
        return g.sum(1)
    return g.view(T, K // C, C, O).sum(2).sum(1)


# ---- the PARAMETER side of a product's backward as a graph node of its own -------------------------------------------------------
# `ctx.needs_input_grad` of a custom Function says which inputs require grad, not which gradients THIS backward pass needs: in
# the force pass (autograd.grad(E, pos, create_graph=True); /root/reference/example/dist_train.py:90-92) no parameter gradient
# is asked for, yet a Function that computes "ga, gw, gb" in one backward forms them all -- every weight gradient of the step a
# second time (19 tall reductions and their sums per layer pair, 3.3 ms of 42).  The engine prunes NODES: so the parameter
# gradients live in a node of their own whose forward output is a zero-stride stand-in of the product's shape (no memory, no
# launch), consumed by the product's node, which hands it its own incoming gradient.  In a pass that differentiates no
# parameter the node is never run.  The other operand reaches the node in a `_Held` box, not as an input: an input edge to
# (something that leads to) the positions would make the node part of the force pass; the operand keeps its own history, so
# a backward of this backward (parameter gradients under create_graph=True) still differentiates through it.
_ZERO = {}


class _Held(object):
    __slots__ = ("t", "ready")

    def __init__(self, t):
        self.t = t
        self.ready = None



def _stand_in(shape, ref):
    z = _ZERO.get((ref.device, ref.dtype))
    if z is None:
        z = _ZERO[(ref.device, ref.dtype)] = torch.zeros((), dtype=ref.dtype, device=ref.device)
    return z.expand(shape)


def _wants_grad(*ts):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


class _TallBmmRows(torch.autograd.Function):
    """y[t] = a[t] @ w[t] (+ b[t]); the backward forms the gradient of `a` only and passes its incoming gradient on to `side`
    (the stand-in of `_TallBmmParams`, or None when no parameter requires grad)."""

    @staticmethod
    def forward(ctx, a, w, b, side):
        ctx.save_for_backward(w)
        ctx.side = side is not None
        return torch.bmm(a, w) if b is None else torch.baddbmm(b[:, None, :], a, w)

    @staticmethod
    def backward(ctx, g):
        (w,) = ctx.saved_tensors
        g = g.contiguous()
        ga = tall_bmm(g, w.transpose(1, 2), None) if ctx.needs_input_grad[0] else None
        return ga, None, None, (g if ctx.side else None)


class _TallBmmParams(torch.autograd.Function):
    """(w, b | None; a held) -> stand-in of a @ w; the backward forms gw = a^T g (`_gram_over_rows`) and gb (column sums)."""

    @staticmethod
    def forward(ctx, w, b, held):
        ctx.held = held
        ctx.has_b = b is not None
        a = held.t
        return _stand_in((a.size(0), a.size(1), w.size(2)), a)

    @staticmethod
    def backward(ctx, g):
        a = ctx.held.t
        gw = _gram_over_rows(a, g) if ctx.needs_input_grad[0] else None
        gb = _col_sum_over_rows(g) if ctx.has_b and ctx.needs_input_grad[1] else None
        return gw, gb, None


def tall_bmm(a, w, b):
    """y[t] = a[t] @ w[t] (+ b[t]) for TALL a [T,K,I] (K = the rows of a relation) and small w [T,I,O]: the node-level
    linears of the training step (rmnet.py:52, 94-100) and the read-out.  Forward and input gradient are ordinary batched
    GEMMs; the weight gradient goes through `_gram_over_rows`, in a graph node of its own (see above).  The input gradient is
    again a `tall_bmm` (its own weight gradient, needed by the second-order pass, is the same tall reduction)."""
    a = a.contiguous()
    side = _TallBmmParams.apply(w, b, _Held(a)) if _wants_grad(w, b) else None
    return _TallBmmRows.apply(a, w, b, side)


class TallBmm(object):
    apply = staticmethod(tall_bmm)


def tall_linear(a, weight, bias=None):
    """nn.Linear semantics (y = a W^T + b) for a tall 2-D a through TallBmm."""
    y = TallBmm.apply(a[None], weight.t()[None], None if bias is None else bias[None])
    return y[0]


# ---- node-level stages of the training step as one launch per order of differentiation (csrc/train_node_kernels.hip) ----
def _node_op(op, ins, outs, rows, H, c0=0.0, c1=0.0):
    from . import _lib
    from .ops import _stream
    ai = (ctypes.c_void_p * len(ins))(*[None if t is None else t.data_ptr() for t in ins])
    ao = (ctypes.c_void_p * len(outs))(*[None if t is None else t.data_ptr() for t in outs])
    _lib.check(_lib.load().hermnet_train_node_op(op, ai, len(ins), ao, len(outs), rows, H, c0, c1, _stream()),
               "hermnet_train_node_op")


def node_kernels_ok(x):
    """The fused node-level stages take fp32 GPU rows whose width is a multiple of 4 (<= 1024)."""
    return x.is_cuda and x.dtype == torch.float32 and x.size(-1) % 4 == 0 and x.size(-1) <= 1024


class SiLU2(torch.autograd.Function):
    """x sigmoid(x), differentiable twice with ONE launch per order (torch's own double backward of silu is ~10)."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.nn.functional.silu(x)

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        return _SiLUBwd.apply(gy, x)


class _SiLUBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gy, x):
        gy, x = _c(gy), _c(x)
        ctx.save_for_backward(gy, x)
        gx = torch.empty_like(x)
        _node_op(1, [gy, x], [gx], x.numel() // 4, 4)
        return gx

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, u):
        gy, x = ctx.saved_tensors
        cg, cx = torch.empty_like(x), torch.empty_like(x)
        _node_op(2, [_c(u), gy, x], [cg, cx], x.numel() // 4, 4)
        return cg, cx


class LayerNorm2(torch.autograd.Function):
    """LayerNorm without affine over the last axis (rows [N,H]), differentiable twice with one launch per order."""

    @staticmethod
    def forward(ctx, x, eps):
        ctx.save_for_backward(x)
        ctx.eps = eps
        return torch.nn.functional.layer_norm(x, (x.size(-1),), eps=eps)

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        return _LayerNormBwd.apply(gy, x, ctx.eps), None


class _LayerNormBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gy, x, eps):
        gy, x = _c(gy), _c(x)
        ctx.save_for_backward(gy, x)
        ctx.eps = eps
        gx = torch.empty_like(x)
        _node_op(9, [gy, x], [gx], x.numel() // x.size(-1), x.size(-1), eps)
        return gx

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, v):
        gy, x = ctx.saved_tensors
        cg, cx = torch.empty_like(x), torch.empty_like(x)
        _node_op(10, [_c(v), gy, x], [cg, cx], x.numel() // x.size(-1), x.size(-1), ctx.eps)
        return cg, cx, None


class Residual(torch.autograd.Function):
    """x1 = m c (x + dx), vec1 = m (vec + dvec): the residual behind the message (rmnet.py:24-26) with the row mask of
    hermnet.py:51 -- one launch instead of five; its backward `_MaskScale` (one launch) serves both summands and is its own
    backward."""

    @staticmethod
    def forward(ctx, x, dx, vec, dv, mask, c):
        x, dx, vec, dv = _c(x), _c(dx), _c(vec), _c(dv)
        R, H = x.shape
        ctx.save_for_backward(mask)
        ctx.c, ctx.has_vec = c, vec is not None
        ctx.set_materialize_grads(False)
        x1, v1 = torch.empty_like(x), torch.empty_like(dv)
        _node_op(11, [x, dx, vec, dv, mask], [x1, v1], R, H, c)
        return x1, v1

    @staticmethod
    def backward(ctx, g1, gv1):
        (mask,) = ctx.saved_tensors
        gx, gv = _MaskScale.apply(g1, gv1, mask, ctx.c)
        return gx, gx, (gv if ctx.has_vec else None), gv, None, None


class _MaskScale(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g1, gv1, mask, c):
        ref = g1 if g1 is not None else gv1
        R, H = ref.size(0), ref.size(-1)
        ctx.save_for_backward(mask)
        ctx.c = c
        ctx.set_materialize_grads(False)
        o1 = torch.empty(R, H, dtype=ref.dtype, device=ref.device)
        o2 = torch.empty(R, 3, H, dtype=ref.dtype, device=ref.device)
        _node_op(12, [_c(g1), _c(gv1), mask], [o1, o2], R, H, c)
        return o1, o2

    @staticmethod
    def backward(ctx, u1, uv):
        (mask,) = ctx.saved_tensors
        if u1 is None and uv is None:
            return None, None, None, None
        a, b = _MaskScale.apply(u1, uv, mask, ctx.c)
        return a, b, None, None


class UpdateMid(torch.autograd.Function):
    """(vp [R,3,2H] = (v1 | v2), xt [R,H]) -> (vec_dot [R,H], [xt | sqrt(sum_d v2^2 + eps)] [R,2H])   (rmnet.py:96-99)."""

    @staticmethod
    def forward(ctx, vp, xt, c, eps):
        vp, xt = _c(vp), _c(xt)
        R, H = xt.shape
        ctx.save_for_backward(vp)
        ctx.c, ctx.eps = c, eps
        vdot, xin = torch.empty_like(xt), torch.empty(R, 2 * H, dtype=xt.dtype, device=xt.device)
        _node_op(3, [vp, xt], [vdot, xin], R, H, c, eps)
        return vdot, xin

    @staticmethod
    def backward(ctx, g_vdot, g_xin):
        (vp,) = ctx.saved_tensors
        g_vp, g_xt = _UpdateMidBwd.apply(g_vdot, g_xin, vp, ctx.c, ctx.eps)
        return g_vp, g_xt, None, None


class _UpdateMidBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g_vdot, g_xin, vp, c, eps):
        g_vdot, g_xin = _c(g_vdot), _c(g_xin)
        R, H = g_vdot.shape
        ctx.save_for_backward(g_vdot, g_xin, vp)
        ctx.c, ctx.eps = c, eps
        ctx.set_materialize_grads(False)
        g_vp, g_xt = torch.empty_like(vp), torch.empty_like(g_vdot)
        _node_op(4, [g_vdot, g_xin, vp], [g_vp, g_xt], R, H, c, eps)
        return g_vp, g_xt

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, u_vp, u_xt):
        g_vdot, g_xin, vp = ctx.saved_tensors
        R, H = g_vdot.shape
        c_a, c_b, c_vp = torch.empty_like(g_vdot), torch.empty_like(g_xin), torch.empty_like(vp)
        _node_op(5, [_c(u_vp), _c(u_xt), g_vdot, g_xin, vp], [c_a, c_b, c_vp], R, H, ctx.c, ctx.eps)
        return c_a, c_b, c_vp, None, None


class UpdateOut(torch.autograd.Function):
    """(q [R,3H], vec_dot, vp, xt, vt [R,3,H], mask [R] | None) -> (m (xt + (q1 + q2 vec_dot) s), m (vt + q3 v1))
    (rmnet.py:101-107, 29-31; the row mask of hermnet.py:51,56-57)."""

    @staticmethod
    def forward(ctx, q, vdot, vp, xt, vt, mask, s):
        q, vdot, vp, xt, vt = _c(q), _c(vdot), _c(vp), _c(xt), _c(vt)
        R, H = xt.shape
        ctx.save_for_backward(q, vdot, vp, mask)
        ctx.s = s
        ctx.set_materialize_grads(False)
        xo, vo = torch.empty_like(xt), torch.empty_like(vt)
        _node_op(6, [q, vdot, vp, xt, vt, mask], [xo, vo], R, H, s)
        return xo, vo

    @staticmethod
    def backward(ctx, gx, gv):
        q, vdot, vp, mask = ctx.saved_tensors
        if gx is None:
            gx = torch.zeros_like(vdot)
        g_q, g_vdot, g_vp, g_xt, g_vt = _UpdateOutBwd.apply(gx, gv, q, vdot, vp, mask, ctx.s)
        return g_q, g_vdot, g_vp, g_xt, g_vt, None, None


class _UpdateOutBwd(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gx, gv, q, vdot, vp, mask, s):
        gx, gv = _c(gx), _c(gv)
        R, H = gx.shape
        ctx.save_for_backward(gx, gv, q, vdot, vp, mask)
        ctx.s = s
        ctx.set_materialize_grads(False)
        g_q, g_vdot, g_vp = torch.empty_like(q), torch.empty_like(vdot), torch.empty_like(vp)
        g_xt = torch.empty_like(gx)
        g_vt = torch.empty(R, 3, H, dtype=gx.dtype, device=gx.device)
        _node_op(7, [gx, gv, q, vdot, vp, mask], [g_q, g_vdot, g_vp, g_xt, g_vt], R, H, s)
        return g_q, g_vdot, g_vp, g_xt, g_vt

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, c_q, c_vdot, c_vp, c_xt, c_vt):
        gx, gv, q, vdot, vp, mask = ctx.saved_tensors
        R, H = gx.shape
        d_gx, d_q, d_vdot, d_vp = torch.empty_like(gx), torch.empty_like(q), torch.empty_like(vdot), torch.empty_like(vp)
        d_gv = torch.empty(R, 3, H, dtype=gx.dtype, device=gx.device)
        _node_op(8, [_c(c_q), _c(c_vdot), _c(c_vp), _c(c_xt), _c(c_vt), gx, gv, q, vdot, vp, mask],
                 [d_gx, d_gv, d_q, d_vdot, d_vp], R, H, ctx.s)
        return d_gx, (d_gv if gv is not None else None), d_q, d_vdot, d_vp, None, None


def _edge_message_torch(X, R, V, U):
    """The per-edge message algebra of rmnet.py:58-66 in differentiable torch ops (host tensors, widths that are not a
    multiple of 4): S = Xs Rs, M_d = (Xb Rb) U_d + V_d (Xa Ra)."""
    H = X.size(1) // 3
    # (unbind of the [E,3,H] view: its backward is ONE stack, where three column slices each zero-fill a full gradient)
    xs, xa, xb = X.view(-1, 3, H).unbind(1)
    rs, ra, rb = R.view(-1, 3, H).unbind(1)
    M = (xb * rb)[:, None, :] * U[:, :, None]
    if V is not None:
        M = torch.addcmul(M, V, (xa * ra)[:, None, :])
    return xs * rs, M


def _c(t):
    return None if t is None else t.contiguous()


class EdgeMessage(torch.autograd.Function):
    """(X [E,3H], R [E,3H], V [E,3,H] | None, U [E,3]) -> (S [E,H], M [E,3,H]) on the GPU: `hermnet_edge_message_fwd`.
    The map is multilinear, so its backward (`EdgeMessageGrad`) and the backward of that are per-edge products and
    channel sums again -- three streaming kernels (csrc/train_kernels.hip) for what the autograd graph of the torch
    expression spreads over ~35 elementwise / reduction launches per layer and order."""

    @staticmethod
    def forward(ctx, X, R, V, U):
        from . import _lib
        from .ops import _stream
        X, R, V, U = _c(X), _c(R), _c(V), _c(U)
        E, H = X.size(0), X.size(1) // 3
        S = torch.empty(E, H, dtype=X.dtype, device=X.device)
        M = torch.empty(E, 3, H, dtype=X.dtype, device=X.device)
        P = _lib.ptr
        _lib.check(_lib.load().hermnet_edge_message_fwd(P(X), P(R), P(V), P(U), E, H, None, None, None, P(S), P(M), _stream()),
                   "hermnet_edge_message_fwd")
        ctx.save_for_backward(X, R, V, U)
        return S, M

    @staticmethod
    def backward(ctx, GS, GM):
        X, R, V, U = ctx.saved_tensors
        gX, gR, gV, gU = EdgeMessageGrad.apply(GS, GM, X, R, V, U)
        return gX, gR, gV, gU


class EdgeMessageGrad(torch.autograd.Function):
    """First-order cotangents of `EdgeMessage` (`hermnet_edge_message_bwd`); differentiable once more
    (`hermnet_edge_message_bwd2`), which is what `autograd.grad(E, pos, create_graph=True)` + `loss.backward()` need."""

    @staticmethod
    def forward(ctx, GS, GM, X, R, V, U):
        from . import _lib
        from .ops import _stream
        E, H = X.size(0), X.size(1) // 3
        GS = torch.zeros(E, H, dtype=X.dtype, device=X.device) if GS is None else _c(GS)
        GM = torch.zeros(E, 3, H, dtype=X.dtype, device=X.device) if GM is None else _c(GM)
        gX, gR = torch.empty_like(X), torch.empty_like(R)
        gV = None if V is None else torch.empty_like(V)
        gU = torch.empty_like(U)
        P = _lib.ptr
        _lib.check(_lib.load().hermnet_edge_message_bwd(P(GS), P(GM), P(X), P(R), P(V), P(U), E, H, None, None, None, None,
                                                        P(gX), P(gR), P(gV), P(gU), _stream()), "hermnet_edge_message_bwd")
        ctx.save_for_backward(GS, GM, X, R, V, U)
        return gX, gR, gV, gU

    @staticmethod
    def backward(ctx, cX, cR, cV, cU):
        from . import _lib
        from .ops import _stream
        GS, GM, X, R, V, U = ctx.saved_tensors
        E, H = X.size(0), X.size(1) // 3
        cX, cR, cV, cU = _c(cX), _c(cR), _c(cV), _c(cU)
        dGS, dGM = torch.empty_like(GS), torch.empty_like(GM)
        dX, dR = torch.empty_like(X), torch.empty_like(R)
        dV = None if V is None else torch.empty_like(V)
        dU = torch.empty_like(U)
        P = _lib.ptr
        _lib.check(_lib.load().hermnet_edge_message_bwd2(P(cX), P(cR), P(cV), P(cU), P(GS), P(GM), P(X), P(R), P(V), P(U),
                                                         E, H, None, None, None, None, P(dGS), P(dGM), P(dX), P(dR), P(dV), P(dU),
                                                         _stream()), "hermnet_edge_message_bwd2")
        return dGS, dGM, dX, dR, dV, dU


def _segsum(x, key):
    """out[r] = sum of x[k] over the edges k of row r (`hermnet_segment_sum`: rows gathered inside the sum, list order)."""
    from .ops import _stream
    x = x.contiguous()
    out = torch.empty((key.n_rows,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    P = _lib.ptr
    _lib.check(_lib.load().hermnet_segment_sum(P(x), P(key.perm), P(key.rowptr), key.n_rows, x[0].numel(), P(out), _stream()),
               "hermnet_segment_sum")
    return out


def _rows_buffer(R, r_rows, keys):
    """Buffer for a gradient of R that the edge kernels write row by row (at r_rows): rows no edge points to must read
    zero -- all of them zeroed, or only the listed padding rows (keys[4])."""
    if r_rows is None:
        return torch.empty_like(R)
    pad = keys[4] if len(keys) > 4 else None
    if pad is None:
        return torch.zeros_like(R)
    out = torch.empty_like(R)
    if pad.numel() > 0:
        out.index_fill_(0, pad, 0.0)
    return out


class MessageAlgebra(torch.autograd.Function):
    """(xh [T N,3H], vec [N,3,H] | None, R [E,3H], U [E,3]) -> (dx [N,H], dvec [N,3,H]): gather of x_j / vec_j, the
    per-edge algebra and the aggregation of rmnet.py:58-73 with NODE-level inputs and outputs.  The kernels of
    `EdgeMessage` read their gathered operands through row indices (x_j = xh[(relation, source)], vec_j = vec[source],
    cotangents = g[target]) and the row sums follow inside the function, so no [E, 3H] copy of a gathered operand and
    no per-edge gradient ever enters the autograd graph: what two graph nodes share and the engine has to add up is
    node-sized.  Twice differentiable through `MessageAlgebraGrad`.  keys = (targets, sources, (relation, source), rows
    of R or None, padding rows of R or None)."""

    @staticmethod
    def forward(ctx, xh, vec, R, U, keys, R2=None):
        from .ops import _stream
        k_tgt, k_all, k_xh, r_rows = keys[:4]
        xh, vec, R, U = _c(xh), _c(vec), _c(R), _c(U)
        E, H = U.size(0), R.size(1) // 3
        P = _lib.ptr
        # R2: the same values as R under a second autograd identity (BandP's second output) -- the backward below is R's
        # second consumer in the graph and differentiates through R2, so that R's two gradients are not summed by the engine
        ctx.save_for_backward(xh, vec, R if R2 is None else R2, U)
        ctx.keys = keys
        if _row_sums_inside():
            # (the sums over a target's edges stay in registers: no [E,4H] round trip through HBM)
            dx = torch.empty(k_tgt.n_rows, H, dtype=R.dtype, device=R.device)
            dv = torch.empty(k_tgt.n_rows, 3, H, dtype=R.dtype, device=R.device)
            _lib.check(_lib.load().hermnet_edge_message_fwd_rows(
                P(xh), P(R), P(vec), P(U), E, H, P(k_xh.idx), P(k_all.idx), P(r_rows), P(k_tgt.rowptr), P(k_tgt.perm),
                k_tgt.n_rows, P(dx), P(dv), _stream()), "hermnet_edge_message_fwd_rows")
            return dx, dv
        S = torch.empty(E, H, dtype=R.dtype, device=R.device)
        M = torch.empty(E, 3, H, dtype=R.dtype, device=R.device)
        _lib.check(_lib.load().hermnet_edge_message_fwd(P(xh), P(R), P(vec), P(U), E, H, P(k_xh.idx), P(k_all.idx), P(r_rows),
                                                        P(S), P(M), _stream()), "hermnet_edge_message_fwd")
        return _segsum(S, k_tgt), _segsum(M, k_tgt)

    @staticmethod
    def backward(ctx, g_dx, g_dv):
        xh, vec, R, U = ctx.saved_tensors
        g_xh, g_vec, gR, gU = MessageAlgebraGrad.apply(g_dx, g_dv, xh, vec, R, U, ctx.keys)
        return g_xh, g_vec, gR, gU, None, None


class MessageAlgebraGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g_dx, g_dv, xh, vec, R, U, keys):
        from .ops import _stream
        k_tgt, k_all, k_xh, r_rows = keys[:4]
        E, H = U.size(0), R.size(1) // 3
        g_dx = torch.zeros(k_tgt.n_rows, H, dtype=R.dtype, device=R.device) if g_dx is None else _c(g_dx)
        g_dv = torch.zeros(k_tgt.n_rows, 3, H, dtype=R.dtype, device=R.device) if g_dv is None else _c(g_dv)
        # (R kept in another edge order with padding rows: rows no edge points to get no gradient)
        gR = _rows_buffer(R, r_rows, keys)
        gU = torch.empty_like(U)
        P = _lib.ptr
        ctx.save_for_backward(g_dx, g_dv, xh, vec, R, U)
        ctx.keys = keys
        if _row_sums_inside() and _groups_of_sources(k_xh, k_all):
            # groups = the (relation, source) rows of xh: gX arrives summed, gV as one partial sum per relation
            G, Ns = k_xh.n_rows, k_all.n_rows
            gX = torch.empty(G, 3 * H, dtype=R.dtype, device=R.device)
            gVp = None if vec is None else torch.empty(G, 3, H, dtype=R.dtype, device=R.device)
            _lib.check(_lib.load().hermnet_edge_message_bwd_rows(
                P(g_dx), P(g_dv), P(xh), P(R), P(vec), P(U), E, H, P(k_xh.idx), P(k_all.idx), P(k_tgt.idx), P(r_rows),
                P(k_xh.rowptr), P(k_xh.perm), G, P(gX), P(gR), P(gVp), P(gU), _stream()), "hermnet_edge_message_bwd_rows")
            return gX, (None if vec is None else gVp.view(G // Ns, Ns, 3, H).sum(0)), gR, gU
        gX = torch.empty(E, 3 * H, dtype=R.dtype, device=R.device)
        gV = None if vec is None else torch.empty(E, 3, H, dtype=R.dtype, device=R.device)
        _lib.check(_lib.load().hermnet_edge_message_bwd(P(g_dx), P(g_dv), P(xh), P(R), P(vec), P(U), E, H, P(k_xh.idx),
                                                        P(k_all.idx), P(k_tgt.idx), P(r_rows), P(gX), P(gR), P(gV), P(gU),
                                                        _stream()), "hermnet_edge_message_bwd")
        return _segsum(gX, k_xh), (None if vec is None else _segsum(gV, k_all)), gR, gU

    @staticmethod
    def backward(ctx, c_xh, c_vec, cR, cU):
        from .ops import _stream
        g_dx, g_dv, xh, vec, R, U = ctx.saved_tensors
        k_tgt, k_all, k_xh, r_rows = ctx.keys[:4]
        E, H = U.size(0), R.size(1) // 3
        c_xh, c_vec, cR, cU = _c(c_xh), _c(c_vec), _c(cR), _c(cU)
        new = lambda *shape: torch.empty(*shape, dtype=R.dtype, device=R.device)
        dR = _rows_buffer(R, r_rows, ctx.keys)
        P = _lib.ptr
        if _row_sums_inside() and _groups_of_sources(k_xh, k_all):
            G, Ns = k_xh.n_rows, k_all.n_rows
            dGS, dGM, dX, dU = new(E, H), new(E, 3, H), new(G, 3 * H), new(E, 3)
            dVp = None if vec is None else new(G, 3, H)
            _lib.check(_lib.load().hermnet_edge_message_bwd2_rows(
                P(c_xh), P(cR), P(c_vec), P(cU), P(g_dx), P(g_dv), P(xh), P(R), P(vec), P(U), E, H, P(k_xh.idx), P(k_all.idx),
                P(k_tgt.idx), P(r_rows), P(k_xh.rowptr), P(k_xh.perm), G, P(dGS), P(dGM), P(dX), P(dR), P(dVp), P(dU),
                _stream()), "hermnet_edge_message_bwd2_rows")
            return (_segsum(dGS, k_tgt), _segsum(dGM, k_tgt), dX, (None if vec is None else dVp.view(G // Ns, Ns, 3, H).sum(0)),
                    dR, dU, None)
        dGS, dGM, dX, dU = new(E, H), new(E, 3, H), new(E, 3 * H), new(E, 3)
        dV = None if vec is None else new(E, 3, H)
        _lib.check(_lib.load().hermnet_edge_message_bwd2(P(c_xh), P(cR), P(c_vec), P(cU), P(g_dx), P(g_dv), P(xh), P(R), P(vec),
                                                         P(U), E, H, P(k_xh.idx), P(k_all.idx), P(k_tgt.idx), P(r_rows), P(dGS),
                                                         P(dGM), P(dX), P(dR), P(dV), P(dU), _stream()),
                   "hermnet_edge_message_bwd2")
        return (_segsum(dGS, k_tgt), _segsum(dGM, k_tgt), _segsum(dX, k_xh), (None if vec is None else _segsum(dV, k_all)),
                dR, dU, None)


def _row_sums_inside():
    from . import switches
    return switches.train_row_sums


def _groups_of_sources(k_xh, k_all):
    """The (relation, source) groups of k_xh tile the source rows of k_all: T groups per source row."""
    return k_all.n_rows > 0 and k_xh.n_rows % k_all.n_rows == 0


def _train_kernels(t):
    return t.is_cuda and t.dtype == torch.float32 and (t.size(-1) // 3) % 4 == 0 and t.size(0) > 0


def edge_message(X, R, V, U):
    """S, M of the per-edge message algebra: the kernels on the GPU (fp32, width a multiple of 4), torch ops otherwise."""
    if _train_kernels(X):
        return EdgeMessage.apply(X, R, V, U)
    return _edge_message_torch(X, R, V, U)


def _run_lengths(sorted_keys, n):
    """How often each of 0 .. n-1 occurs in an ascending key vector (bincount without its host read)."""
    ends = torch.searchsorted(sorted_keys, torch.arange(1, n + 1, device=sorted_keys.device))
    return torch.diff(ends, prepend=ends.new_zeros(1))


def _row_keys(graph, T, Nt, Ns, Ek):
    """(key of the targets of the first Ek CSR edges [Nt target rows], key of their sources [Ns source rows], key of
    their (relation, source) rows of xh.view(T Ns, 3H), key of the residual rows or None) for `message_scatter_generic`,
    from the graph's CSR / CSC orders; built once per graph.  Nt = Ns for HVNet; HTNet has one target row per atom and
    pair relation and reads the residual from the atom's own source row (`graph.res_row`)."""
    keys = getattr(graph, "_row_keys", None)
    if keys is not None:
        return keys
    dev = graph.csr_rowptr.device
    rowptr = graph.csr_rowptr.long()
    nk = int(graph.type_rowptr_host[-1])
    lengths = rowptr[1:] - rowptr[:-1]
    lengths = torch.cat([lengths[:nk], lengths.new_zeros(Nt - nk)])          # edges into unknown-element rows: not summed
    # (no host reads here: sizes are passed where an op would read them back -- repeat_interleave -- and run lengths come
    # from the sorted keys, not from bincount, which reads its input's extrema back)
    tgt_row = torch.repeat_interleave(torch.arange(Nt, device=dev), lengths, output_size=Ek)
    k_tgt = _RowKey(tgt_row, None, lengths, Nt)
    src = graph.csr_src.long()
    crp, cpos = graph.csc_rowptr.long(), graph.csc_pos.long()               # groups (relation, source row) over CSR positions
    # all relations at once: sorted by (source row) = the T groups of a row merged; built by one stable sort
    order = torch.argsort(src[:Ek], stable=True)
    k_all = _RowKey(src[:Ek], order, _run_lengths(src[:Ek].index_select(0, order), Ns), Ns)
    # rows of xh.view(T * Ns, 3H): (relation of the edge's target, source row) -- the CSC groups themselves
    rel_of_edge = torch.bucketize(torch.arange(Ek, device=dev), graph.rel_edge_bounds_dev()[1:T + 1].long(), right=True)
    k_xh = _RowKey(rel_of_edge * Ns + src[:Ek], cpos[:Ek], crp[1:T * Ns + 1] - crp[:T * Ns], T * Ns)
    k_res = None
    if graph.res_row is not None:
        res = graph.res_row.long()
        order_r = torch.argsort(res, stable=True)
        k_res = _RowKey(res, order_r, _run_lengths(res.index_select(0, order_r), Ns), Ns)
    # rows of a known element (the others stay zero, hermnet.py:51): once per graph, not once per layer
    known = torch.arange(Nt, device=dev) < graph.type_rowptr[T:T + 1].long()
    graph._row_keys = (k_tgt, k_all, k_xh, k_res, known.to(torch.float32))
    return graph._row_keys


_SCALE = {}


def _message_scale(H, ref):
    """[3H] factors of the three message parts (1, 1/sqrt(3H), 1/sqrt(H); rmnet.py:64-66): a constant per width."""
    key = (H, ref.device, ref.dtype)
    sc = _SCALE.get(key)
    if sc is None:
        sc = ref.new_ones(3 * H)
        sc[H:2 * H] = 1 / math.sqrt(3.0 * H)
        sc[2 * H:] = 1 / math.sqrt(H)
        _SCALE[key] = sc
    return sc


def message_scatter_generic(xh, vec, x, edge, edge_embed, w_rbf, b_rbf, graph):
    """Same contract as the fused kernel (rmnet.py:24-26,55-73) for a MATERIALISED basis `edge_embed`
    [E,R] (CSR order): library GEMM per relation + gather / segmented-sum device ops, differentiable to any order by
    PyTorch autograd.  Path of train() mode (parameter gradients, create_graph=True) and of the optional
    radial bases."""
    T, Ns, H3 = xh.shape                                       # Ns source rows (rows of x / vec / xh[t])
    H = H3 // 3
    N = graph.N                                                # target rows (= Ns for HVNet; HTNet: one per atom and pair)
    # rows are relation-ordered and CSR is row-ordered: the edges of relation t are ONE contiguous CSR range
    # (no per-relation masks or gathers of the edge arrays)
    bucketed = isinstance(edge_embed, BucketedBasis)
    bounds = None if bucketed else graph.rel_edge_bounds()     # (per relation: the materialised-basis route only)
    Ek = graph.rel_edge_total()                                # edges whose target has a known element
    k_tgt, k_all, k_xh, k_res, known = _row_keys(graph, T, N, Ns, Ek)
    known = known.to(x.dtype)
    # the constant factors of the vector message (1/sqrt(3H) on the `a` part, 1/sqrt(H) on `b`, rmnet.py:64-66) ride on
    # the [3H, R] projection weights, not on per-edge tensors
    sc = _message_scale(H, x)
    dx = dv = None
    parts = []
    if bucketed:
        if Ek > 0:     # rbf_proj (rmnet.py:55) as one batched product on the bucketed basis; R stays in its sorted order
            R, R2 = edge_embed.project(w_rbf, b_rbf, sc)
            dx, dv = MessageAlgebra.apply(xh.reshape(T * Ns, 3 * H), vec, R, edge_embed.unit_vectors(edge, Ek),
                                          (k_tgt, k_all, k_xh, edge_embed.slot, edge_embed.pad), R2)
    else:
        # (split, not slices: the backward of a split is ONE cat, a slice's zero-fills the whole [E,R] gradient)
        emb = edge_embed.split([bounds[t + 1] - bounds[t] for t in range(T)] + [edge_embed.size(0) - Ek])
        for t in range(T):
            e0, e1 = bounds[t], bounds[t + 1]
            if e1 > e0:
                parts.append(TallLinear.apply(emb[t], w_rbf[t] * sc[:, None], b_rbf[t] * sc))   # rbf_proj, rmnet.py:55
    if parts:
        R = parts[0] if len(parts) == 1 else torch.cat(parts, 0)                   # [Ek, 3H]
        if _train_kernels(R):
            # gather x_j / vec_j (rmnet.py:58), x_j * rbfh and the vector message (:61-66), aggregation (:69-73): one
            # twice-differentiable function with node-level inputs and outputs (csrc/train_kernels.hip)
            dx, dv = MessageAlgebra.apply(xh.reshape(T * Ns, 3 * H), vec, R, edge[:Ek, :3], (k_tgt, k_all, k_xh, None, None))
        else:
            X = GatherRows.apply(xh.reshape(T * Ns, 3 * H), k_xh)                  # x_j of every edge, rmnet.py:58
            V = None if vec is None else GatherRows.apply(vec, k_all)
            S, M = edge_message(X, R, V, edge[:Ek, :3])
            dx = SumRows.apply(S, k_tgt)
            dv = SumRows.apply(M, k_tgt)
    if dx is None:                                             # no edge into a known element
        dx, dv = x.new_zeros(N, H), x.new_zeros(N, 3, H)
    # the residual (rmnet.py:24-26) reads the target atom's own row: the same row for HVNet, `res_row` for HTNet
    xr = x if k_res is None else GatherRows.apply(x, k_res)
    vr = 0 if vec is None else (vec if k_res is None else GatherRows.apply(vec, k_res))
    if node_kernels_ok(x) and torch.is_tensor(dx) and dx.size(0) == xr.size(0):
        return Residual.apply(xr, dx, (None if vec is None else vr), dv, known, 1 / math.sqrt(2.0))
    x1 = (xr + dx) * (1 / math.sqrt(2.0)) * known[:, None]
    vec1 = (vr + dv) * known[:, None, None]
    return x1, vec1
