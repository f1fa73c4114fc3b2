"""Autograd wrappers around the C-ABI kernels (host side of the operator boundary,
SURVEY.md section 8(b)).  Tensors are allocated by PyTorch; the library only enqueues
kernels on the current stream.  CUDA(HIP) tensors only -- there is no CPU path."""
import ctypes

import torch

from . import _lib


class KernelTimer(object):
    """Optional per-launch timing with HIP events recorded on the launch stream (used by
    bench.py for the roofline figure).  Nothing is synchronised here; call `summary()` after
    the caller's own device synchronisation."""

    def __init__(self, prefix="message_scatter"):
        self.pairs = {}
        # only launches whose name starts with one of these are bracketed: an event pair costs ~15 us of host time
        self.prefix = (prefix,) if isinstance(prefix, str) else tuple(prefix)

    def launch(self, name, fn):
        if not name.startswith(self.prefix):
            return fn()
        a = torch.cuda.Event(enable_timing=True)
        b = torch.cuda.Event(enable_timing=True)
        a.record()
        rc = fn()
        b.record()
        self.pairs.setdefault(name, []).append((a, b))
        return rc

    def summary(self):
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v) / len(v)) for k, v in self.pairs.items()}


_TIMER = None


def set_kernel_timer(timer):
    global _TIMER
    _TIMER = timer


def _launch(name, fn):
    return fn() if _TIMER is None else _TIMER.launch(name, fn)


def _stream():
    """hipStream_t of PyTorch's current stream as an integer (the raw getter: no Stream object per launch)."""
    return torch._C._cuda_getCurrentRawStream(torch.cuda.current_device())


def _require_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError("hermnet_amd.%s: the hot path runs on MI355X only (got a %s tensor); "
                           "there is no CPU fallback" % (what, t.device))


class RbfDescriptor(object):
    """Host mirror of `hn_rbf_desc` (Gaussian basis * envelope, rmnet.py:156-193)."""

    def __init__(self, offset, rc, env_kind, env_p):
        self.offset = offset
        self.num_rbf = int(offset.numel())
        self.inv_rc = 1.0 / rc
        # GaussianSmearing: coeff = -0.5 / (offset[1] - offset[0]).item() ** 2
        self.coeff = None
        self.env_kind = env_kind
        self.env_p = env_p

    def struct(self):
        if self.coeff is None:
            o = self.offset.detach().float().cpu()
            self.coeff = -0.5 / float(o[1] - o[0]) ** 2
        return _lib.RbfDesc(self.offset.data_ptr(), self.num_rbf, self.inv_rc, self.coeff,
                            self.env_kind, self.env_p)


def _split_t(graph):
    """`split_t` of hermnet_message_scatter_bwd (one relation per workgroup in the 16-lanes-per-edge backward): measured slower
    on balanced compositions in round 2 and never chosen since; the host code always passes 0."""
    return 0


class EdgeGeometry(torch.autograd.Function):
    """`HVNet.with_edge` (hermnet.py:133-152) -> edge[E,4] = (rhat, d) in CSR order."""

    @staticmethod
    def forward(ctx, pos, cell, graph):
        _require_gpu(pos, "EdgeGeometry")
        lib = _lib.load()
        pos_c = pos.detach().float().contiguous()
        edge = torch.empty(graph.E, 4, dtype=torch.float32, device=pos.device)
        cell_c = None
        if graph.shift is not None and cell is not None:
            cell_c = cell.detach().float().reshape(-1, 3, 3).contiguous()
        batch32 = graph.batch32
        _lib.check(lib.hermnet_edge_geometry_fwd(
            _lib.ptr(pos_c), _lib.ptr(graph.src_id), _lib.ptr(graph.tgt_id),
            _lib.ptr(graph.shift if cell_c is not None else None), _lib.ptr(cell_c), _lib.ptr(batch32),
            graph.E, _lib.ptr(edge), _stream()), "hermnet_edge_geometry_fwd")
        ctx.graph = graph
        ctx.keep = (pos_c, cell_c)
        ctx.cell_shape = None if cell is None else tuple(cell.shape)
        return edge

    @staticmethod
    def backward(ctx, gedge):
        graph = ctx.graph
        lib = _lib.load()
        gD = gedge.float().contiguous()
        gpos_rows = torch.empty(graph.num_src or graph.N, 3, dtype=torch.float32, device=gD.device)
        if graph.num_src:
            # HTNet graph (virtual target rows): ordered segment sums, no atomics.  dE/dpos[i] = sum over the
            # edges leaving i of gD - sum over the edges entering i of gD.
            P_, B_, Te, TR = graph.triadic_pairs, graph.block, graph.T // graph.triadic_pairs, graph.T
            g3 = gD[:, :3].contiguous()
            rp_t, rp_s = graph.csr_rowptr.long(), graph.csc_rowptr.long()
            into = torch.segment_reduce(g3, "sum", lengths=rp_t[1:] - rp_t[:-1], unsafe=True)            # [Nt,3]
            outof = torch.segment_reduce(g3.index_select(0, graph.csc_pos.long()), "sum",
                                         lengths=rp_s[1:] - rp_s[:-1], unsafe=True)                      # [TR*Ns,3]
            gpos_rows = outof.view(TR, graph.num_src, 3).sum(0)
            gpos_rows[:Te * B_] -= into.view(Te, P_, B_, 3).sum(1).reshape(Te * B_, 3)
        elif graph.out_rowptr is None:      # device-built graphs: out-edges from the CSC order (one sort fewer)
            _lib.check(lib.hermnet_edge_geometry_bwd_csc(
                _lib.ptr(gD), _lib.ptr(graph.csr_rowptr), _lib.ptr(graph.csc_rowptr), _lib.ptr(graph.csc_pos),
                graph.T, graph.N, _lib.ptr(gpos_rows), _stream()), "hermnet_edge_geometry_bwd_csc")
        else:
            _lib.check(lib.hermnet_edge_geometry_bwd(
                _lib.ptr(gD), _lib.ptr(graph.csr_rowptr), None, _lib.ptr(graph.out_rowptr),
                _lib.ptr(graph.out_edges), graph.N, _lib.ptr(gpos_rows), _stream()), "hermnet_edge_geometry_bwd")
        gcell = None
        if ctx.needs_input_grad[1] and ctx.keep[1] is not None:
            # D = ... + shift @ cell[batch[src]]  =>  dE/dcell[b] = sum_{e in b} shift_e (x) gD_e
            # (what `virial_calc`, utils.py:153-155, differentiates for NPT runs)
            outer = graph.shift[:, :, None] * gD[:, None, :3]
            nb = ctx.keep[1].size(0)
            if nb == 1:
                gcell = outer.sum(0, keepdim=True)
            else:
                b = graph.batch32.long()[graph.src_id.long()]
                gcell = torch.zeros(nb, 3, 3, dtype=gD.dtype, device=gD.device).index_add_(0, b, outer)
            gcell = gcell.reshape(ctx.cell_shape)
        return gpos_rows.index_select(0, graph.row_of_node), gcell, None


class TrueEdgeGradient(torch.autograd.Function):
    """Identity on edge = (rhat, d) whose backward turns the TRUE gradient (dE/drhat, dE/dd), as produced
    by PyTorch autograd on the generic (optional radial basis) path, into the Cartesian dE/dD that
    `EdgeGeometry.backward` consumes:  gD = gd rhat + (gr - (gr.rhat) rhat) / d."""

    @staticmethod
    def forward(ctx, edge):
        ctx.save_for_backward(edge)
        return edge.clone()

    @staticmethod
    def backward(ctx, g):
        (edge,) = ctx.saved_tensors
        rh, d = edge[:, :3], edge[:, 3:4]
        gr, gd = g[:, :3], g[:, 3:4]
        gD = gd * rh + (gr - (gr * rh).sum(1, keepdim=True) * rh) / d
        return torch.cat([gD, torch.zeros_like(d)], dim=1)


class MessageScatter(torch.autograd.Function):
    """rbf_proj + propagate + residual of one HeteroVertexConv layer, all relations
    (rmnet.py:24-26, 55-73; utils.py:11-24).  Returns (x1, vec1).

    The `edge` input carries (rhat, d); the gradient returned for it is the Cartesian
    gradient w.r.t. the edge vector D (what `EdgeGeometry.backward` consumes).
    First-order only: gradients w.r.t. rbf_proj weights are not produced (force path)."""

    @staticmethod
    def forward(ctx, xh, vec, x, edge, wt, brbf, graph, rbf):
        _require_gpu(x, "MessageScatter")
        lib = _lib.load()
        H = x.size(1)
        xh = xh.contiguous()
        x = x.contiguous()
        vec_c = None if vec is None else vec.contiguous()
        x1 = torch.empty_like(x)
        vec1 = torch.empty(x.size(0), 3, H, dtype=x.dtype, device=x.device)
        gs, rs = graph.as_struct(), rbf.struct()
        _lib.check(_launch("message_scatter_fwd" + ("" if vec_c is not None else "_l0"), lambda: lib.hermnet_message_scatter_fwd(
            ctypes.byref(gs), ctypes.byref(rs), H, _lib.ptr(xh), None, _lib.ptr(vec_c), _lib.ptr(x),
            _lib.ptr(wt), _lib.ptr(brbf), _lib.ptr(edge), _lib.ptr(x1), _lib.ptr(vec1), None, 1, 0, _stream())),
            "hermnet_message_scatter_fwd")
        ctx.save_for_backward(xh, vec_c, edge, wt, brbf)
        ctx.graph, ctx.rbf, ctx.H = graph, rbf, H
        return x1, vec1

    @staticmethod
    def backward(ctx, gx1, gvec1):
        xh, vec, edge, wt, brbf = ctx.saved_tensors
        graph, rbf, H = ctx.graph, ctx.rbf, ctx.H
        lib = _lib.load()
        gx1 = gx1.contiguous()
        gvec1 = gvec1.contiguous()
        gxh = torch.empty_like(xh)
        split = _split_t(graph)
        gvec = None if vec is None else (torch.empty((graph.T,) + tuple(vec.shape), dtype=vec.dtype, device=vec.device)
                                         if split else torch.empty_like(vec))
        gx = torch.empty_like(gx1)
        gedge = torch.zeros(H // 64, graph.E, 4, dtype=torch.float32, device=gx1.device)
        part = None
        if graph.edge_table is not None and vec is not None and graph.T > 1 and not split:
            part = torch.empty((graph.T,) + tuple(vec.shape), dtype=vec.dtype, device=vec.device)
        gs, rs = graph.as_struct(), rbf.struct()
        _lib.check(_launch("message_scatter_bwd" + ("" if vec is not None else "_l0"), lambda: lib.hermnet_message_scatter_bwd(
            ctypes.byref(gs), ctypes.byref(rs), H, _lib.ptr(xh), None, _lib.ptr(vec), _lib.ptr(wt), _lib.ptr(brbf),
            _lib.ptr(edge), _lib.ptr(gx1), _lib.ptr(gvec1), _lib.ptr(gxh), _lib.ptr(gvec), _lib.ptr(gx),
            _lib.ptr(gedge), split, _lib.ptr(graph.edge_table), _lib.ptr(part), None, None, 0, _stream())), "hermnet_message_scatter_bwd")
        if split and gvec is not None:
            gvec = gvec.sum(0)
        return gxh, gvec, gx, gedge.sum(0), None, None, None, None


def edge_radial_table(graph, rbf, edge):
    """[E+1,32] per-edge radial record (window start, 12 tap pairs, unit vector) in CSC order -- the order the
    channel-per-lane backward kernel walks --; ONE launch per step: geometry and radial basis are the same for every
    layer (`include/hermnet_hip.h`: hermnet_edge_radial_table)."""
    E = edge.size(0)
    table = torch.empty(E + 1, 32, dtype=torch.float32, device=edge.device)      # (+1: the stream reads one record ahead)
    gs, rs = graph.as_struct(), rbf.struct()
    _lib.check(_launch("edge_radial_table", lambda: _lib.load().hermnet_edge_radial_table(
        ctypes.byref(gs), ctypes.byref(rs), _lib.ptr(edge), _lib.ptr(table), _stream())), "hermnet_edge_radial_table")
    return table
