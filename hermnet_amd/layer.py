"""One HeteroVertexConv layer (`HermNet/hermnet.py:37-65` + `rmnet.py:21-32`) as a single
autograd node with a hand-written first-order backward.

The launch sequence is fixed and short: LayerNorm, 2 GEMMs, the fused message kernel, and per
relation three GEMMs joined by fused elementwise kernels (`csrc/node_kernels.hip`).  Parameters
are treated as constants (energy/force evaluation; parameter gradients are not produced --
the training step runs through the differentiable device-op path, `HVNet.forward` in train() mode).
"""
import ctypes
import os as _os

import torch

from . import _lib, nodeops, switches
from .ops import _launch, _split_t, _stream

P = _lib.ptr

# switches.node_chain = False: node-level linears through library GEMMs joined by the stage kernels (the only path for widths
# the chain kernels are not instantiated for) instead of the four chain kernels of csrc/node_chain.hip.
def _node_chain_enabled():
    return switches.node_chain


class LayerWeights(object):
    """Kernel-ready views of one HeteroVertexConv's parameters, rebuilt when they change."""

    def __init__(self, mods):
        self.mods = list(mods)
        self.key = None
        self.builds = 0               # how often the copies were (re)built (guard.ParamGuard stamps its fingerprints with it)
        # (owner module, name) of every parameter, collected once: walking `module.parameters()` on every
        # forward costs ~0.13 ms of host time per layer
        self._slots = [(sub, name) for m in self.mods for sub in m.modules() for name in sub._parameters]

    def _version_key(self):
        # identity + version of the CURRENT parameter objects (a replaced Parameter or an in-place update both
        # change the key); data_ptr covers `.to(device)` / `.data = ...`
        key = []
        for sub, name in self._slots:
            p = sub._parameters[name]
            if p is not None:               # e.g. vec_proj has bias=False
                key.append((id(p), p._version, p.data_ptr()))
        return tuple(key)

    @staticmethod
    def _pad(w, dim, parts, H, Hp, scale=None):
        """Zero-pad `parts` consecutive blocks of H entries along `dim` to Hp entries each; `scale[k]` multiplies
        block k first."""
        if scale is not None:
            w = w.clone()
            for k, f in enumerate(scale):
                if f != 1.0:
                    w.narrow(dim, k * H, H).mul_(f)
        if Hp == H:
            return w
        shp = list(w.shape)
        shp[dim] = parts * Hp
        out = w.new_zeros(shp)
        for k in range(parts):
            out.narrow(dim, k * Hp, H).copy_(w.narrow(dim, k * H, H))
        return out

    @torch.no_grad()
    def refresh(self):
        """Widths that are not a multiple of 64 (the reference accepts any, hermnet.py:84-88): every channel axis is
        zero-padded to Hp = next multiple of 64, the column-block width of the message kernels.  Padded channels are
        exactly zero everywhere (zero weight rows/columns and biases; LayerNorm statistics over the real H,
        `h_real`), and the kernels' 1/sqrt(Hp) factors are turned back into 1/sqrt(H) by scaling the weights that
        feed them linearly by s = sqrt(Hp/H): the `a` and `b` thirds of x_proj[2] (message, rmnet.py:63-66) and the
        middle third of xvec_proj[2] (the factor of `vdot`, rmnet.py:97,104)."""
        key = self._version_key()
        if key == self.key:
            return self
        ml = [m.message_layer for m in self.mods]
        ul = [m.update_layer for m in self.mods]
        H = ml[0].x_proj[0].weight.size(0)
        Hp = (H + 63) // 64 * 64
        s = (Hp / H) ** 0.5
        self.h_real = H if Hp != H else 0
        pad = lambda w, dim, parts, scale=None: self._pad(w, dim, parts, H, Hp, scale)
        # LayerNorm affine folded into the first Linear: (n*g + b) W1^T + b1 = n (W1*g)^T + (W1 b + b1)
        w1 = [pad(pad(m.x_proj[0].weight * m.x_layernorm.weight[None, :], 0, 1), 1, 1) for m in ml]
        b1 = [pad(m.x_proj[0].weight @ m.x_layernorm.bias + m.x_proj[0].bias, 0, 1) for m in ml]
        self.w1cat = torch.cat(w1, 0).contiguous()                                  # [T*H, H]
        self.b1cat = torch.cat(b1, 0).contiguous()                                  # [T*H]
        self.w1cat_t = self.w1cat.t().contiguous()                                  # [H, T*H]
        sab = None if Hp == H else (1.0, s, s)
        self.w2 = torch.stack([pad(pad(m.x_proj[2].weight, 0, 3, sab), 1, 1) for m in ml], 0).contiguous()   # [T, 3H, H]
        self.w2t = self.w2.transpose(1, 2).contiguous()                             # [T, H, 3H]
        self.b2 = torch.stack([pad(m.x_proj[2].bias, 0, 3, sab) for m in ml], 0)[:, None, :].contiguous()     # [T,1,3H]
        self.wt = torch.stack([pad(m.rbf_proj.weight.t(), 1, 3) for m in ml], 0).contiguous()  # [T, R, 3H]
        self.brbf = torch.stack([pad(m.rbf_proj.bias, 0, 3) for m in ml], 0).contiguous()      # [T, 3H]
        self.wv = [pad(pad(u.vec_proj.weight, 0, 2), 1, 1).contiguous() for u in ul]           # [2H, H]
        self.wvt = [w.t().contiguous() for w in self.wv]                            # [H, 2H]
        self.wx0 = [pad(pad(u.xvec_proj[0].weight, 0, 1), 1, 2).contiguous() for u in ul]      # [H, 2H]
        self.wx0t = [w.t().contiguous() for w in self.wx0]
        self.bx0 = [pad(u.xvec_proj[0].bias, 0, 1) for u in ul]
        sq = None if Hp == H else (1.0, s, 1.0)
        self.wx2 = [pad(pad(u.xvec_proj[2].weight, 0, 3, sq), 1, 1).contiguous() for u in ul]  # [3H, H]
        self.wx2t = [w.t().contiguous() for w in self.wx2]
        self.bx2 = [pad(u.xvec_proj[2].bias, 0, 3, sq) for u in ul]
        # stacked copies for the uniform-block layout (one batched GEMM per stage)
        st = lambda lst: torch.stack(lst, 0).contiguous()
        self.wv_s, self.wvt_s = st(self.wv), st(self.wvt)
        self.wx0_s, self.wx0t_s, self.bx0_s = st(self.wx0), st(self.wx0t), st(self.bx0)[:, None, :].contiguous()
        self.wx2_s, self.wx2t_s, self.bx2_s = st(self.wx2), st(self.wx2t), st(self.bx2)[:, None, :].contiguous()
        # MFMA-operand-order copies for the chain kernels (include/hermnet_hip.h: frag(W))
        self.chain = nodeops.chain_supported(Hp)
        if self.chain:
            fr = nodeops.weight_fragments
            w1s = self.w1cat.view(len(ml), Hp, Hp)
            self.w1f, self.w1tf = fr(w1s), fr(w1s.transpose(1, 2).contiguous())
            self.w2f, self.w2tf = fr(self.w2), fr(self.w2t)
            self.wvf, self.wvtf = fr(self.wv_s), fr(self.wvt_s)
            self.wx0f, self.wx0tf = fr(self.wx0_s), fr(self.wx0t_s)
            self.wx2f, self.wx2tf = fr(self.wx2_s), fr(self.wx2t_s)
            self.wvf16 = self.w1f16 = None
            if Hp == 128:        # the update chain's 16-row form (csrc/node_chain16.hip) reads frag16 copies
                f16 = nodeops.weight_fragments16
                self.wvf16, self.wvtf16 = f16(self.wv_s), f16(self.wvt_s)
                self.wx0f16, self.wx0tf16 = f16(self.wx0_s), f16(self.wx0t_s)
                self.wx2f16, self.wx2tf16 = f16(self.wx2_s), f16(self.wx2t_s)
                # the node projection on 16-row tiles (the fused layer boundary, csrc/node_chain16.hip)
                self.w1f16, self.w1tf16 = f16(w1s), f16(w1s.transpose(1, 2).contiguous())
                self.w2f16, self.w2tf16 = f16(self.w2), f16(self.w2t)
        self.key = key
        self.builds += 1
        return self


def _msg_fwd(graph, rbf, H, xh, vec, x, w, edge, xh_bias=True, ranges=None, zero_unknown=True, out=None, range_rows=0):
    """`xh_bias=False`: xh already includes x_proj's bias (chain kernels).  `ranges` [T,2] int32 (device): only these
    target rows of every relation (atom shards: two launches over complementary ranges around the halo exchange;
    the second passes the first one's result as `out`); `range_rows`: how many rows they cover (host int)."""
    lib = _lib.load()
    b2 = w.b2 if xh_bias else None
    if out is None:
        x1 = torch.empty(graph.N, H, dtype=x.dtype, device=x.device)          # target rows (= source rows unless HTNet)
        vec1 = torch.empty(graph.N, 3, H, dtype=x.dtype, device=x.device)
    else:
        x1, vec1 = out
    gs, rs = graph.as_struct(), rbf.struct()
    _lib.check(_launch("message_scatter_fwd" + ("" if vec is not None else "_l0"),
                       lambda: lib.hermnet_message_scatter_fwd(
                           ctypes.byref(gs), ctypes.byref(rs), H, P(xh), P(b2), P(vec), P(x), P(w.wt), P(w.brbf), P(edge),
                           P(x1), P(vec1), P(ranges), 1 if zero_unknown else 0, int(range_rows), _stream())),
               "hermnet_message_scatter_fwd")
    return x1, vec1


def _bwd_sums_deferrable(graph, H):
    """The message backward can leave its finishing launch to the consumer (hermnet_message_scatter_bwd with gx = NULL):
    channel-per-lane form, targets = sources (HVNet rows), 32-bit offsets.  Width 128 only: there the update backward
    keeps two or three workgroups per CU and forms the sums half a wave per row (measured, same job: 2.98 vs 3.01 ms at
    configs[1], 27.2 vs 27.6 ms at 100k atoms); the wide kernels (one workgroup per CU) lose more in their prologue than
    the two launches cost (H = 512: 22.5 vs 20.8 ms), so they keep the launches."""
    return (H == 128 and graph.edge_table is not None and not graph.num_src and getattr(graph, "res_row", None) is None
            and not _split_t(graph) and _lib.get_option("bwd_lanes16") == 0
            and graph.N * 3 * H * 4 < 2 ** 32 and switches.defer_sums())


def _msg_bwd(graph, rbf, H, xh, vec, w, edge, gx1, gvec1, gedge, xh_bias=True, ranges=None, out=None, finish=True):
    """`gedge` [H/64, E, 4] (zero-filled by the caller) receives the per-column-block Cartesian edge gradients.
    `ranges` = (device [k,2] int32, host list of (lo, hi)): only these SOURCE rows (atom shards: the halo rows first,
    the others while their gradients travel; the second call passes the first one's buffers as `out`).
    `finish=False` (only where `_bwd_sums_deferrable`): no finishing launch -- returns (gxh, per-relation partial sums of
    gvec [T,N,3,H] or None); the sums over the relations and the residual's identity terms are the consumer's."""
    lib = _lib.load()
    b2 = w.b2 if xh_bias else None
    split = _split_t(graph)
    if not finish:
        if out is None:
            gxh = torch.empty_like(xh)
            part = None if vec is None else torch.empty((graph.T,) + tuple(vec.shape), dtype=vec.dtype, device=vec.device)
        else:
            gxh, part = out
        rd, rh, nr = None, None, 0
        if ranges is not None:         # (the "proj" halo exchange: the halo source rows first, the others while they travel)
            rd, host = ranges
            nr = len(host)
            rh = (ctypes.c_int * (2 * nr))(*[v for lo_hi in host for v in lo_hi])
        gs, rs = graph.as_struct(), rbf.struct()
        _lib.check(_launch("message_scatter_bwd" + ("" if vec is not None else "_l0"),
                           lambda: lib.hermnet_message_scatter_bwd(
                               ctypes.byref(gs), ctypes.byref(rs), H, P(xh), P(b2), P(vec), P(w.wt), P(w.brbf), P(edge),
                               P(gx1), P(gvec1), P(gxh), None, None, P(gedge), 0, P(graph.edge_table), P(part),
                               P(rd), rh, nr, _stream())),
                   "hermnet_message_scatter_bwd")
        return gxh, part
    if out is None:
        gxh = torch.empty_like(xh)
        gvec = None if vec is None else (torch.empty((graph.T,) + tuple(vec.shape), dtype=vec.dtype, device=vec.device)
                                         if split else torch.empty_like(vec))
        gx = torch.empty(xh.size(1), H, dtype=gx1.dtype, device=gx1.device)   # source rows
        # workspace of the channel-per-lane form: per-relation partial sums of gvec
        part = None
        if graph.edge_table is not None and vec is not None and graph.T > 1 and not split:
            part = torch.empty((graph.T,) + tuple(vec.shape), dtype=vec.dtype, device=vec.device)
    else:
        gxh, gvec, gx, part = out
    rd, rh, nr = None, None, 0
    if ranges is not None:
        rd, host = ranges
        nr = len(host)
        rh = (ctypes.c_int * (2 * nr))(*[v for lo_hi in host for v in lo_hi])
    gs, rs = graph.as_struct(), rbf.struct()
    _lib.check(_launch("message_scatter_bwd" + ("" if vec is not None else "_l0"),
                       lambda: lib.hermnet_message_scatter_bwd(
                           ctypes.byref(gs), ctypes.byref(rs), H, P(xh), P(b2), P(vec), P(w.wt), P(w.brbf), P(edge),
                           P(gx1), P(gvec1), P(gxh), P(gvec), P(gx), P(gedge), split, P(graph.edge_table), P(part),
                           P(rd), rh, nr, _stream())),
               "hermnet_message_scatter_bwd")
    if ranges is not None:
        return gxh, gvec, gx, part
    if split and gvec is not None:
        gvec = gvec.sum(0)
    return gxh, gvec, gx


def _virtual_residual(graph, gx1, gvec1, gx_in, gvec_in, H, ranges=None):
    """HTNet: the residual (rmnet.py:24-26) reads the atom's own row from each of its P virtual target rows; its gradient
    returns as the sum over those rows (the message backward adds no identity term with virtual targets).  `ranges`
    (device [k,2], host list): only these source rows (the ranged launches of the halo overlap)."""
    if not graph.num_src:
        return
    P_, B_, Te = graph.triadic_pairs, graph.block, graph.T // graph.triadic_pairs
    if gx_in.is_cuda and gvec_in is not None:
        nodeops.pair_sum_accumulate(gx1, gvec1, gx_in, gvec_in, Te, P_, B_, 0.5 ** 0.5, 1.0, ranges=ranges)
        return
    sel = slice(0, Te * B_)
    if ranges is not None:
        sel = torch.zeros(Te * B_, dtype=torch.bool)
        for lo, hi in ranges[1]:
            sel[lo:min(hi, Te * B_)] = True
    gx_in[:Te * B_][sel] += (gx1.view(Te, P_, B_, H).sum(1).reshape(Te * B_, H) * (0.5 ** 0.5))[sel]
    if gvec_in is not None:
        gvec_in[:Te * B_][sel] += gvec1.view(Te, P_, B_, 3, H).sum(1).reshape(Te * B_, 3, H)[sel]


class EdgeGradSink(object):
    """One buffer [layers, H/64, E, 4] for the edge gradients of a whole step: every layer's backward kernel
    writes its slice, and `EdgeFanout.backward` reduces all slices in ONE pass (instead of a zero-fill, a
    block sum and an autograd accumulation per layer)."""

    def __init__(self, layers, nblk, E, device, zero=True):
        """`zero=False`: every edge has a target of a known element, so every slot is written by the kernels and the
        69 MB (config 2) zero-fill can be skipped."""
        self.shape = (layers, nblk, E, 4)
        self.device = device
        self.zero = zero
        self.buf = None

    def slice(self, li):
        if self.buf is None:     # edges to targets of an unknown element are never written: zero once if there are any
            alloc = torch.zeros if self.zero else torch.empty
            self.buf = alloc(self.shape, dtype=torch.float32, device=self.device)
            if not self.zero and switches.debug_poison():
                self.buf.fill_(float("nan"))         # (tests: a slot the kernels did not write would poison the forces)
        return self.buf[li]


class EdgeFanout(torch.autograd.Function):
    """edge [E,4] -> one handle per layer, an [H/64, E, 4] stride-0 view of it (so that a layer can return its
    per-column-block gradient slices unreduced); backward = sum over every slice of every layer."""

    @staticmethod
    def forward(ctx, edge, sink):
        ctx.sink = sink
        L, nblk = sink.shape[0], sink.shape[1]
        return tuple(edge.unsqueeze(0).expand(nblk, *edge.shape) for _ in range(L))

    @staticmethod
    def backward(ctx, *grads):
        sink = ctx.sink
        buf, ctx.sink.buf = sink.buf, None            # the buffer belongs to this backward pass only
        if buf is not None and all(g is not None and g.data_ptr() == buf[i].data_ptr() for i, g in enumerate(grads)):
            L, nblk, E, _ = sink.shape
            return buf.view(L * nblk, E, 4).sum(0), None
        live = [g.sum(0) for g in grads if g is not None]     # partial graphs: plain accumulation
        return (torch.stack(live, 0).sum(0) if live else None), None


# gradients handed down as partial sums, keyed by (graph, index of the consuming layer) (nodeops.PendingGrads);
# an entry lives from one layer's backward to the next one's (HVNet.forward clears leftovers of an interrupted pass)
_PENDING = {}
# the NEXT layer's node projection, computed by a layer's fused update launch: (graph, index of the next layer) ->
# (x_out, (hb, xh, mean, rstd)); HeteroVertexConv.forward hands it to that layer (`pre`)
_PRE_NEXT = {}


def _boundary_mode():
    """`switches.boundary_mode` -- how the node launches of a layer boundary are cut, where `nodeops.fused_boundary_supported`
    (width 128, 16-row update tiles, HVNet rows: csrc/node_chain16.hip):
      0 (default)  every phase a launch of its own: 64-row projection kernels + 16-row update kernels, the input gradients handed
                   down as partial sums (the round-4 form);
      4            the BACKWARD boundary as one launch: the projection's backward of layer l + 1 runs inside the update backward
                   of layer l (sums over the relations in registers, LayerNorm backward on the tile: no [T, N, H] partial sums in
                   memory);
      1            both boundaries as one launch each (the next layer's projection inside the update launch as well);
      3            only the forward boundary;
      2            the 16-row phases of mode 1 as launches of their own (the bit-for-bit check of the fused kernels).
    Measured in the model (configs[1], one box, three interleaved rounds).  With fp32 MFMAs (profiles/r05_boundary_ab.log):
    0: 2.944, 4: 2.945, 3: 2.971, 1: 2.988 ms per step.  Since the products run as bf16 splits (profiles/r05_boundary_ab_split.log):
    0: 2.80, 4: 2.82, 3: 2.87, 1: 2.89 -- with the matrix pipe 2.7 x cheaper a tile's time is its weight stream, and a 64-row
    projection tile streams a quarter of the bytes per row of a 16-row one: the fused forms lose what that gains."""
    return int(switches.boundary_mode)


class FusedRelationalLayer(torch.autograd.Function):
    """(x, vec, edge) -> (x_out, vec_out) for one layer, relation (row) order."""

    @staticmethod
    def forward(ctx, x, vec, edge, graph, rbf, w, sink=None, li=0, halo=None, defer=False, pre=None, w_next=None, proj=False):
        """`edge`: [E,4], or this layer's handle from `EdgeFanout` ([H/64,E,4] stride-0 view; same memory).
        x / vec live in SOURCE rows, the outputs in TARGET rows; the two coincide for HVNet and differ for HTNet
        (`graph.num_src`: one target row per atom and pair relation, relations.build_triadic).
        `halo` (atom shards, chain path only; `sharding.HaloOverlap`): the exchange of the halo rows of (x, vec) is
        still due -- it runs here, around the node projection: pack, start the all-to-all, project the row tiles
        that hold no halo row, wait, unpack IN PLACE, project the rest.  The backward mirrors it.
        `defer` (HeteroVertexConv.forward: x and vec are the outputs of the chain layer below and of nothing else): the
        backward hands its input gradients down as partial sums (`nodeops.PendingGrads`) and the update backward of the
        layer below forms them in its own launch -- two small launches per layer boundary less, same bits.
        `pre` (chain path without halo): the node projection of THIS x, already launched by the caller (HVNet.forward runs the
        first layer's beside the relation build; round 5: every later layer's comes out of the fused update launch of the
        layer below).
        `w_next` (round 5): the NEXT layer's weights -- its node projection of the rows this layer produces runs inside this
        layer's update launch where `nodeops.fused_boundary_supported` (result left in `_PRE_NEXT`).
        `proj` (round 6, with `halo`): x and vec come straight from the chain layer below (as for `defer`), so the exchange may
        take the "proj" form: projected rows travel forward, partial sums of gradients backward."""
        Ns, H = x.shape
        N = graph.N
        T = graph.T
        rp = graph.type_rowptr_host
        nk = rp[-1]
        x = x.contiguous()
        vec = None if vec is None else vec.contiguous()
        # --- node projection of every relation: xh[t] = x_proj_t(LayerNorm_t(x))  (rmnet.py:52)
        uni, B = graph.uniform and nk > 0, graph.block
        ctx.chain = w.chain and _node_chain_enabled()
        if ctx.chain:
            # three launches: node_pre_fwd (LayerNorm + x_proj of every relation), the message kernel, node_update_fwd
            # atom shards, the "proj" form of the exchange (round 6): the halo rows travel as what the message kernel gathers --
            # xh[t] of every relation and vec, 12H floats per atom -- so no node kernel runs a second time on the halo tiles
            # and the backward hands its gradients down as partial sums like the unsharded layer (`defer`)
            ctx.proj = bool(proj) and halo is not None and vec is not None and (
                _bwd_sums_deferrable(graph, H) or not (ctx.needs_input_grad[0] or ctx.needs_input_grad[2]))
            if halo is None:
                hb, xh, mean, rstd = pre if pre is not None else nodeops.node_pre_fwd(x, w, T, src_ranges=graph.src_ranges)
                x1, vec1 = _msg_fwd(graph, rbf, H, xh, vec, x, w, edge, xh_bias=False)
            elif ctx.proj:
                #   project every row (halo rows from stale inputs: replaced below) -> pack (xh | vec) of the rows the peers
                #   need -> start the all-to-all -> messages into the targets that read no halo row -> the STREAM waits ->
                #   unpack IN PLACE -> messages into the remaining targets
                from .sharding import _all_to_all_rows_start, comm_wait
                plan = halo.plan
                hb, xh, mean, rstd = nodeops.node_pre_fwd(x, w, T, src_ranges=graph.src_ranges)
                send = nodeops.halo_proj_rows(0, xh, vec, plan.send_idx)
                recv, work = _all_to_all_rows_start(send, plan.send_counts, plan.recv_counts, plan.group)
                if switches.debug_poison():
                    # (tests: nothing that runs before the unpack may depend on a halo row)
                    nodeops.halo_proj_rows(2, xh, vec, plan.recv_idx, torch.full_like(recv, float("nan")))
                x1, vec1 = out = _msg_fwd(graph, rbf, H, xh, vec, x, w, edge, xh_bias=False, ranges=halo.fwd_early,
                                          zero_unknown=True, range_rows=halo.early_rows)
                comm_wait(work, "fwd", sum(plan.send_counts), sum(plan.recv_counts))
                nodeops.halo_proj_rows(2, xh, vec, plan.recv_idx, recv)
                if halo.late_rows > 0:
                    _msg_fwd(graph, rbf, H, xh, vec, x, w, edge, xh_bias=False, ranges=halo.fwd_late, zero_unknown=False,
                             out=out, range_rows=halo.late_rows)
            else:
                # The exchange runs behind the node projection AND the message kernel of every target that reads no
                # halo row ("run interior edges while the halo is in flight"):
                #   pack -> start the all-to-all -> project every row (halo rows from stale inputs: redone below) ->
                #   messages into the early targets -> the STREAM waits -> unpack IN PLACE -> project the tiles that hold
                #   a halo row -> messages into the remaining targets.
                from .sharding import _all_to_all_rows_start
                plan = halo.plan
                send = nodeops.halo_rows(0, x, vec, plan.send_idx)
                recv, work = _all_to_all_rows_start(send, plan.send_counts, plan.recv_counts, plan.group)
                if switches.debug_poison():
                    # (tests: nothing that runs before the unpack may depend on a halo row)
                    nodeops.halo_rows(2, x, vec, plan.recv_idx, torch.full_like(recv, float("nan")))
                hb, xh, mean, rstd = pre = nodeops.node_pre_fwd(x, w, T, src_ranges=graph.src_ranges)
                x1, vec1 = out = _msg_fwd(graph, rbf, H, xh, vec, x, w, edge, xh_bias=False, ranges=halo.fwd_early,
                                          zero_unknown=True, range_rows=halo.early_rows)
                from .sharding import comm_wait
                comm_wait(work, "fwd", sum(plan.send_counts), sum(plan.recv_counts))
                if plan.recv_idx.numel() > 0:  # (a rank without halo atoms has nothing to redo)
                    nodeops.halo_rows(2, x, vec, plan.recv_idx, recv)
                    nodeops.node_pre_fwd(x, w, T, src_ranges=graph.src_ranges, windows=halo.windows, mode=1, out=pre)
                if halo.late_rows > 0:
                    _msg_fwd(graph, rbf, H, xh, vec, x, w, edge, xh_bias=False, ranges=halo.fwd_late, zero_unknown=False,
                             out=out, range_rows=halo.late_rows)
            ctx.halo = halo
            ctx.defer = (bool(defer) and halo is None and vec is not None) or ctx.proj
            mode = _boundary_mode()
            if (w_next is not None and halo is None and mode in (1, 2, 3)
                    and nodeops.fused_boundary_supported(graph, H, w, w_next)):
                if mode != 2:
                    x_out, vec_out, vp, h2b, q23, nrm, pre_next = nodeops.node_update_pre_fwd(x1, vec1, w, graph, w_next)
                else:
                    x_out, vec_out, vp, h2b, q23, nrm = nodeops.node_update_fwd(x1, vec1, w, graph)
                    pre_next = nodeops.node_pre_fwd16(x_out, w_next, T)
                _PRE_NEXT[(id(graph), li + 1)] = (x_out, pre_next)
            else:
                x_out, vec_out, vp, h2b, q23, nrm = nodeops.node_update_fwd(x1, vec1, w, graph)
            ctx.save_for_backward(x, mean, rstd, hb, xh, vec, edge, vp, h2b, q23, nrm)
            ctx.graph, ctx.rbf, ctx.w, ctx.sink, ctx.li = graph, rbf, w, sink, li
            return x_out, vec_out
        if halo is not None:
            raise RuntimeError("the in-layer halo exchange belongs to the chain path (HeteroVertexConv.forward decides)")
        n, mean, rstd = nodeops.layernorm_fwd(x, 1e-5, h_real=w.h_real)
        h = _launch("gemm", lambda: torch.addmm(w.b1cat, n, w.w1cat.t()))                                     # [Ns, T*H]
        a = nodeops.ssilu_fwd(h)
        # (biases that would be broadcast over a batched GEMM's rows are added by the consuming kernel
        # instead: baddbmm with a broadcast bias first copies it over the whole output)
        xh = _launch("gemm", lambda: torch.bmm(a.view(Ns, T, H).transpose(0, 1), w.w2t))                      # [T, Ns, 3H], + b2 on load
        # --- fused edge part + residual (rmnet.py:55-73, 24-26)
        x1, vec1 = _msg_fwd(graph, rbf, H, xh, vec, x, w, edge)
        # --- PaiNNUpdate on the rows of each relation (rmnet.py:94-107)
        vp = torch.empty(N, 3, 2 * H, dtype=x.dtype, device=x.device)
        h2 = torch.empty(N, H, dtype=x.dtype, device=x.device)
        q = torch.empty(N, 3 * H, dtype=x.dtype, device=x.device)
        if uni:   # every relation owns `B` rows: one batched GEMM per stage
            _launch("gemm", lambda: torch.bmm(vec1[:nk].view(T, 3 * B, H), w.wvt_s, out=vp[:nk].view(T, 3 * B, 2 * H)))
        else:
            for t in range(T):
                lo, hi = rp[t], rp[t + 1]
                if hi > lo:
                    _launch("gemm", lambda: torch.mm(vec1[lo:hi].view(-1, H), w.wvt[t], out=vp[lo:hi].view(-1, 2 * H)))
        vdot, xin = nodeops.update_mid(vp, x1, nk, H)
        if uni:
            _launch("gemm", lambda: torch.bmm(xin[:nk].view(T, B, 2 * H), w.wx0t_s, out=h2[:nk].view(T, B, H)))
        else:
            for t in range(T):
                lo, hi = rp[t], rp[t + 1]
                if hi > lo:
                    _launch("gemm", lambda: torch.addmm(w.bx0[t], xin[lo:hi], w.wx0t[t], out=h2[lo:hi]))
        kb = dict(bias=w.bx0_s, rows_per_bias=B) if uni else {}
        a2 = nodeops.ssilu_fwd(h2[:nk], **kb) if nk > 0 else h2[:0]
        if uni:
            _launch("gemm", lambda: torch.bmm(a2.view(T, B, H), w.wx2t_s, out=q[:nk].view(T, B, 3 * H)))
        else:
            for t in range(T):
                lo, hi = rp[t], rp[t + 1]
                if hi > lo:
                    _launch("gemm", lambda: torch.addmm(w.bx2[t], a2[lo:hi], w.wx2t[t], out=q[lo:hi]))
        qb = dict(qbias=w.bx2_s, rows_per_bias=B) if uni else {}
        x_out, vec_out = nodeops.update_out(q, vdot, vp, x1, vec1, graph.row_active, N, nk, H, **qb)
        ctx.save_for_backward(x, mean, rstd, h, xh, vec, edge, vp, vdot, xin, h2, q)
        ctx.graph, ctx.rbf, ctx.w, ctx.sink, ctx.li = graph, rbf, w, sink, li
        return x_out, vec_out

    @staticmethod
    def backward(ctx, gxo, gvo):
        graph, rbf, w = ctx.graph, ctx.rbf, ctx.w
        N = graph.N
        T = graph.T
        rp = graph.type_rowptr_host
        nk = rp[-1]
        gxo = gxo.contiguous()
        gvo = gvo.contiguous()
        uni, B = graph.uniform and nk > 0, graph.block
        if ctx.chain:
            x, mean, rstd, hb, xh, vec, edge, vp, h2b, q23, nrm = ctx.saved_tensors
            Ns, H = x.shape
            # (the layer above may have left its finishing launches to this one: the buffers arrive unfilled -- and they must be
            # the very buffers it registered: a copy made on the way would hold garbage, so that is refused loudly)
            pend = _PENDING.pop((id(graph), ctx.li), None)
            if pend is not None and (pend.gx.data_ptr() != gxo.data_ptr() or pend.gvec.data_ptr() != gvo.data_ptr()):
                raise RuntimeError("hermnet_amd: the gradients handed down as partial sums (layer %d) did not arrive in the "
                                   "buffers they were registered with; set the environment variable named in "
                                   "hermnet_amd/switches.py: defer_sums to 0" % (ctx.li + 1))
            gx1, gvec1 = nodeops.node_update_bwd(gxo, gvo, vp, h2b, q23, nrm, w, graph, pending=pend)
        else:
            x, mean, rstd, h, xh, vec, edge, vp, vdot, xin, h2, q = ctx.saved_tensors
            Ns, H = x.shape
            qb = dict(qbias=w.bx2_s, rows_per_bias=B) if uni else {}
            kb = dict(bias=w.bx0_s, rows_per_bias=B) if uni else {}
            gq, gvdot, gvp, gx1, gvec1 = nodeops.update_out_bwd(gxo, gvo, q, vdot, vp, graph.row_active, N, nk, H, **qb)
            gxin = torch.empty(N, 2 * H, dtype=x.dtype, device=x.device)
            ga2 = torch.empty(N, H, dtype=x.dtype, device=x.device)
            if uni:
                _launch("gemm", lambda: torch.bmm(gq[:nk].view(T, B, 3 * H), w.wx2_s, out=ga2[:nk].view(T, B, H)))
            else:
                for t in range(T):
                    lo, hi = rp[t], rp[t + 1]
                    if hi > lo:
                        _launch("gemm", lambda: torch.mm(gq[lo:hi], w.wx2[t], out=ga2[lo:hi]))
            gh2 = nodeops.ssilu_bwd(ga2, h2, nk, 1, H, H, H, **kb) if nk > 0 else ga2[:0]
            if uni:
                _launch("gemm", lambda: torch.bmm(gh2.view(T, B, H), w.wx0_s, out=gxin[:nk].view(T, B, 2 * H)))
            else:
                for t in range(T):
                    lo, hi = rp[t], rp[t + 1]
                    if hi > lo:
                        _launch("gemm", lambda: torch.mm(gh2[lo:hi], w.wx0[t], out=gxin[lo:hi]))
            nodeops.update_mid_bwd(gvdot, gxin, vp, xin, gvp, gx1, nk, H)
            if uni:
                gv = gvec1[:nk].view(T, 3 * B, H)
                _launch("gemm", lambda: torch.baddbmm(gv, gvp[:nk].view(T, 3 * B, 2 * H), w.wv_s, out=gv))
            else:
                for t in range(T):
                    lo, hi = rp[t], rp[t + 1]
                    if hi > lo:
                        g = gvec1[lo:hi].view(-1, H)
                        _launch("gemm", lambda: torch.addmm(g, gvp[lo:hi].view(-1, 2 * H), w.wv[t], out=g))
        fan = edge.dim() == 3          # handle from EdgeFanout: return the slices unreduced
        if fan and ctx.sink is not None:
            gedge = ctx.sink.slice(ctx.li)
        else:
            gedge = torch.zeros(H // 64, graph.E, 4, dtype=torch.float32, device=gx1.device)
        halo = ctx.halo if ctx.chain else None
        if halo is not None and ctx.proj:
            # The gradients of the halo SOURCE rows first, as they stand behind the message backward -- gxh[t] of every relation
            # and the per-relation partial sums of gvec, summed while they are packed; cleared here: the local halo rows were
            # overwritten in the forward --; they travel to their owners while the other source rows are computed; the owners add
            # them to their own gxh / partial sums in list order, and ONE node_pre_bwd over every row follows.  The input
            # gradients go down as partial sums (`defer`): no finishing launch, no windowed node launch, no LayerNorm backward.
            from .sharding import _all_to_all_rows_start, comm_wait
            plan = halo.plan
            gxh = torch.empty_like(xh)
            gv_parts = torch.empty((T,) + tuple(vec.shape), dtype=vec.dtype, device=vec.device)
            bufs = (gxh, gv_parts)
            if halo.bwd_first_rows[1]:
                _msg_bwd(graph, rbf, H, xh, vec, w, edge, gx1, gvec1, gedge, xh_bias=False, ranges=halo.bwd_first_rows,
                         out=bufs, finish=False)
            gsend = nodeops.halo_proj_rows(1, gxh, gv_parts, plan.recv_idx)           # pack and clear: none stays here
            back, work = _all_to_all_rows_start(gsend, plan.recv_counts, plan.send_counts, plan.group)
            if halo.bwd_rest_rows[1]:
                _msg_bwd(graph, rbf, H, xh, vec, w, edge, gx1, gvec1, gedge, xh_bias=False, ranges=halo.bwd_rest_rows,
                         out=bufs, finish=False)
            comm_wait(work, "bwd", sum(plan.recv_counts), sum(plan.send_counts))
            nodeops.halo_proj_accumulate(gxh, gv_parts, plan, back)                 # gradients of my atoms used elsewhere
            gn_parts = nodeops.node_pre_bwd(gxh, hb, x, mean, rstd, w, src_ranges=graph.src_ranges, parts_only=True)
            gx_total, gvec_in = torch.empty_like(x), torch.empty_like(vec)          # filled by the layer below
            if switches.debug_poison():                # (tests: nothing reads them before that)
                gx_total.fill_(float("nan"))
                gvec_in.fill_(float("nan"))
            pend = _PENDING[(id(graph), ctx.li - 1)] = nodeops.PendingGrads(
                gx_total, gvec_in, gn_parts, gv_parts, x, mean, rstd, gx1, gvec1, w.h_real)
            pend.w_above = w
            ge = gedge if fan else (gedge[0] if gedge.size(0) == 1 else gedge.sum(0))
            return (gx_total, gvec_in, ge) + (None,) * 10
        if halo is not None:
            # the gradients of the halo rows first: they travel to their owners while the other rows are computed
            from .sharding import _all_to_all_rows_start
            plan = halo.plan
            bufs = None
            if halo.bwd_first[1]:
                bufs = _msg_bwd(graph, rbf, H, xh, vec, w, edge, gx1, gvec1, gedge, xh_bias=False, ranges=halo.bwd_first)
                gxh, gvec_in, gx_in, _ = bufs
                _virtual_residual(graph, gx1, gvec1, gx_in, gvec_in, H, halo.bwd_first)
                out = nodeops.node_pre_bwd(gxh, hb, x, mean, rstd, w, add=gx_in, src_ranges=graph.src_ranges,
                                           windows=halo.windows, mode=1)
                gsend = nodeops.halo_rows(1, out[0], gvec_in, plan.recv_idx)           # pack and clear: none stays here
            else:                                                                   # (a rank without halo atoms)
                out = None
                gsend = x.new_empty(0, 4 * H)
            back, work = _all_to_all_rows_start(gsend, plan.recv_counts, plan.send_counts, plan.group)
            if halo.bwd_rest[1]:
                bufs = _msg_bwd(graph, rbf, H, xh, vec, w, edge, gx1, gvec1, gedge, xh_bias=False, ranges=halo.bwd_rest,
                                out=bufs)
                _virtual_residual(graph, gx1, gvec1, bufs[2], bufs[1], H, halo.bwd_rest)
            gxh, gvec_in, gx_in, _ = bufs
            res = nodeops.node_pre_bwd(gxh, hb, x, mean, rstd, w, add=gx_in, src_ranges=graph.src_ranges,
                                       windows=halo.windows, mode=2, out=out)
            gx_total = res[0]
            from .sharding import comm_wait
            comm_wait(work, "bwd", sum(plan.recv_counts), sum(plan.send_counts))
            nodeops.halo_accumulate(gx_total, gvec_in, plan, back)                  # gradients of my atoms used elsewhere
            ge = gedge if fan else (gedge[0] if gedge.size(0) == 1 else gedge.sum(0))
            return (gx_total, gvec_in, ge) + (None,) * 10
        ge = gedge if fan else (gedge[0] if gedge.size(0) == 1 else gedge.sum(0))
        if ctx.chain and _bwd_sums_deferrable(graph, H):
            if ctx.defer and ctx.needs_input_grad[0]:
                gxh, gv_parts = _msg_bwd(graph, rbf, H, xh, vec, w, edge, gx1, gvec1, gedge, xh_bias=False, finish=False)
                mode, chain = _boundary_mode(), None
                if mode in (1, 2, 4) and nodeops.fused_boundary_supported(graph, H, w):
                    # round 5: this layer's projection backward runs inside the update backward of the layer below (1), or as
                    # the same 16-row phase in a launch of its own (2)
                    if mode != 2:
                        gn_parts, chain = None, (gxh, hb, w.w2tf16, w.w1tf16)
                    else:
                        gn_parts = nodeops.node_pre_bwd16(gxh, hb, w)
                else:
                    gn_parts = nodeops.node_pre_bwd(gxh, hb, x, mean, rstd, w, src_ranges=graph.src_ranges, parts_only=True)
                gx_total, gvec_in = torch.empty_like(x), torch.empty_like(vec)      # filled by the layer below
                if switches.debug_poison():            # (tests: nothing reads them before that)
                    gx_total.fill_(float("nan"))
                    gvec_in.fill_(float("nan"))
                pend = _PENDING[(id(graph), ctx.li - 1)] = nodeops.PendingGrads(
                    gx_total, gvec_in, gn_parts, gv_parts, x, mean, rstd, gx1, gvec1, w.h_real, chain=chain)
                pend.w_above = w            # (keeps the fragment copies alive; the CPU restatement of the tests reads it)
                return (gx_total, gvec_in, ge) + (None,) * 10
            if vec is None and not ctx.needs_input_grad[0]:     # the first layer: nothing below wants gx / gvec
                _msg_bwd(graph, rbf, H, xh, vec, w, edge, gx1, gvec1, gedge, xh_bias=False, finish=False)
                return (None, None, ge) + (None,) * 10
        gxh, gvec_in, gx_in = _msg_bwd(graph, rbf, H, xh, vec, w, edge, gx1, gvec1, gedge, xh_bias=not ctx.chain)
        _virtual_residual(graph, gx1, gvec1, gx_in, gvec_in, H)
        gx_total = None
        if ctx.needs_input_grad[0]:
            if ctx.chain:
                gx_total = nodeops.node_pre_bwd(gxh, hb, x, mean, rstd, w, add=gx_in, src_ranges=graph.src_ranges)
            else:
                ga = _launch("gemm", lambda: torch.bmm(gxh, w.w2))                                            # [T, Ns, H]
                gh = nodeops.ssilu_bwd(ga, h, Ns, T, H, H, Ns * H)                   # [Ns, T*H]
                gn = _launch("gemm", lambda: torch.mm(gh, w.w1cat))                                           # [N, H]
                gx_total = nodeops.layernorm_bwd(gn, x, mean, rstd, add=gx_in, h_real=w.h_real)
        return (gx_total, gvec_in, ge) + (None,) * 10


class EnergyHead(torch.autograd.Function):
    """`out_energy` (hermnet.py:113-117,129) on relation-ordered rows: Linear (library GEMM, bias in the epilogue),
    then ScaledSiLU + the H/2 -> 1 Linear in one kernel; one autograd node with a hand-written backward.
    Parameters are constants here (eval() mode; train() mode runs the nn.Sequential)."""

    @staticmethod
    def forward(ctx, x, w0, b0, w2, b2, mask=None):
        """`mask` [N] (optional) multiplies the per-row energies: padding rows of the relation order -> 0."""
        w2v = w2.reshape(-1).contiguous()
        ctx.mask = mask
        ctx.mfma = x.is_cuda and nodeops.head16_supported(x.size(1), w0.size(0))
        if ctx.mfma:       # the H -> C product on the matrix pipe (csrc/node_chain16.hip), one launch each way
            w0c = w0.contiguous()
            wf, wtf = _head_fragments(w0c)
            h, e = nodeops.energy_head16_fwd(x, wf, b0, w2v, b2, mask)
            ctx.save_for_backward(h, wtf, w2v)
            ctx.fused, ctx.H = True, x.size(1)
            return e
        ctx.fused = x.is_cuda and nodeops.head_fused_supported(x.size(1), w0.size(0))
        if ctx.fused:      # one launch each way, no library GEMM (csrc/node_kernels.hip: energy_head_fused_kernel)
            w0c = w0.contiguous()
            h, e = nodeops.energy_head_fused_fwd(x, _transposed_once(w0c), b0, w2v, b2, mask)
            ctx.save_for_backward(h, w0c, w2v)
            return e
        h = _launch("gemm", lambda: torch.addmm(b0, x, w0.t()))                       # [N, H/2]
        ctx.save_for_backward(h, w0, w2v)
        return nodeops.energy_head_fwd(h, w2v, b2, mask)      # [N]

    @staticmethod
    def backward(ctx, ge):
        h, w0, w2v = ctx.saved_tensors
        if ctx.mfma:
            return nodeops.energy_head16_bwd(ge.contiguous(), h, w0, w2v, ctx.H, ctx.mask), None, None, None, None, None
        if ctx.fused:
            return nodeops.energy_head_fused_bwd(ge.contiguous(), h, w0, w2v, ctx.mask), None, None, None, None, None
        gh = nodeops.energy_head_bwd(ge.contiguous(), h, w2v, ctx.mask)
        return _launch("gemm", lambda: torch.mm(gh, w0)), None, None, None, None, None


_T_CACHE = []


def _head_fragments(w):
    """(frag16(W0), frag16(W0^T)) of the read-out's first weight, rebuilt only when it changes (the cache of `_transposed_once`;
    guard.ParamGuard covers writes through `.data`)."""
    key = ("frag16", w.data_ptr(), w._version, tuple(w.shape))
    for ent in _T_CACHE:
        if ent[0] == key:
            return ent[2]
    fr = (nodeops.weight_fragments16(w), nodeops.weight_fragments16(w.t().contiguous()))
    _T_CACHE.insert(0, (key, w, fr))
    del _T_CACHE[4:]
    return fr


def _transposed_once(w):
    """w^T (contiguous), rebuilt only when the parameter changes (storage address, version, shape; the cache holds the
    source tensor, so its address cannot be reused while the entry lives)."""
    key = (w.data_ptr(), w._version, tuple(w.shape))
    for ent in _T_CACHE:
        if ent[0] == key:
            return ent[2]
    wt = w.t().contiguous()
    _T_CACHE.insert(0, (key, w, wt))
    del _T_CACHE[4:]
    return wt
