"""Relational message-passing modules: parameter containers with the reference's names
and state_dict layout (`HermNet/rmnet.py`), whose edge-level work runs in the gfx950
kernels of `hermnet_amd/csrc/message_kernels.hip`.

What stays in PyTorch here is node-level dense algebra (LayerNorm + the two MLPs,
`rmnet.py:52` and `rmnet.py:94-107`): plain library GEMMs with elementwise epilogues.
"""
import math
import os

import torch
from torch import nn
import torch.nn.functional as F

from . import _lib
from .ops import MessageScatter, RbfDescriptor


class ScaledSiLU(nn.Module):
    """`rmnet.py:110-117`."""

    def __init__(self):
        super().__init__()
        self.scale_factor = 1 / 0.6

    def forward(self, x):
        return F.silu(x) * self.scale_factor


class GaussianSmearing(nn.Module):
    """PyG `GaussianSmearing(start, stop, num_gaussians)` as used at `rmnet.py:156-158`:
    buffer `offset` = linspace, python-float `coeff`."""

    def __init__(self, start=0.0, stop=5.0, num_gaussians=50):
        super().__init__()
        offset = torch.linspace(start, stop, num_gaussians)
        self.coeff = -0.5 / (offset[1] - offset[0]).item() ** 2
        self.register_buffer("offset", offset)


class PolynomialEnvelope(nn.Module):
    """`rmnet.py:175-193` (evaluated inside the kernels: `hermnet_math.h: hn_envelope`)."""

    def __init__(self, exponent):
        super().__init__()
        assert exponent > 0
        self.p = int(exponent)
        self.kind = _lib.HN_ENV["polynomial"]


class ExponentialEnvelope(nn.Module):
    """`rmnet.py:196-208`."""

    def __init__(self):
        super().__init__()
        self.p = 0
        self.kind = _lib.HN_ENV["exponential"]


class SphericalBesselBasis(nn.Module):
    """`rmnet.py:211-233` (parameters only; see RadialBasis.descriptor)."""

    def __init__(self, num_radial, cutoff):
        super().__init__()
        self.norm_const = math.sqrt(2 / (cutoff ** 3))
        self.frequencies = nn.Parameter(data=math.pi * torch.arange(1, num_radial + 1).float(), requires_grad=True)


class BernsteinBasis(nn.Module):
    """`rmnet.py:236-275` (parameters only)."""

    def __init__(self, num_radial, pregamma_initial=0.45264):
        super().__init__()
        from scipy.special import binom
        import numpy as np
        self.register_buffer("prefactor", torch.tensor(binom(num_radial - 1, np.arange(num_radial)), dtype=torch.float),
                             persistent=False)
        self.pregamma = nn.Parameter(data=torch.tensor(pregamma_initial, dtype=torch.float), requires_grad=True)
        exp1 = torch.arange(num_radial)
        self.register_buffer("exp1", exp1[None, :], persistent=False)
        self.register_buffer("exp2", (num_radial - 1 - exp1)[None, :], persistent=False)


class RadialBasis(nn.Module):
    """`rmnet.py:134-172`.  The basis is never materialised as an [E, R] tensor: the
    kernels evaluate envelope * Gaussian taps per edge in registers."""

    def __init__(self, num_radial, cutoff, rbf={"name": "gaussian"}, envelope={"name": "polynomial", "exponent": 5}):
        super().__init__()
        self.inv_cutoff = 1 / cutoff
        self.cutoff = cutoff
        self.num_radial = num_radial
        env_name = envelope["name"].lower()
        env_hparams = {k: v for k, v in envelope.items() if k != "name"}
        if env_name == "polynomial":
            self.envelope = PolynomialEnvelope(**env_hparams)
        elif env_name == "exponential":
            self.envelope = ExponentialEnvelope(**env_hparams)
        else:
            raise ValueError(f"Unknown envelope function '{env_name}'.")
        rbf_name = rbf["name"].lower()
        rbf_hparams = {k: v for k, v in rbf.items() if k != "name"}
        self.rbf_name = rbf_name
        if rbf_name == "gaussian":
            self.rbf = GaussianSmearing(start=0, stop=1, num_gaussians=num_radial, **rbf_hparams)
        elif rbf_name == "spherical_bessel":
            self.rbf = SphericalBesselBasis(num_radial=num_radial, cutoff=cutoff, **rbf_hparams)
        elif rbf_name == "bernstein":
            self.rbf = BernsteinBasis(num_radial=num_radial, **rbf_hparams)
        else:
            raise ValueError(f"Unknown radial basis function '{rbf_name}'.")
        self._desc = None

    # largest Gaussian basis whose rbf_proj column block fits the backward kernel's LDS tile ((R + 23) tap rows x 64
    # channels x 16 B <= 160 KiB, csrc/message_bwd_cl.hip; the forward tile would take R <= 190)
    FUSED_MAX_RBF = 137

    @property
    def fused(self):
        """True when the basis is the one the fused gfx950 kernel evaluates in registers: Gaussian (the reference
        default, hermnet.py:87) with at most FUSED_MAX_RBF functions.  Larger Gaussian bases (the reference accepts any
        `num_rbf`, hermnet.py:86) take the same route as the Bessel / Bernstein bases: basis materialised from the
        kernel's distances, rbf_proj as a library GEMM, gather / index_add device ops."""
        return self.rbf_name == "gaussian" and self.num_radial <= self.FUSED_MAX_RBF

    def forward(self, d):
        """`rmnet.py:168-172` as differentiable device ops: envelope(d/rc)[:,None] * rbf(d/rc) -> [E,R].
        Only used for the optional bases (spherical Bessel, Bernstein); the Gaussian default never
        materialises this tensor."""
        u = d * self.inv_cutoff
        if isinstance(self.envelope, PolynomialEnvelope):
            p = self.envelope.p
            a, b, c = -(p + 1) * (p + 2) / 2, p * (p + 2), -p * (p + 1) / 2
            env = 1 + a * u ** p + b * u ** (p + 1) + c * u ** (p + 2)
        else:
            env = torch.exp(-(u ** 2) / ((1 - u) * (1 + u)))
        env = torch.where(u < 1, env, torch.zeros_like(u))
        if self.rbf_name == "gaussian":
            rbf = torch.exp(self.rbf.coeff * (u.view(-1, 1) - self.rbf.offset.view(1, -1)) ** 2)
        elif self.rbf_name == "spherical_bessel":
            rbf = self.rbf.norm_const / u[:, None] * torch.sin(self.rbf.frequencies * u[:, None])
        else:
            gamma = F.softplus(self.rbf.pregamma)
            ed = torch.exp(-gamma * u)[:, None]
            rbf = self.rbf.prefactor * (ed ** self.rbf.exp1) * ((1 - ed) ** self.rbf.exp2)
        return env[:, None] * rbf

    def bucketed(self, d, bounds, T):
        """`BucketedBasis` of the edges [0, bounds[T]) (CSR order: relation t owns [bounds[t], bounds[t+1])) with
        distances d.  Window of bucket b: centres b*20 - 5 .. b*20 + 26; an edge with floor(u/delta) - 5 = lo goes to
        bucket (lo + 5) // 20, so the twelve centres lo .. lo + 11 around u lie inside.  One host read (group sizes)."""
        Ek, dev = bounds[T], d.device
        W, C = BucketedBasis.WIDTH, BucketedBasis.CHUNK
        S = W - 12
        off = self.rbf.offset
        R = off.numel()
        delta = float(off[1] - off[0]) if R > 1 else 1.0
        nb = (R + 4) // S + 1
        u = d[:Ek] * self.inv_cutoff
        lo = torch.floor(u.detach() / delta).clamp(min=0, max=R + S).long() - 5
        bucket = ((lo + 5) // S).clamp(max=nb - 1)
        rel = torch.repeat_interleave(torch.arange(T, device=dev),
                                      torch.tensor([bounds[t + 1] - bounds[t] for t in range(T)], device=dev))
        key = rel * nb + bucket
        cnt = torch.bincount(key, minlength=T * nb)
        cnt_h = cnt.tolist()                                                                 # the host read
        chunks = [(c + C - 1) // C for c in cnt_h]
        nc = sum(chunks)
        start_pad = torch.tensor([0] + chunks[:-1], device=dev).cumsum(0) * C                # first padded row of a group
        start = torch.cumsum(cnt, 0) - cnt
        order = torch.argsort(key, stable=True)
        ks = key[order]
        slot = torch.empty(Ek, dtype=torch.long, device=dev)
        slot[order] = start_pad[ks] + torch.arange(Ek, device=dev) - start[ks]
        src = torch.full((nc * C,), Ek, dtype=torch.long, device=dev)                        # padding rows -> the dummy entry
        src[slot] = torch.arange(Ek, device=dev)
        group = torch.repeat_interleave(torch.arange(T * nb, device=dev), torch.tensor(chunks, device=dev))
        up = torch.cat([u, u.new_zeros(1)]).index_select(0, src)                              # [nc * C], differentiable
        if isinstance(self.envelope, PolynomialEnvelope):
            p = self.envelope.p
            a, b, c = -(p + 1) * (p + 2) / 2, p * (p + 2), -p * (p + 1) / 2
            env = 1 + a * up ** p + b * up ** (p + 1) + c * up ** (p + 2)
        else:   # (evaluated at 0 beyond the cutoff: exp(+large) there would turn the masked branch's zero gradient into NaN)
            us = torch.where(up < 1, up, torch.zeros_like(up))
            env = torch.exp(-(us ** 2) / ((1 - us) * (1 + us)))
        env = torch.where((up < 1) & (src < Ek), env, torch.zeros_like(up))
        k = ((group % nb) * S - 5)[:, None] + torch.arange(W, device=dev)[None, :]           # [nc, 32] centre indices
        mu = off[k.clamp(0, R - 1)]
        colok = ((k >= 0) & (k < R)).to(up.dtype)
        phi = torch.exp(self.rbf.coeff * (up.view(nc, C, 1) - mu[:, None, :]) ** 2) * (env.view(nc, C, 1) * colok[:, None, :])
        return BucketedBasis(phi, group, slot, nb, R)

    def descriptor(self):
        if self.rbf_name != "gaussian":
            raise NotImplementedError("the fused kernel evaluates the Gaussian basis only; use RadialBasis.forward")
        off = self.rbf.offset
        if self._desc is None or self._desc.offset.data_ptr() != off.data_ptr():
            self._desc = RbfDescriptor(off, self.cutoff, self.envelope.kind, self.envelope.p)
        return self._desc


class PaiNNMessage(nn.Module):
    """Parameters of `rmnet.py:35-76`.  Node part (`x_proj(LayerNorm(x))`) runs here; the edge
    part (rbf_proj, gather, message, aggregate) runs fused for all relations in
    `hermnet_message_scatter_fwd` (see HeteroVertexConv)."""

    def __init__(self, hidden_channels, num_rbf):
        super().__init__()
        self.hidden_channels = hidden_channels
        self.x_proj = nn.Sequential(
            nn.Linear(hidden_channels, hidden_channels),
            ScaledSiLU(),
            nn.Linear(hidden_channels, hidden_channels * 3),
        )
        self.rbf_proj = nn.Linear(num_rbf, hidden_channels * 3)
        self.inv_sqrt_3 = 1 / math.sqrt(3.0)
        self.inv_sqrt_h = 1 / math.sqrt(hidden_channels)
        self.x_layernorm = nn.LayerNorm(hidden_channels)

    def node_projection(self, x):
        """xh = x_proj(LayerNorm(x))  (`rmnet.py:52`)."""
        return self.x_proj(self.x_layernorm(x))


class PaiNNUpdate(nn.Module):
    """`rmnet.py:79-107` (node-level; dense GEMMs + elementwise)."""

    def __init__(self, hidden_channels):
        super().__init__()
        self.hidden_channels = hidden_channels
        self.vec_proj = nn.Linear(hidden_channels, hidden_channels * 2, bias=False)
        self.xvec_proj = nn.Sequential(
            nn.Linear(hidden_channels * 2, hidden_channels),
            ScaledSiLU(),
            nn.Linear(hidden_channels, hidden_channels * 3),
        )
        self.inv_sqrt_2 = 1 / math.sqrt(2.0)
        self.inv_sqrt_h = 1 / math.sqrt(hidden_channels)

    def forward(self, x, vec):
        H = self.hidden_channels
        vec1, vec2 = torch.split(self.vec_proj(vec), H, dim=-1)
        vec_dot = (vec1 * vec2).sum(dim=1) * self.inv_sqrt_h
        x_vec_h = self.xvec_proj(torch.cat([x, torch.sqrt(torch.sum(vec2 ** 2, dim=-2) + 1e-8)], dim=-1))
        xvec1, xvec2, xvec3 = torch.split(x_vec_h, H, dim=-1)
        dx = (xvec1 + xvec2 * vec_dot) * self.inv_sqrt_2
        dvec = xvec3.unsqueeze(1) * vec1
        return dx, dvec


class PaiNNModule(nn.Module):
    """`rmnet.py:11-32`: message + residual + update for one relation."""

    def __init__(self, hidden_channels=512, num_rbf=128):
        super().__init__()
        self.num_rbf = num_rbf
        self.message_layer = PaiNNMessage(hidden_channels, num_rbf)
        self.update_layer = PaiNNUpdate(hidden_channels)
        self.inv_sqrt_2 = 1 / math.sqrt(2.0)


class _RowKey(object):
    """One edge -> row assignment `idx` [K] (values < n_rows) in both forms the pair below needs: the index vector for
    the gather, and (perm, lengths) -- the edges sorted by row and the run lengths of all n_rows rows -- for a sum
    without atomics.  perm = None: the edges already come sorted by row."""

    def __init__(self, idx, perm, lengths, n_rows):
        self.idx, self.perm, self.lengths, self.n_rows = idx, perm, lengths, int(n_rows)
        self.rowptr = torch.zeros(self.n_rows + 1, dtype=torch.long, device=idx.device)
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
        if (x.is_cuda and x.dtype == torch.float32 and width % 4 == 0
                and os.environ.get("HERMNET_TRAIN_KERNELS", "1") != "0"):
            # one pass: the rows are gathered inside the sum (csrc/train_kernels.hip: hermnet_segment_sum)
            return _segsum(x, key)
        xs = x if key.perm is None else x.index_select(0, key.perm)
        return torch.segment_reduce(xs.contiguous(), "sum", lengths=key.lengths, unsafe=True)

    @staticmethod
    def backward(ctx, g):
        return GatherRows.apply(g, ctx.key), None


class BucketedBasis(object):
    """The Gaussian basis of the first Ek edges, SORTED by (relation of the target, distance bucket) and cut to the 32
    centres of the edge's bucket: phi [nc, C, 32] (chunks of C rows; every (relation, bucket) group is padded to whole
    chunks with zero rows), group [nc] = relation * nb + bucket of each chunk, slot [Ek] = row of every edge in that
    order.  rbf_proj then is ONE batched [C,32] x [32,3H] product per layer instead of three dense [E_t,R] x [R,3H]
    GEMMs: a quarter of the FLOPs (a Gaussian is < 2e-8 of its peak six widths away, and a 32-centre window holds
    every centre within six widths of any distance of its bucket), still on the matrix pipe, still plain torch ops --
    differentiable to any order."""

    WIDTH, CHUNK = 32, 1024

    def __init__(self, phi, group, slot, nb, num_radial):
        self.phi, self.group, self.slot, self.nb, self.num_radial = phi, group, slot, int(nb), int(num_radial)

    def project(self, w_rbf, b_rbf, scale):
        """rbf_proj of every relation (rmnet.py:55) on the bucketed basis -> R [nc * C, 3H] in the sorted edge order;
        `scale` [3H] multiplies the output channels."""
        T, S = len(w_rbf), BucketedBasis.WIDTH - 12
        wt = torch.stack([(w_rbf[t] * scale[:, None]).t() for t in range(T)])               # [T, R, 3H]
        need = (self.nb - 1) * S + BucketedBasis.WIDTH                                       # rows the windows reach
        wt = F.pad(wt, (0, 0, 5, max(need - 5 - wt.size(1), 0)))                             # centre k sits at row k + 5
        win = wt.unfold(1, BucketedBasis.WIDTH, S)[:, :self.nb]                              # [T, nb, 3H, 32]
        win = win.permute(0, 1, 3, 2).reshape(T * self.nb, BucketedBasis.WIDTH, -1)
        bias = torch.stack([b_rbf[t] * scale for t in range(T)])                             # [T, 3H]
        wc = win.index_select(0, self.group)
        bc = bias.index_select(0, self.group // self.nb)
        return torch.baddbmm(bc[:, None, :], self.phi, wc).reshape(-1, wc.size(2))


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


class MessageAlgebra(torch.autograd.Function):
    """(xh [T N,3H], vec [N,3,H] | None, R [E,3H], U [E,3]) -> (dx [N,H], dvec [N,3,H]): gather of x_j / vec_j, the
    per-edge algebra and the aggregation of rmnet.py:58-73 with NODE-level inputs and outputs.  The kernels of
    `EdgeMessage` read their gathered operands through row indices (x_j = xh[(relation, source)], vec_j = vec[source],
    cotangents = g[target]) and the row sums follow inside the function, so no [E, 3H] copy of a gathered operand and
    no per-edge gradient ever enters the autograd graph: what two graph nodes share and the engine has to add up is
    node-sized.  Twice differentiable through `MessageAlgebraGrad`.  keys = (targets, sources, (relation, source), rows
    of R or None)."""

    @staticmethod
    def forward(ctx, xh, vec, R, U, keys):
        from .ops import _stream
        k_tgt, k_all, k_xh, r_rows = keys
        xh, vec, R, U = _c(xh), _c(vec), _c(R), _c(U)
        E, H = U.size(0), R.size(1) // 3
        S = torch.empty(E, H, dtype=R.dtype, device=R.device)
        M = torch.empty(E, 3, H, dtype=R.dtype, device=R.device)
        P = _lib.ptr
        _lib.check(_lib.load().hermnet_edge_message_fwd(P(xh), P(R), P(vec), P(U), E, H, P(k_xh.idx), P(k_all.idx), P(r_rows),
                                                        P(S), P(M), _stream()), "hermnet_edge_message_fwd")
        ctx.save_for_backward(xh, vec, R, U)
        ctx.keys = keys
        return _segsum(S, k_tgt), _segsum(M, k_tgt)

    @staticmethod
    def backward(ctx, g_dx, g_dv):
        xh, vec, R, U = ctx.saved_tensors
        g_xh, g_vec, gR, gU = MessageAlgebraGrad.apply(g_dx, g_dv, xh, vec, R, U, ctx.keys)
        return g_xh, g_vec, gR, gU, None


class MessageAlgebraGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, g_dx, g_dv, xh, vec, R, U, keys):
        from .ops import _stream
        k_tgt, k_all, k_xh, r_rows = keys
        E, H = U.size(0), R.size(1) // 3
        g_dx = torch.zeros(k_tgt.n_rows, H, dtype=R.dtype, device=R.device) if g_dx is None else _c(g_dx)
        g_dv = torch.zeros(k_tgt.n_rows, 3, H, dtype=R.dtype, device=R.device) if g_dv is None else _c(g_dv)
        gX = torch.empty(E, 3 * H, dtype=R.dtype, device=R.device)
        # (R kept in another edge order with padding rows: rows no edge points to get no gradient)
        gR = torch.empty_like(R) if r_rows is None else torch.zeros_like(R)
        gV = None if vec is None else torch.empty(E, 3, H, dtype=R.dtype, device=R.device)
        gU = torch.empty_like(U)
        P = _lib.ptr
        _lib.check(_lib.load().hermnet_edge_message_bwd(P(g_dx), P(g_dv), P(xh), P(R), P(vec), P(U), E, H, P(k_xh.idx),
                                                        P(k_all.idx), P(k_tgt.idx), P(r_rows), P(gX), P(gR), P(gV), P(gU),
                                                        _stream()), "hermnet_edge_message_bwd")
        ctx.save_for_backward(g_dx, g_dv, xh, vec, R, U)
        ctx.keys = keys
        return _segsum(gX, k_xh), (None if vec is None else _segsum(gV, k_all)), gR, gU

    @staticmethod
    def backward(ctx, c_xh, c_vec, cR, cU):
        from .ops import _stream
        g_dx, g_dv, xh, vec, R, U = ctx.saved_tensors
        k_tgt, k_all, k_xh, r_rows = ctx.keys
        E, H = U.size(0), R.size(1) // 3
        c_xh, c_vec, cR, cU = _c(c_xh), _c(c_vec), _c(cR), _c(cU)
        new = lambda *shape: torch.empty(*shape, dtype=R.dtype, device=R.device)
        dGS, dGM, dX, dU = new(E, H), new(E, 3, H), new(E, 3 * H), new(E, 3)
        dR = torch.empty_like(R) if r_rows is None else torch.zeros_like(R)
        dV = None if vec is None else new(E, 3, H)
        P = _lib.ptr
        _lib.check(_lib.load().hermnet_edge_message_bwd2(P(c_xh), P(cR), P(c_vec), P(cU), P(g_dx), P(g_dv), P(xh), P(R), P(vec),
                                                         P(U), E, H, P(k_xh.idx), P(k_all.idx), P(k_tgt.idx), P(r_rows), P(dGS),
                                                         P(dGM), P(dX), P(dR), P(dV), P(dU), _stream()),
                   "hermnet_edge_message_bwd2")
        return (_segsum(dGS, k_tgt), _segsum(dGM, k_tgt), _segsum(dX, k_xh), (None if vec is None else _segsum(dV, k_all)),
                dR, dU, None)


def _train_kernels(t):
    return (t.is_cuda and t.dtype == torch.float32 and (t.size(-1) // 3) % 4 == 0 and t.size(0) > 0
            and os.environ.get("HERMNET_TRAIN_KERNELS", "1") != "0")


def edge_message(X, R, V, U):
    """S, M of the per-edge message algebra: the kernels on the GPU (fp32, width a multiple of 4), torch ops otherwise."""
    if _train_kernels(X):
        return EdgeMessage.apply(X, R, V, U)
    return _edge_message_torch(X, R, V, U)


def _row_keys(graph, T, Nt, Ns, bounds):
    """(key of the targets of the first Ek CSR edges [Nt target rows], key of their sources [Ns source rows], key of
    their (relation, source) rows of xh.view(T Ns, 3H), key of the residual rows or None) for `message_scatter_generic`,
    from the graph's CSR / CSC orders; built once per graph.  Nt = Ns for HVNet; HTNet has one target row per atom and
    pair relation and reads the residual from the atom's own source row (`graph.res_row`)."""
    keys = getattr(graph, "_row_keys", None)
    if keys is not None:
        return keys
    dev = graph.csr_rowptr.device
    rowptr = graph.csr_rowptr.long()
    Ek = bounds[T]
    nk = int(graph.type_rowptr_host[-1])
    lengths = rowptr[1:] - rowptr[:-1]
    lengths = torch.cat([lengths[:nk], lengths.new_zeros(Nt - nk)])          # edges into unknown-element rows: not summed
    tgt_row = torch.repeat_interleave(torch.arange(Nt, device=dev), lengths)
    k_tgt = _RowKey(tgt_row, None, lengths, Nt)
    src = graph.csr_src.long()
    crp, cpos = graph.csc_rowptr.long(), graph.csc_pos.long()               # groups (relation, source row) over CSR positions
    # all relations at once: sorted by (source row) = the T groups of a row merged; built by one stable sort
    order = torch.argsort(src[:Ek], stable=True)
    k_all = _RowKey(src[:Ek], order, torch.bincount(src[:Ek], minlength=Ns), Ns)
    # rows of xh.view(T * Ns, 3H): (relation of the edge's target, source row) -- the CSC groups themselves
    rel_of_edge = torch.repeat_interleave(torch.arange(T, device=dev),
                                          torch.tensor([bounds[t + 1] - bounds[t] for t in range(T)], device=dev))
    k_xh = _RowKey(rel_of_edge * Ns + src[:Ek], cpos[:Ek], crp[1:T * Ns + 1] - crp[:T * Ns], T * Ns)
    k_res = None
    if graph.res_row is not None:
        res = graph.res_row.long()
        k_res = _RowKey(res, torch.argsort(res, stable=True), torch.bincount(res, minlength=Ns), Ns)
    graph._row_keys = (k_tgt, k_all, k_xh, k_res)
    return graph._row_keys


def message_scatter_generic(xh, vec, x, edge, edge_embed, w_rbf, b_rbf, graph):
    """Same contract as the fused kernel (rmnet.py:24-26,55-73) for a MATERIALISED basis `edge_embed`
    [E,R] (CSR order): library GEMM per relation + gather / segmented-sum device ops, differentiable to any order by
    PyTorch autograd.  Path of train() mode (parameter gradients, create_graph=True) and of the optional
    radial bases."""
    T, Ns, H3 = xh.shape                                       # Ns source rows (rows of x / vec / xh[t])
    H = H3 // 3
    N = graph.N                                                # target rows (= Ns for HVNet; HTNet: one per atom and pair)
    rel_row = torch.bucketize(torch.arange(N, device=x.device), graph.type_rowptr.long()[1:], right=True)
    # rows are relation-ordered and CSR is row-ordered: the edges of relation t are ONE contiguous CSR range
    # (no per-relation masks or gathers of the edge arrays)
    bounds = graph.rel_edge_bounds()
    Ek = bounds[T]                                             # edges whose target has a known element
    k_tgt, k_all, k_xh, k_res = _row_keys(graph, T, N, Ns, bounds)
    # the constant factors of the vector message (1/sqrt(3H) on the `a` part, 1/sqrt(H) on `b`, rmnet.py:64-66) ride on
    # the [3H, R] projection weights, not on per-edge tensors
    sc = x.new_ones(3 * H)
    sc[H:2 * H] = 1 / math.sqrt(3.0 * H)
    sc[2 * H:] = 1 / math.sqrt(H)
    dx = x.new_zeros(N, H)
    dv = x.new_zeros(N, 3, H)
    parts = []
    if isinstance(edge_embed, BucketedBasis):
        if Ek > 0:     # rbf_proj (rmnet.py:55) as one batched product on the bucketed basis; R stays in its sorted order
            R = edge_embed.project(w_rbf, b_rbf, sc)
            dx, dv = MessageAlgebra.apply(xh.reshape(T * Ns, 3 * H), vec, R, edge[:Ek, :3], (k_tgt, k_all, k_xh, edge_embed.slot))
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
            dx, dv = MessageAlgebra.apply(xh.reshape(T * Ns, 3 * H), vec, R, edge[:Ek, :3], (k_tgt, k_all, k_xh, None))
        else:
            X = GatherRows.apply(xh.reshape(T * Ns, 3 * H), k_xh)                  # x_j of every edge, rmnet.py:58
            V = None if vec is None else GatherRows.apply(vec, k_all)
            S, M = edge_message(X, R, V, edge[:Ek, :3])
            dx = SumRows.apply(S, k_tgt)
            dv = SumRows.apply(M, k_tgt)
    known = (rel_row < T).to(x.dtype)
    # the residual (rmnet.py:24-26) reads the target atom's own row: the same row for HVNet, `res_row` for HTNet
    xr = x if k_res is None else GatherRows.apply(x, k_res)
    vr = 0 if vec is None else (vec if k_res is None else GatherRows.apply(vec, k_res))
    x1 = (xr + dx) * (1 / math.sqrt(2.0)) * known[:, None]
    vec1 = (vr + dv) * known[:, None, None]
    return x1, vec1


def _relational_layer_batched(mlist, x, vec, edge, graph, edge_embed):
    """`relational_layer` for the uniform row layout with every per-relation Linear run as ONE batched GEMM over
    the [T, block, .] view of the rows (stacked parameters; gradients flow back through the stacking).  Same
    arithmetic as the per-relation loop, a third of the kernel launches: the differentiable path is launch-bound."""
    T, N, H = len(mlist), graph.N, x.size(1)           # N target rows (x / vec hold graph.num_src source rows for HTNet)
    B, nk = graph.block, graph.type_rowptr_host[-1]
    ml = [m.message_layer for m in mlist]
    ul = [m.update_layer for m in mlist]
    st = lambda ts: torch.stack(list(ts), 0)
    # --- x_proj(LayerNorm(x)) of every relation for every row (rmnet.py:52)
    n = F.layer_norm(x, (H,))
    g, b = st(m.x_layernorm.weight for m in ml), st(m.x_layernorm.bias for m in ml)                 # [T,H]
    w1, b1 = st(m.x_proj[0].weight for m in ml), st(m.x_proj[0].bias for m in ml)                   # [T,H,H], [T,H]
    w2, b2 = st(m.x_proj[2].weight for m in ml), st(m.x_proj[2].bias for m in ml)                   # [T,3H,H], [T,3H]
    xn = n.unsqueeze(0) * g[:, None, :] + b[:, None, :]                                              # [T,N,H]
    h = torch.baddbmm(b1[:, None, :], xn, w1.transpose(1, 2))
    xh = torch.baddbmm(b2[:, None, :], F.silu(h) * ml[0].x_proj[1].scale_factor, w2.transpose(1, 2))   # [T,N,3H]
    x1, vec1 = message_scatter_generic(xh, vec, x, edge, edge_embed, [m.rbf_proj.weight for m in ml],
                                       [m.rbf_proj.bias for m in ml], graph)
    # --- PaiNNUpdate on the rows of each relation (rmnet.py:94-107), blocks of B rows
    wv = st(u.vec_proj.weight for u in ul)                                                           # [T,2H,H]
    wx0, bx0 = st(u.xvec_proj[0].weight for u in ul), st(u.xvec_proj[0].bias for u in ul)           # [T,H,2H], [T,H]
    wx2, bx2 = st(u.xvec_proj[2].weight for u in ul), st(u.xvec_proj[2].bias for u in ul)           # [T,3H,H], [T,3H]
    xt, vt = x1[:nk].view(T, B, H), vec1[:nk].view(T, B, 3, H)
    vp = torch.bmm(vt.reshape(T, B * 3, H), wv.transpose(1, 2)).view(T, B, 3, 2 * H)
    v1, v2 = vp.view(T, B, 3, 2, H).unbind(3)
    vdot = (v1 * v2).sum(dim=2) * ul[0].inv_sqrt_h
    xin = torch.cat([xt, torch.sqrt((v2 ** 2).sum(dim=2) + 1e-8)], dim=-1)                           # [T,B,2H]
    h2 = torch.baddbmm(bx0[:, None, :], xin, wx0.transpose(1, 2))
    q = torch.baddbmm(bx2[:, None, :], F.silu(h2) * ul[0].xvec_proj[1].scale_factor, wx2.transpose(1, 2))
    q1, q2, q3 = q.view(T, B, 3, H).unbind(2)
    xo = xt + (q1 + q2 * vdot) * ul[0].inv_sqrt_2
    vo = vt + q3.unsqueeze(2) * v1
    x_out, v_out = xo.reshape(nk, H), vo.reshape(nk, 3, H)
    if nk < N:   # atoms whose element is not in `elems`: zero rows (hermnet.py:51)
        x_out = torch.cat([x_out, x.new_zeros(N - nk, H)], 0)
        v_out = torch.cat([v_out, x.new_zeros(N - nk, 3, H)], 0)
    return x_out * graph.row_active[:, None], v_out * graph.row_active[:, None, None]


def relational_layer(mods, x, vec, edge, graph, rbf, edge_embed=None):
    """One HeteroVertexConv layer in relation (row) order: `hermnet.py:37-65`.

    x [N,H], vec [N,3,H] or None (layer 0: zeros, `hermnet.py:124`); returns new (x, vec).
    """
    mlist = list(mods)
    H = x.size(1)
    if edge_embed is not None and graph.uniform and graph.type_rowptr_host[-1] > 0:
        return _relational_layer_batched(mlist, x, vec, edge, graph, edge_embed)
    # xh_t for every row and relation (sources of relation t carry the TARGET type's projection, SURVEY A5 i)
    xh = torch.stack([m.message_layer.node_projection(x) for m in mlist], dim=0)
    wt = torch.stack([m.message_layer.rbf_proj.weight.detach().t() for m in mlist], dim=0).contiguous()
    brbf = torch.stack([m.message_layer.rbf_proj.bias.detach() for m in mlist], dim=0).contiguous()
    if edge_embed is None:
        x1, vec1 = MessageScatter.apply(xh, vec, x, edge, wt, brbf, graph, rbf)
    else:
        x1, vec1 = message_scatter_generic(xh, vec, x, edge, edge_embed,
                                           [m.message_layer.rbf_proj.weight for m in mlist],
                                           [m.message_layer.rbf_proj.bias for m in mlist], graph)
    xs, vs = [], []
    rp = graph.type_rowptr_host
    for t, m in enumerate(mlist):
        lo, hi = rp[t], rp[t + 1]
        if hi == lo:
            continue
        xt, vt = x1[lo:hi], vec1[lo:hi]
        dx, dvec = m.update_layer(xt, vt)
        xs.append(xt + dx)
        vs.append(vt + dvec)
    nk = rp[-1]
    if nk < graph.N:   # atoms whose element is not in `elems`: zero rows (hermnet.py:51)
        xs.append(x.new_zeros(graph.N - nk, H))
        vs.append(x.new_zeros(graph.N - nk, 3, H))
    x_out = torch.cat(xs, 0) if len(xs) != 1 else xs[0]
    v_out = torch.cat(vs, 0) if len(vs) != 1 else vs[0]
    # relations without any edge are skipped by the reference (hermnet.py:56-57); padding rows stay zero
    x_out = x_out * graph.row_active[:, None]
    v_out = v_out * graph.row_active[:, None, None]
    return x_out, v_out
