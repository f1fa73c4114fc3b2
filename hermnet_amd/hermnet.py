"""HVNet / HeteroVertexConv with the reference's constructor, `forward(data)` signature and
state_dict keys (`HermNet/hermnet.py`), executing the hot path on MI355X."""
import os
from typing import Dict, List, Union

import torch
from torch import nn

from . import switches
from .elements import atomic_numbers
from .ops import EdgeGeometry, TrueEdgeGradient, edge_radial_table
from .relations import RelationalGraph
from .sharding import HaloExchange, HaloExchangeFeatures, HaloGradReturn, SumAcrossRanks
from .layer import (EdgeFanout, EdgeGradSink, EnergyHead, FusedRelationalLayer, LayerWeights, _node_chain_enabled, _PENDING,
                    _PRE_NEXT)
from .rmnet import PaiNNModule, RadialBasis, ScaledSiLU, relational_layer


class HeteroVertexConv(nn.Module):
    """`hermnet.py:11-65`: one PaiNNModule per element type; weights are selected by the
    TARGET atom's type.  All relations run in one fused launch."""

    def __init__(self, mods: Dict[str, nn.Module]):
        super().__init__()
        self.mods = nn.ModuleDict(mods)
        self._weights = None

    def forward(self, data):
        """data must come from `HVNet.forward` (carries the relation-ordered graph)."""
        g = data.get("_hn_graph")
        if g is None:
            raise RuntimeError("HeteroVertexConv.forward needs a Data prepared by HVNet.forward")
        # atom shards: the halo rows of (x, vec) are still with their owners (HVNet.forward leaves the exchange to the
        # layer that consumes them).  The fused chain path overlaps it with its node projection; every other path runs
        # it first.
        halo, data._hn_halo = data.get("_hn_halo"), None
        w = None
        ready = data.get("_hn_weights")          # (HVNet.forward refreshed every layer's copies up front: guard.ParamGuard)
        if ready is not None and ready[data.get("_hn_layer", 0)].mods[0] is next(iter(self.mods.values())):
            w = ready[data.get("_hn_layer", 0)]
        if w is None and halo is not None and data.get("_hn_edge_embed") is None and switches.fused_layer:
            if self._weights is None:
                self._weights = LayerWeights(self.mods.values())
            w = self._weights.refresh()
        # (the exchange inside the layer needs the chain kernels and the channel-per-lane backward, which takes row ranges;
        # switches.halo_overlap() == "0" runs the blocking exchange in front of the layer instead)
        if halo is not None and not (w is not None and w.chain and _node_chain_enabled()
                                     and halo.fwd_early is not None and g.N * 3 * data.x.size(1) * 4 < 2 ** 32
                                     # (a rank that neither sends nor receives joins the collective in its plain form:
                                     # nothing to hide, and the asynchronous form costs ~15 us of stream hand-offs)
                                     and sum(halo.plan.recv_counts) + sum(halo.plan.send_counts) > 0
                                     and switches.halo_overlap() != "0"):
            data.x, data.vec = HaloExchangeFeatures.apply(data.x, data.vec, halo.plan)
            halo = None
        if data.get("_hn_edge_embed") is not None:
            # optional radial bases (Bessel / Bernstein): materialised basis, device ops + autograd
            data.x, data.vec = relational_layer(self.mods.values(), data.x, data.vec, data._hn_edge, g, None,
                                                edge_embed=data._hn_edge_embed)
            return data
        if not switches.fused_layer:
            # debugging path: same kernels for the edge part, node algebra through PyTorch autograd
            data.x, data.vec = relational_layer(self.mods.values(), data.x, data.vec, data._hn_edge, g, data._hn_rbf)
            return data
        if self._weights is None:
            self._weights = LayerWeights(self.mods.values())
        handles, li = data.get("_hn_edge_handles"), data.get("_hn_layer", 0)
        edge = data._hn_edge if handles is None else handles[li]
        w = w if w is not None else self._weights.refresh()
        # (x, vec) straight from the chain layer below -- both still carry THAT autograd node (an in-place exchange of halo
        # rows or any other op in between replaces it) -- and this layer is their only reader: its backward may hand its
        # input gradients down as partial sums (layer.FusedRelationalLayer, `defer`)
        node = data.get("_hn_chain_node")
        # (not under anomaly detection -- its NaN check would read the not-yet-filled buffers -- and not when a tensor hook
        # would: both are debugging situations, which get the finishing launches; switches.defer_sums() forces them)
        straight = (node is not None and data.vec is not None and data.x.grad_fn is node and data.vec.grad_fn is node
                    and not g.num_src and not torch.is_anomaly_enabled()
                    and not getattr(data.x, "_backward_hooks", None) and not getattr(data.vec, "_backward_hooks", None))
        defer = straight and halo is None and (data.get("_hn_shard") is None or data.get("_hn_lone"))
        # atom shards: the exchange in its "proj" form (layer.FusedRelationalLayer; switches.halo_overlap() "1": the round-4
        # form, x | vec rows with windowed node launches around the exchange; "0": the blocking exchange in front of the layer)
        proj = straight and halo is not None and switches.halo_overlap() not in ("0", "1")
        # the node projection of THIS x, already computed: the first layer's by HVNet.forward (side stream), every later layer's
        # by the fused update launch of the layer below (round 5)
        pre, data._hn_pre0 = data.get("_hn_pre0"), None
        if pre is not None and not (w.chain and _node_chain_enabled() and halo is None
                                    and pre[0].data_ptr() == data.x.data_ptr() and pre[0].shape == data.x.shape):
            pre = None
        # the next layer's weights: its projection of the rows this layer produces can run inside this layer's update launch
        w_next = None
        if ready is not None and li + 1 < len(ready) and halo is None and (data.get("_hn_shard") is None or data.get("_hn_lone")):
            w_next = ready[li + 1]
        data.x, data.vec = FusedRelationalLayer.apply(data.x, data.vec, edge, g, data._hn_rbf, w,
                                                      data.get("_hn_edge_sink"), li, halo, defer,
                                                      None if pre is None else pre[1], w_next, proj)
        data._hn_pre0 = _PRE_NEXT.pop((id(g), li + 1), None)
        data._hn_chain_node = data.x.grad_fn if (w.chain and _node_chain_enabled() and not g.num_src) else None
        return data


def nodeops_tile_rows(Hp):
    """Rows per tile of the node pre kernels at padded width Hp (64 when the width has no chain kernels: unused then)."""
    from . import nodeops
    return nodeops.chain_tile_rows(Hp) or 64


class GraphEnergies(torch.autograd.Function):
    """Per-graph sums of the per-row energies (hermnet.py:130: scatter(per_atom_energy, batch)) as an ORDERED segment
    reduction -- rows -> atoms -> atoms sorted by graph -> segment sums: no atomics, bit-reproducible -- and its adjoint
    `_GraphSpread` (every row reads its graph's gradient: two launches where autograd's own chain through the three steps
    takes ~15).  Both maps are linear and each is the other's backward, so the pair differentiates to any order."""

    @staticmethod
    def forward(ctx, e_rows, graph, batch):
        ctx.graph, ctx.batch, ctx.rows = graph, batch, e_rows.size(0)
        pa = e_rows.index_select(0, graph.row_of_node).index_select(0, graph.graph_perm)
        # (unsafe=True: no host-side validation of `lengths` -- they come from the same `batch` vector; the check would
        # synchronise, which also forbids capturing the step into a hipGraph)
        return torch.segment_reduce(pa, "sum", lengths=graph.graph_lengths, unsafe=True)

    @staticmethod
    def backward(ctx, ge):
        return _GraphSpread.apply(ge, ctx.graph, ctx.batch, ctx.rows), None, None


class _GraphSpread(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ge, graph, batch, rows):
        ctx.graph, ctx.batch = graph, batch
        rg = getattr(graph, "_row_graph", None)
        if rg is None or rg[0].numel() != rows:
            # graph index of every row of e_rows and a 0 / 1 mask of the rows that hold an atom (HTNet: source rows)
            idx = torch.zeros(rows, dtype=torch.long, device=ge.device).scatter_(0, graph.row_of_node, batch)
            real = torch.zeros(rows, dtype=ge.dtype, device=ge.device).scatter_(
                0, graph.row_of_node, torch.ones(graph.row_of_node.numel(), dtype=ge.dtype, device=ge.device))
            rg = graph._row_graph = (idx, real)
        return ge.index_select(0, rg[0]) * rg[1]

    @staticmethod
    def backward(ctx, c_rows):
        return GraphEnergies.apply(c_rows.contiguous(), ctx.graph, ctx.batch), None, None, None


class HVNet(nn.Module):
    """Heterogeneous Vertex Network (`hermnet.py:68-152`).

    forward(data) -> energy [num_graphs]; data needs pos, atomic_number, edge_index, batch and,
    for periodic systems, cell [B,3,3] + edge_shift [E,3].
    """

    def __init__(self, elems: Union[str, List[str]], rc: float = 5.,
                 intensive: bool = False, num_layers: int = 5,
                 hidden_channels: int = 512, num_rbf: int = 128,
                 rbf={"name": "gaussian"},
                 envelope={"name": "polynomial", "exponent": 5}):
        super().__init__()
        if isinstance(elems, str):
            elems = [elems]
        self.elems = list(elems)
        self.rc = rc   # read by the calculators (plugin/ase_interface/calculator.py:49)
        self.num_layers = num_layers
        self.hidden_channels = hidden_channels
        self.intensive = intensive

        self.embed = nn.Embedding(len(atomic_numbers), hidden_channels)
        self.radial_basis = RadialBasis(num_radial=num_rbf, cutoff=rc, rbf=rbf, envelope=envelope)
        self.hermconvs = nn.ModuleList()
        for _ in range(num_layers):
            self.hermconvs.append(HeteroVertexConv(
                mods={ntype: PaiNNModule(hidden_channels=hidden_channels, num_rbf=num_rbf) for ntype in self.elems}))
        self.out_energy = nn.Sequential(
            nn.Linear(hidden_channels, hidden_channels // 2),
            ScaledSiLU(),
            nn.Linear(hidden_channels // 2, 1),
        )

    # ---- cached kernel-ready parameter copies (layer.LayerWeights) and their guard (guard.ParamGuard) -----------------
    def invalidate_caches(self):
        """Drop every derived copy of the parameters (MFMA-order weights, folded biases, transposes): the next forward
        rebuilds them.  Runs by itself on `load_state_dict`, `.to()` / `.cuda()` / `.float()` (`_apply`) and `train()` /
        `eval()` transitions; call it after writing parameters through `.data` (a write the caches' keys cannot see --
        the device-side guard would otherwise answer the next step with NaN and repair on the one after)."""
        from . import layer as _layer
        self.__dict__["_cache_epoch"] = self.__dict__.get("_cache_epoch", 0) + 1    # (captured steps compare it: graph.py)
        for conv in self.hermconvs:
            if getattr(conv, "_weights", None) is not None:
                conv._weights.key = None
        del _layer._T_CACHE[:]
        g = self.__dict__.get("_guard")
        if g is not None:
            g.armed_for, g._event = None, None
        return self

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate_caches()
        return out

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        if "hermconvs" in self._modules:
            self.invalidate_caches()
        return out

    def train(self, mode=True):
        if "hermconvs" in self._modules and mode != self.training:
            self.invalidate_caches()
        return super().train(mode)

    def _refresh_weights(self, dev):
        """Every layer's kernel-ready copies, current; fingerprints recorded / checked on the device (one launch)."""
        from .guard import ParamGuard
        guard_on = dev.type == "cuda" and switches.param_guard()
        for _ in range(2):
            ws = []
            for conv in self.hermconvs:
                if conv._weights is None:
                    conv._weights = LayerWeights(conv.mods.values())
                ws.append(conv._weights.refresh())
            if not guard_on or not ws:
                return ws
            g = self.__dict__.get("_guard")
            if g is None:
                g = self.__dict__["_guard"] = ParamGuard(self)
            # (the stamp: which builds of the layers' copies + which tensors the embedding and the read-out are right now)
            extra = tuple((p_.data_ptr(), p_._version) for p_ in (self.embed.weight, self.out_energy[0].weight,
                                                                 self.out_energy[0].bias, self.out_energy[2].weight,
                                                                 self.out_energy[2].bias))
            if not g.step(dev, tuple((id(w), w.builds) for w in ws) + extra, ws[0].b1cat):
                return ws
            self.invalidate_caches()            # the previous step ran on stale copies (its result was NaN): rebuild
        return ws

    # eval() treats parameters as constants (energy / force evaluation).  Set True to take the differentiable
    # device-op path in eval() as well (fine-tuning or gradient diagnostics with dropout-free eval semantics).
    eval_param_grads = False

    @staticmethod
    def _require_device(pos):
        if not pos.is_cuda:
            raise RuntimeError("hermnet_amd.HVNet runs on MI355X only (data is on %s); there is no CPU fallback"
                               % pos.device)

    def _require_fp32(self, pos):
        """The kernels read raw pointers as float32 (the reference is fp32-only too: `torch.tensor(0.0)` at
        hermnet.py:146 breaks `.double()`), so other dtypes are refused instead of being misread."""
        wd = self.embed.weight.dtype
        if wd != torch.float32 or pos.dtype != torch.float32:
            raise TypeError("hermnet_amd.HVNet computes in float32: parameters are %s, data.pos is %s "
                            "(call model.float() / pos.float())" % (wd, pos.dtype))

    def _build_graph(self, data, zl, shard):
        """Relation-ordered graph of this neighbour list (the replacement of `in_subgraph`, utils.py:11-24)."""
        rel_active = None if shard is None else shard.rel_active(zl)
        return RelationalGraph.build(data.atomic_number, data.edge_index, zl,
                                     edge_shift=data.get("edge_shift") if data.get("cell") is not None else None,
                                     batch=data.batch, rel_active=rel_active)

    @staticmethod
    def _edge_geometry_autograd(pos, cell, graph):
        """`with_edge` (hermnet.py:133-152) as differentiable device ops, CSR order -> (edge [E,4] = (rhat, d), parts): parts =
        (rhat [E,3], d [E]) as tensors of their own where they exist (consumers that take them skip the slices and their
        zero-filled backward), else None.  Training path only: `create_graph=True` needs second derivatives, which the
        geometry kernel's hand-written backward does not provide."""
        src = graph.src_id.long()
        if pos.is_cuda and not graph.num_src and graph.csc_pos is not None:
            from .trainops import EdgeDiff
            D = EdgeDiff.apply(pos, graph)          # (gather and its adjoint as a closed pair: no index sort, no atomics)
        else:
            D = pos[src] - pos[graph.tgt_id.long()]
        if graph.shift is not None and cell is not None:
            c = cell.reshape(-1, 3, 3)
            D = D + torch.einsum("ni,nij->nj", graph.shift.to(D.dtype), c[graph.batch32.long()[src]])
        if D.is_cuda and D.dtype == torch.float32:
            from .trainops import EdgeUnit
            U, d = EdgeUnit.apply(D)                # (one launch per order of differentiation)
            return torch.cat([U, d[:, None]], dim=1), (U, d)
        d = D.norm(dim=-1)
        d = torch.where(d.abs() <= 1.0e-6, torch.full_like(d, 1.0e-6), d)           # hermnet.py:146-147
        return torch.cat([D / d[:, None], d[:, None]], dim=1), None

    def forward(self, data):
        """Two execution modes behind the same signature, chosen like any PyTorch module chooses:
        `eval()` (calculators, MD, validation -- `calculator.py:73`, `lmp_calc.py:46`, `dist_train.py:109`):
        the fused gfx950 kernels, first-order gradients w.r.t. pos / cell; ALL parameters are constants there
        (no parameter receives a gradient; set `model.eval_param_grads = True` to get them in eval()).
        `train()` with grad mode on
        (`dist_train.py:81-99`): every op is a differentiable device op, so `autograd.grad(E, pos,
        create_graph=True)` and `loss.backward()` give the gradients of all parameters."""
        pos = data.pos
        self._require_device(pos)
        self._require_fp32(pos)
        train = (self.training or self.eval_param_grads) and torch.is_grad_enabled()
        if data.get("batch") is None:
            # the reference fails here (scatter(..., None), hermnet.py:130); a single graph is meant
            data.batch = torch.zeros(pos.size(0), dtype=torch.long, device=pos.device)
        zl = [atomic_numbers[el] for el in self.elems]
        # atom-sharded evaluation (hermnet_amd/sharding.py): this rank's atoms + one-hop halo
        shard = data.get("_hn_shard")
        # a padded neighbour list (neighbor.neighbor_search_padded: NULL edges behind the real ones, the count on the device)
        padded = data.get("_hn_edge_count") is not None
        if padded and train:
            raise NotImplementedError("a padded neighbour list runs through the fused eval() path")
        fused = self.radial_basis.fused and not train
        # (the 16-lanes-per-edge backward stages a whole weight tile: num_rbf <= 176; the channel-per-lane form, which takes
        # wider bases in two windows, addresses its gathers with 32-bit offsets -- beyond ~2.3 M local atoms at width 128 such a
        # basis takes the materialised route instead of failing in the backward: ADVICE r5)
        if fused and self.radial_basis.num_radial > 176 and pos.size(0) * 1.2 * 3 * ((self.hidden_channels + 63) // 64 * 64) * 4 >= 2 ** 32:
            fused = False
        graph = self._build_graph(data, zl, shard)
        rbf = self.radial_basis.descriptor() if fused else None
        row_plan = None
        if shard is not None:
            if train:
                raise NotImplementedError("atom-sharded evaluation is first-order (energy/forces); train with "
                                          "DistributedDataParallel over whole graphs (example/dist_train.py:63)")
            if shard.owned_mask.device != pos.device:
                shard.to(pos.device)
            # ONE rank that neither sends nor receives a row (a plain world-1 plan): nothing to exchange with anybody -- the
            # collectives (and the stream hand-offs they cost: ~0.12 ms per step, round 4) are skipped and the layers take
            # the unsharded forms (deferred sums).  With peers every rank joins every collective, rows or not.
            lone = shard.world == 1 and shard.atom_plan.send_idx.numel() == 0 and shard.atom_plan.recv_idx.numel() == 0
            data._hn_lone = lone
            if lone:
                pass
            elif shard.halo_pos_local:     # coordinates are there; their gradients still go home
                pos = HaloGradReturn.apply(pos, shard.atom_plan)
            else:
                pos = HaloExchange.apply(pos, shard.atom_plan)              # halo coordinates from their owners
            if not lone:
                row_plan = shard.row_plan(graph.row_of_node)
        # hermnet.py:123, row order (pads: Z=0).  eval(): every parameter is a constant, the embedding included
        # (the fused layers produce no parameter gradients; a partial set would be worse than none)
        if train and pos.is_cuda:
            from .trainops import embedding_rows
            x = embedding_rows(self.embed.weight, graph.z_rows.long())
        else:
            x = self.embed(graph.z_rows) if train else torch.nn.functional.embedding(graph.z_rows, self.embed.weight.detach())
        H = self.hidden_channels
        Hp = (H + 63) // 64 * 64 if (fused and switches.fused_layer) else H
        if Hp != H:
            # widths that are not a multiple of 64 run on the same kernels with zero-padded channels (layer.LayerWeights)
            x = torch.nn.functional.pad(x, (0, Hp - H))
        data._hn_pre0 = None
        data._hn_weights = None
        if fused and switches.fused_layer:
            data._hn_weights = self._refresh_weights(pos.device)
        if train:
            edge, edge_parts = self._edge_geometry_autograd(pos, data.get("cell"), graph)
        else:
            edge = EdgeGeometry.apply(pos, data.get("cell"), graph)      # with_edge, hermnet.py:133-152

        vec = None                                                          # zeros, hermnet.py:124
        if not fused and not train:
            edge = TrueEdgeGradient.apply(edge)      # autograd's (rhat, d) gradients -> Cartesian for the kernel
        data._hn_graph, data._hn_edge, data._hn_rbf = graph, edge, rbf
        if fused and edge.requires_grad and edge.is_cuda:
            # forces wanted: the backward message kernels read the radial quantities of an edge from this table
            graph.edge_table = edge_radial_table(graph, rbf, edge.detach())
        # rmnet.py:168-172 for the optional bases only: [E,R] basis from the kernel's distances
        if fused:
            data._hn_edge_embed = None
        elif train and pos.is_cuda and self.radial_basis.rbf_name == "gaussian" and H % 4 == 0 and graph.T > 0:
            # train(): the basis sorted by (relation, distance bucket) and cut to the bucket's 32 centres (trainops.BucketedBasis)
            parts = edge_parts
            Ek = graph.rel_edge_total()              # (no host read when every atom has an element of the model)
            data._hn_edge_embed = self.radial_basis.bucketed(edge[:, 3] if parts is None else parts[1], Ek, graph.T,
                                                             graph.rel_edge_bounds_dev())
            if parts is not None:
                data._hn_edge_embed.set_unit_vectors(edge, parts[0], Ek)
        else:
            data._hn_edge_embed = self.radial_basis(edge[:, 3])
        data.x, data.vec = x, vec
        data._hn_edge_handles = data._hn_edge_sink = data._hn_halo = data._hn_chain_node = None
        if fused and edge.requires_grad and switches.fused_layer:
            # one reduction of the edge gradients per step instead of one per layer (layer.EdgeFanout)
            # (atoms of an unknown element own the rows past type_rowptr[T]; only edges INTO them go unwritten)
            # (the slots of a padded list's NULL edges stay unwritten too, but nothing reads them: the position gradient
            # walks the CSC segments, and NULL edges are in none)
            all_known = graph.num_src == 0 and graph.N == graph.type_rowptr_host[-1]
            data._hn_edge_sink = EdgeGradSink(len(self.hermconvs), Hp // 64, graph.E, pos.device, zero=not all_known)
            data._hn_edge_handles = EdgeFanout.apply(edge, data._hn_edge_sink)
        _PENDING.clear()                      # (leftovers of a backward pass that did not complete)
        _PRE_NEXT.clear()
        for li, conv in enumerate(self.hermconvs):
            data._hn_layer = li
            data = conv(data)
            if row_plan is not None and li + 1 < len(self.hermconvs):
                # one exchange per layer: (x | vec) rows of halo atoms, 4H floats each -- due before the next layer reads
                # them, run BY that layer (overlapped with its node projection where it can, HeteroVertexConv.forward)
                data._hn_halo = shard.halo_overlap(graph, nodeops_tile_rows(Hp))
        graph._keep = None
        x = data.x
        if Hp != H:
            x = x[:, :H]                                                    # the read-out sees the real channels

        head_params = (self.out_energy[0].weight, self.out_energy[0].bias, self.out_energy[2].weight,
                       self.out_energy[2].bias)
        single = shard is None and graph.num_graphs == 1
        # one structure on an atom shard (the sharded MD case): the read-out masks the rows of halo atoms like it masks padding
        # rows, the rank's share is ONE ordered sum over its rows -- no gather back to atom order, no mask product, no count
        # (round 6: ~10 small launches per step less, forward and backward)
        shard_rows = None
        if (shard is not None and shard.num_graphs == 1 and not train and fused
                and switches.fused_layer and (self.hidden_channels // 2) % 4 == 0):
            shard_rows = shard.owned_rows(graph, x.size(0))
        if train or not fused or (self.hidden_channels // 2) % 4 != 0 or not switches.fused_layer:
            if train and x.is_cuda:
                # (the same nn.Sequential, its two Linears through trainops.TallBmm: weight gradients over ~2e4 rows)
                from .trainops import tall_linear
                h_ = tall_linear(x, self.out_energy[0].weight, self.out_energy[0].bias)
                e_rows = tall_linear(self.out_energy[1](h_), self.out_energy[2].weight, self.out_energy[2].bias).squeeze(1)
            else:
                e_rows = self.out_energy(x).squeeze(1)                      # hermnet.py:129, row order
            if single:
                e_rows = e_rows * graph.row_real
        else:
            e_rows = EnergyHead.apply(x.contiguous(), *[p.detach() for p in head_params],
                                      graph.row_real if single else shard_rows)
        if shard_rows is not None:
            energy = e_rows.sum().reshape(1)
            if shard.world > 1:
                energy = SumAcrossRanks.apply(energy, shard.group)
            if self.intensive:               # (mean over the atoms of the WHOLE structure: every rank knows their number)
                energy = energy / max(shard.num_atoms_global, 1)
            return energy
        if single:
            # one graph: padding rows are masked instead of gathering back to atom order (the gather's
            # backward is an index_put, ~50 us); fixed summation order, bit-reproducible
            energy = e_rows.sum().reshape(1)
            if self.intensive:
                energy = energy / max(graph.num_atoms, 1)
            return energy
        batch = data.batch.long()
        if shard is None and graph.graph_perm is not None:
            # scatter(per_atom_energy, batch, reduce=sum|mean) (hermnet.py:130) as an ordered segment reduction: no atomics,
            # so the energies are bit-reproducible run to run
            energy = GraphEnergies.apply(e_rows, graph, batch)
            if self.intensive:
                energy = energy / graph.graph_lengths.clamp(min=1).to(energy.dtype)
            return energy
        per_atom_energy = e_rows.index_select(0, graph.row_of_node)         # back in atom order
        if shard is not None:
            own = shard.owned_mask.to(per_atom_energy.dtype)
            e_own = per_atom_energy * own
            if shard.num_graphs == 1:      # one structure (the sharded MD case): ordered sums, bit-reproducible
                energy, cnt = e_own.sum().reshape(1), own.sum().reshape(1)
            else:
                # several graphs on a shard: the local atoms sorted by graph (stable, cached with the plan), then ORDERED
                # segment sums -- no index_add atomics, so a rank's share is bit-reproducible run to run like everything else
                perm, lengths = shard.graph_order(batch)
                energy = torch.segment_reduce(e_own.index_select(0, perm), "sum", lengths=lengths, unsafe=True)
                cnt = torch.segment_reduce(own.index_select(0, perm), "sum", lengths=lengths, unsafe=True)
            if shard.world > 1:
                energy, cnt = SumAcrossRanks.apply(energy, shard.group), (SumAcrossRanks.apply(cnt, shard.group) if self.intensive else cnt)
            if self.intensive:
                energy = energy / cnt.clamp(min=1)
            return energy
        # (train(): segment_reduce has no second derivative)
        energy = torch.zeros(graph.num_graphs, dtype=x.dtype, device=x.device).index_add(0, batch, per_atom_energy)
        if self.intensive:
            energy = energy / graph.graph_lengths.clamp(min=1).to(energy.dtype)
        return energy


def triadic_relations(elems):
    """[(module key, centre element, (p, q))] in module order: centres in `elems` order, unordered neighbour pairs
    p <= q in `elems` order -- T * T(T+1)/2 relations (figs/subgraph.svg (c): "A->A<-A", "B->A<-C")."""
    out = []
    for c in elems:
        for i, p in enumerate(elems):
            for q in elems[i:]:
                out.append(("%s_%s-%s" % (c, p, q), c, (p, q)))
    return out


class HeteroTriadicConv(HeteroVertexConv):
    """HTNet's layer: the loop of `hermnet.py:51-61` over triadic relations.  Relation (c; p, q) owns a PaiNNModule
    and sees the edges j -> i with element(i) = c and element(j) in {p, q}; a centre's T(T+1)/2 results are
    averaged.  All relations run in the same fused launches as HVNet's: the graph has one TARGET row per atom and
    pair relation (relations.build_triadic), the kernels gather from the atom-ordered SOURCE rows."""

    def forward(self, data):
        data = super().forward(data)                   # (x, vec) in target rows [T * P * B, ...]
        g = data._hn_graph
        P_, B_, Te = g.triadic_pairs, g.block, g.T // g.triadic_pairs
        if data.get("_hn_edge_embed") is not None:     # train(): plain torch ops, differentiable to any order
            H = data.x.size(1)
            xo = data.x.view(Te, P_, B_, H).mean(1).reshape(Te * B_, H)
            vo = data.vec.view(Te, P_, B_, 3, H).mean(1).reshape(Te * B_, 3, H)
            if g.num_src > Te * B_:                    # atoms of elements outside `elems`: zero rows (hermnet.py:51)
                xo = torch.cat([xo, xo.new_zeros(g.num_src - Te * B_, H)], 0)
                vo = torch.cat([vo, vo.new_zeros(g.num_src - Te * B_, 3, H)], 0)
            data.x, data.vec = xo, vo
        else:
            data.x, data.vec = PairMean.apply(data.x, data.vec, Te, P_, B_, g.num_src)
        return data


class PairMean(torch.autograd.Function):
    """(x, vec) of the [Te][P][B] virtual target rows -> their mean over P in source rows [num_src, ...]; atoms of
    elements outside `elems` (rows behind Te*B) get zero rows (hermnet.py:51).  One launch each way on the GPU."""

    @staticmethod
    def forward(ctx, x, vec, Te, P_, B_, rows_out):
        ctx.dims = (Te, P_, B_, rows_out)
        if x.is_cuda:
            from . import nodeops
            return nodeops.pair_mean(x.contiguous(), vec.contiguous(), Te, P_, B_, rows_out)
        H = x.size(1)
        xo = x.new_zeros(rows_out, H)
        vo = vec.new_zeros(rows_out, 3, H)
        xo[:Te * B_] = x.view(Te, P_, B_, H).mean(1).reshape(Te * B_, H)
        vo[:Te * B_] = vec.view(Te, P_, B_, 3, H).mean(1).reshape(Te * B_, 3, H)
        return xo, vo

    @staticmethod
    def backward(ctx, gx, gvec):
        Te, P_, B_, rows_out = ctx.dims
        if gx.is_cuda:
            from . import nodeops
            gxi, gvi = nodeops.pair_mean(gx.contiguous(), gvec.contiguous(), Te, P_, B_, rows_out, backward=True)
            return gxi, gvi, None, None, None, None
        H = gx.size(1)
        gxi = (gx[:Te * B_].view(Te, 1, B_, H) / P_).expand(Te, P_, B_, H).reshape(Te * P_ * B_, H)
        gvi = (gvec[:Te * B_].view(Te, 1, B_, 3, H) / P_).expand(Te, P_, B_, 3, H).reshape(Te * P_ * B_, 3, H)
        return gxi, gvi, None, None, None, None


class HTNet(HVNet):
    """Heterogeneous Triadic Network (`README.md:27`, `figs/subgraph.svg` (c)).  The reference's class is a stub that
    raises NotImplementedError (`hermnet.py:155-157`); this is the BUILD-DEFINED model specified in DESIGN.md
    ("HTNet"), with HVNet's constructor and `forward(data)`: one PaiNNModule per (centre element, unordered pair of
    neighbour elements), `hermconvs.{l}.mods.{c}_{p}-{q}.*` in the state_dict.  With a single element it IS HVNet.
    Parity is checked against `oracle.htnet_energy` ("parity unpinned")."""

    def __init__(self, elems: Union[str, List[str]], rc: float = 5., intensive: bool = False, num_layers: int = 5,
                 hidden_channels: int = 512, num_rbf: int = 128, rbf={"name": "gaussian"},
                 envelope={"name": "polynomial", "exponent": 5}):
        super().__init__(elems, rc=rc, intensive=intensive, num_layers=0, hidden_channels=hidden_channels,
                         num_rbf=num_rbf, rbf=rbf, envelope=envelope)
        self.num_layers = num_layers
        keys = [k for k, _, _ in triadic_relations(self.elems)]
        for _ in range(num_layers):
            self.hermconvs.append(HeteroTriadicConv(
                mods={k: PaiNNModule(hidden_channels=hidden_channels, num_rbf=num_rbf) for k in keys}))

    def _build_graph(self, data, zl, shard):
        if data.get("_hn_edge_count") is not None:
            raise NotImplementedError("HTNet's triadic relation build takes exact neighbour lists")
        return RelationalGraph.build_triadic(data.atomic_number, data.edge_index, zl,
                                             edge_shift=data.get("edge_shift") if data.get("cell") is not None else None,
                                             batch=data.batch,
                                             rel_active=None if shard is None else shard.rel_active_triadic(zl))

    def forward(self, data):
        """eval(): the fused kernels (Gaussian basis).  train(): the differentiable path of HVNet on the triadic graph --
        every parameter of the 18 (T * P) PaiNNModules per layer gets its gradient, forces included in the loss."""
        if not self.radial_basis.fused:
            raise NotImplementedError("HTNet runs on the fused kernels: Gaussian radial basis")
        return super().forward(data)
