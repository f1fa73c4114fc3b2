"""Relation-ordered graph: the build's replacement for `in_subgraph`
(`HermNet/utils.py:11-24`, called per relation per layer at `hermnet.py:52-54`).

Instead of N_t O(E) scans per relation per layer, edges are sorted ONCE per
neighbour list: atoms are renumbered by (relation of their element, id) and the
edge list is kept in two orders,
  CSR  (row(target), edge id)                      -> forward segmented sums,
  CSC  (relation(target), row(source), CSR pos)    -> backward segmented sums,
plus the out-adjacency by source row for the position gradient.  All index
arrays are int32 device tensors.  One small D2H copy (T+1 row offsets and T edge
counts) tells the host where each relation's rows start.
"""
import os

import torch

from . import switches

_COUNT_CACHE = []      # most recent first: (z tensor, version, batch tensor, version, z_list, result)
_CONST_CACHE = {}


def _cached_i32(values, dev):
    """Small constant index tensors are uploaded once (a pageable H2D copy per step would also break
    hipGraph capture of the step)."""
    key = ("i32", values, str(dev))
    t = _CONST_CACHE.get(key)
    if t is None:
        if len(_CONST_CACHE) > 64:
            _CONST_CACHE.clear()
        t = _CONST_CACHE[key] = torch.tensor(list(values), dtype=torch.int32, device=dev)
    return t


def _cached_u8(values, dev):
    key = ("u8", values, str(dev))
    t = _CONST_CACHE.get(key)
    if t is None:
        t = _CONST_CACHE[key] = torch.tensor(list(values), dtype=torch.uint8, device=dev)
    return t


class RelationalGraph(object):
    """Rows = atoms in relation order.  With `uniform` layout every relation owns a block of
    `block` = max_t N_t rows (short relations are padded with inert rows), so the per-relation node
    GEMMs of a layer are ONE batched GEMM over a [T, block, .] view; atoms of unknown element follow
    after the T blocks.  Without it (very unbalanced compositions) blocks are tight and the host
    loops over relations."""

    __slots__ = ("N", "E", "T", "num_atoms", "uniform", "block", "node_order", "row_of_node", "z_rows",
                 "type_rowptr", "type_rowptr_host", "csr_rowptr", "csr_src", "csr_perm", "csc_rowptr", "csc_tgt",
                 "csc_pos", "out_rowptr", "out_edges", "src_id", "tgt_id", "shift", "row_active", "row_real",
                 "batch_rows", "batch32", "num_graphs", "graph_perm", "graph_lengths", "device", "_cstruct", "_rel_bounds",
                 "edge_table", "num_src", "res_row", "triadic_pairs", "src_real", "_rowptr_c", "src_ranges", "_row_keys",
                 "_upd_tile", "ready", "_keep", "_edge_atoms64", "_edge_sum_keys", "_row_graph", "_all_known")

    def __init__(self):
        self._cstruct = None
        self._rel_bounds = None
        self._all_known = False    # every atom has an element of the model (host knowledge from the atom counts): Ek == E
        self._rowptr_c = None      # type_rowptr_host as a ctypes int array (nodeops._rowptr_host)
        self._upd_tile = None      # (H, tile rows of the update kernels for this row layout) (nodeops.update_tile_rows)
        self._row_keys = None      # gather / segmented-sum keys of the differentiable path (trainops._row_keys)
        self.edge_table = None     # [E+1,32] per-edge radial records of the current step, CSC order (set by HVNet.forward)
        self.num_src = 0           # separate source-row space (HTNet): rows of xh / vec; 0 = same rows as the targets
        self.res_row = None        # [N] int32 source row feeding the residual of each target row, or None
        self.triadic_pairs = 0     # HTNet: pair relations per centre element (target rows = T_elem * pairs * block)
        self.src_real = None       # HTNet: [num_src] 1 for source rows that hold an atom
        self.src_ranges = None     # HTNet: [T,4] int32, the two source-row ranges a relation gathers from (nodeops.node_pre_fwd)
        self._edge_atoms64 = None  # trainops.EdgeDiff: (source, target) atom of every edge as int64
        self._edge_sum_keys = None
        self._row_graph = None     # hermnet.GraphEnergies: graph index of every row
        self._keep = None

    def rel_edge_bounds(self):
        """CSR edge ranges of the relations: edges of relation t are [b[t], b[t+1]) (rows are relation-ordered and
        CSR is row-ordered); edges to unknown-element targets follow after b[T].  One host sync per graph; used by
        the differentiable device-op path only."""
        if self._rel_bounds is None:
            self._rel_bounds = self.csr_rowptr[self.type_rowptr.long()].tolist()
        return self._rel_bounds

    def rel_edge_total(self):
        """Number of edges whose target has a known element (= rel_edge_bounds()[T]) WITHOUT a host read where the host
        already knows it: with no atom of an unknown element every edge counts (the train() path sizes its arrays by it; a
        read-back here was the last device sync of the training step after the relation build)."""
        if self._all_known:
            return self.E
        return self.rel_edge_bounds()[self.T]

    def rel_edge_bounds_dev(self):
        """`rel_edge_bounds` as a device tensor [T + 1] (no host read, no upload)."""
        return self.csr_rowptr[self.type_rowptr.long()]

    @staticmethod
    def build(atomic_number, edge_index, z_list, edge_shift=None, batch=None, rel_active=None, uniform=None):
        """atomic_number [N] int, edge_index [2,E] int (row 0 = source, row 1 = target,
        `hermnet.py:135`), z_list: atomic numbers of the model's elements in module order.
        `uniform`: None = automatic (padding overhead <= 15 %), True/False = forced.

        GPU tensors take the device-side build (`csrc/relation_kernels.hip`); host tensors (tests,
        planning) take the PyTorch restatement below, which defines the expected result bit for bit.
        (`switches.native_relations = False`: the restatement for GPU tensors too -- tests.)"""
        if atomic_number.is_cuda and switches.native_relations:
            return RelationalGraph._build_native(atomic_number, edge_index, z_list, edge_shift, batch, rel_active, uniform)
        return RelationalGraph._build_torch(atomic_number, edge_index, z_list, edge_shift, batch, rel_active, uniform)

    @staticmethod
    def _layout(cnt_host, T, uniform):
        known = sum(cnt_host[:T])
        block = max(cnt_host[:T]) if T > 0 else 0
        if uniform is None:
            uniform = T > 1 and T * block <= 1.15 * known + 64
        uniform = bool(uniform) or T == 1
        if uniform:
            starts = [t * block for t in range(T)] + [T * block]
        else:
            starts, block = [0], 0
            for t in range(T):
                starts.append(starts[-1] + cnt_host[t])
        return uniform, block, starts, starts[T] + cnt_host[T]

    @staticmethod
    def _atom_counts(atomic_number, batch, z_list):
        """(z_list on the device, atoms per relation + unknown [T+1] on the host, number of graphs, atomic numbers as
        int64, {row-layout cache}) of these atoms: a host sync, skipped while the same tensors are passed again (atom
        types and batch assignment do not change along an MD trajectory).  Keyed on the identity of the caller's
        tensor OBJECTS, which the cache keeps alive: an address alone could be reused by a different tensor of the
        same size."""
        from . import _lib
        lib = _lib.load()
        dev = atomic_number.device
        NA, T = int(atomic_number.numel()), len(z_list)
        i32, P = torch.int32, _lib.ptr
        for ent in _COUNT_CACHE:
            if (ent[0] is atomic_number and ent[1] == atomic_number._version and ent[2] is batch
                    and ent[3] == (None if batch is None else batch._version) and ent[4] == tuple(z_list)):
                return ent[5]
        from .ops import _stream
        z = atomic_number.long().contiguous()
        zl = torch.tensor(list(z_list), dtype=i32, device=dev)
        counts = torch.empty(T + 1, dtype=i32, device=dev)
        _lib.check(lib.hermnet_relation_counts(P(z), NA, P(zl), T, P(counts), _stream()), "hermnet_relation_counts")
        nb = batch.long().max().reshape(1) + 1 if (batch is not None and NA > 0) else torch.ones(1, dtype=torch.long, device=dev)
        host = torch.cat([counts.long(), nb]).cpu().tolist()
        hit = (zl, host[:T + 1], int(host[-1]), z, {})
        _COUNT_CACHE.insert(0, (atomic_number, atomic_number._version, batch,
                                None if batch is None else batch._version, tuple(z_list), hit))
        del _COUNT_CACHE[8:]
        return hit

    @staticmethod
    def _build_native(atomic_number, edge_index, z_list, edge_shift, batch, rel_active, uniform):
        import ctypes
        from . import _lib
        from .ops import _stream
        lib = _lib.load()
        g = RelationalGraph()
        dev = atomic_number.device
        NA, E, T = int(atomic_number.numel()), int(edge_index.size(1)), len(z_list)
        g.num_atoms, g.E, g.T, g.device = NA, E, T, dev
        ei = edge_index.long().contiguous()
        i32, P = torch.int32, _lib.ptr
        hit = RelationalGraph._atom_counts(atomic_number, batch, z_list)
        zl, cnt_host, g.num_graphs, z = hit[:4]
        rows_cache = hit[4]                   # {(uniform layout key): row arrays}: they depend on the atoms only
        g.uniform, g.block, starts, N = RelationalGraph._layout(cnt_host, T, uniform)
        g.N, g.type_rowptr_host = N, starts[:T + 1]
        g._all_known = int(cnt_host[T]) == 0
        g.type_rowptr = _cached_i32(tuple(starts[:T + 1]), dev)
        e32 = lambda n: torch.empty(n, dtype=i32, device=dev)
        rows = rows_cache.get((g.uniform, N))
        rows_ready = rows is not None
        if rows is None:
            rows = dict(node_order=e32(NA), row_of_node=e32(NA), z_rows=e32(N),
                        row_real=torch.empty(N, dtype=torch.float32, device=dev))
        g.row_real = rows["row_real"]
        g.row_active = torch.empty(N, dtype=torch.float32, device=dev)
        g.csr_rowptr, g.csr_src, g.csr_perm, g.src_id, g.tgt_id = e32(N + 1), e32(E), e32(E), e32(E), e32(E)
        g.csc_rowptr, g.csc_tgt, g.csc_pos = e32(T * N + 1), e32(E), e32(E)
        # (no out-adjacency: the position gradient reads the out-edges from the CSC order, ops.EdgeGeometry)
        g.out_rowptr = g.out_edges = None
        shift = None if edge_shift is None else edge_shift.float().contiguous()
        g.shift = None if shift is None else torch.empty(E, 3, dtype=torch.float32, device=dev)
        if torch.is_tensor(rel_active):       # device flags (slab plans, sharding.py): no host read
            act = rel_active.to(device=dev, dtype=torch.uint8).contiguous()
        else:
            act = None if rel_active is None else _cached_u8(tuple(bool(a) for a in rel_active), dev)
        wbytes = lib.hermnet_build_relations_workspace(NA, N, E, T)
        work = torch.empty(wbytes, dtype=torch.uint8, device=dev)
        out = _lib.RelationsOut(P(rows["node_order"]), P(rows["row_of_node"]), P(rows["z_rows"]), P(g.row_real), P(g.row_active),
                                P(g.csr_rowptr), P(g.csr_src), P(g.csr_perm), P(g.src_id), P(g.tgt_id), P(g.shift),
                                P(g.csc_rowptr), P(g.csc_tgt), P(g.csc_pos), P(g.out_rowptr), P(g.out_edges))
        _lib.check(lib.hermnet_build_relations(P(z), P(ei), P(shift), NA, E, P(zl), T, P(g.type_rowptr), N, P(act), ctypes.byref(out),
                                               1 if rows_ready else 0, P(work), wbytes, _stream()), "hermnet_build_relations")
        if not rows_ready:
            # int64 copies for the host code's gathers (embedding, index_select), made once per atom set
            rows["z_rows64"] = rows["z_rows"].long()
            rows["row_of_node64"] = rows["row_of_node"].long()
            rows["node_order64"] = rows["node_order"].long()
            rows["batch32"] = None if batch is None else batch.to(i32).contiguous()
            if len(rows_cache) > 4:
                rows_cache.clear()
            rows_cache[(g.uniform, N)] = rows
        g.z_rows, g.row_of_node, g.node_order = rows["z_rows64"], rows["row_of_node64"], rows["node_order64"]
        g.batch32 = rows["batch32"]
        g.batch_rows = None
        if batch is not None and g.num_graphs > 1:
            if "graph_perm" not in rows:          # (depends on the batch vector only, like the row layout: once per atom set)
                b64 = batch.long()
                rows["graph_perm"] = torch.argsort(b64, stable=True)
                rows["graph_lengths"] = torch.zeros(g.num_graphs, dtype=torch.long, device=dev).index_add_(
                    0, b64, torch.ones_like(b64))
            g.graph_perm, g.graph_lengths = rows["graph_perm"], rows["graph_lengths"]
        else:
            g.graph_perm = None
            g.graph_lengths = None
        return g

    @staticmethod
    def _build_torch(atomic_number, edge_index, z_list, edge_shift=None, batch=None, rel_active=None, uniform=None):
        g = RelationalGraph()
        dev = atomic_number.device
        NA = int(atomic_number.numel())
        E = int(edge_index.size(1))
        T = len(z_list)
        g.num_atoms, g.E, g.T, g.device = NA, E, T, dev
        z = atomic_number.long()
        zl = torch.tensor(list(z_list), dtype=torch.long, device=dev)
        eq = z[:, None] == zl[None, :]
        # relation of each atom: first matching element, T for "not in elems" (hermnet.py:53 finds none)
        rel = torch.where(eq.any(1), eq.int().argmax(1), torch.full((NA,), T, dtype=torch.long, device=dev))
        counts = torch.zeros(T + 1, dtype=torch.long, device=dev).index_add_(0, rel, torch.ones_like(rel))
        nb = batch.long().max().reshape(1) + 1 if (batch is not None and NA > 0) else torch.ones(1, dtype=torch.long, device=dev)
        host = torch.cat([counts, nb]).cpu().tolist()          # the one host sync of the build
        cnt_host, g.num_graphs = host[:T + 1], int(host[-1])
        g.uniform, g.block, starts, N = RelationalGraph._layout(cnt_host, T, uniform)
        g.N = N                                                 # rows (>= atoms when padded)
        g.type_rowptr_host = starts[:T + 1]

        ar = torch.arange(NA, device=dev)
        g.node_order = torch.sort(rel.to(torch.int32), stable=True).indices   # atoms sorted by (relation, id)
        first_sorted = torch.zeros(T + 1, dtype=torch.long, device=dev)
        first_sorted[1:] = torch.cumsum(counts[:T], 0)          # position of each relation in the sorted list
        starts_d = torch.tensor(starts, dtype=torch.long, device=dev)
        rel_sorted = rel[g.node_order]
        rows_sorted = ar - first_sorted[rel_sorted] + starts_d[rel_sorted]
        g.row_of_node = torch.empty_like(g.node_order)
        g.row_of_node[g.node_order] = rows_sorted
        g.z_rows = torch.zeros(N, dtype=torch.long, device=dev)
        g.z_rows[g.row_of_node] = z
        g.row_real = torch.zeros(N, dtype=torch.float32, device=dev)
        g.row_real[g.row_of_node] = 1.0
        rel_row = torch.full((N,), T, dtype=torch.long, device=dev)
        for t in range(T):
            rel_row[starts[t]:(starts[t + 1] if t + 1 <= T else N)] = t
        rel_row[starts[T]:] = T

        def rowptr_of(sorted_keys, nkeys):
            """CSR row pointer of an ascending key list (no host sync, unlike bincount)."""
            return torch.searchsorted(sorted_keys, torch.arange(nkeys + 1, device=dev))

        src, tgt = edge_index[0].long(), edge_index[1].long()
        rs, rt = g.row_of_node[src], g.row_of_node[tgt]
        i32k = torch.int32 if (T + 1) * N < 2 ** 31 else torch.long   # narrow keys: half the radix passes
        rt_s, csr_perm = torch.sort(rt.to(i32k), stable=True)
        rt_s = rt_s.long()
        csr_src = rs[csr_perm]
        csr_rowptr = rowptr_of(rt_s, N)

        key2 = rel_row[rt_s] * N + csr_src                      # (relation(target), row(source))
        key2_s, csc_pos = torch.sort(key2.to(i32k), stable=True)
        key2_s = key2_s.long()
        csc_tgt = rt_s[csc_pos]
        csc_rowptr = rowptr_of(key2_s, T * N)

        src_s, out_edges = torch.sort(csr_src.to(i32k), stable=True)
        src_s = src_s.long()
        out_rowptr = rowptr_of(src_s, N)

        # hermnet.py:56-57: a relation without edges is skipped -> its rows stay zero; rows of
        # unknown-type atoms (hermnet.py:51) and padding rows are zero as well.
        # (`rel_active`: the caller knows better, e.g. a shard whose relation has edges on other ranks only)
        if rel_active is None:
            tn = torch.arange(T + 1, device=dev) * N
            act = torch.cat([(csc_rowptr[tn[1:]] - csc_rowptr[tn[:-1]]) > 0,
                             torch.zeros(1, dtype=torch.bool, device=dev)])
        elif torch.is_tensor(rel_active):
            act = torch.cat([rel_active.to(dev).bool(), torch.zeros(1, dtype=torch.bool, device=dev)])
        else:
            act = torch.tensor([bool(a) for a in rel_active] + [False], dtype=torch.bool, device=dev)
        g.row_active = act[rel_row].float() * g.row_real

        i32 = torch.int32
        g.type_rowptr = starts_d[:T + 1].to(i32)
        g.csr_rowptr = csr_rowptr.to(i32)
        g.csr_src = csr_src.to(i32)
        g.csr_perm = csr_perm
        g.csc_rowptr = csc_rowptr.to(i32)
        g.csc_tgt = csc_tgt.to(i32)
        g.csc_pos = csc_pos.to(i32)
        g.out_rowptr = out_rowptr.to(i32)
        g.out_edges = out_edges.to(i32)
        g.src_id = src[csr_perm].to(i32)
        g.tgt_id = tgt[csr_perm].to(i32)
        g.shift = None if edge_shift is None else edge_shift[csr_perm].float().contiguous()
        g.batch32 = None if batch is None else batch.to(i32).contiguous()
        g.batch_rows = None
        if batch is not None:
            g.batch_rows = torch.zeros(N, dtype=torch.long, device=dev)
            g.batch_rows[g.row_of_node] = batch.long()
        # deterministic per-graph read-out: atoms grouped by graph (stable) + segment lengths
        if batch is not None and g.num_graphs > 1:
            g.graph_perm = torch.argsort(batch.long(), stable=True)
            g.graph_lengths = torch.zeros(g.num_graphs, dtype=torch.long, device=dev).index_add_(
                0, batch.long(), torch.ones_like(batch.long()))
        else:
            g.graph_perm = None
            g.graph_lengths = None
        return g

    @staticmethod
    def _build_triadic_native(atomic_number, edge_index, z_list, edge_shift, batch, rel_active=None):
        """`build_triadic` through `hermnet_build_triadic` (csrc/relation_kernels.hip); None when an atom is of an
        element outside `z_list` (the torch build handles those)."""
        import ctypes
        from . import _lib
        from .ops import _stream
        lib = _lib.load()
        dev = atomic_number.device
        NA, E0, T = int(atomic_number.numel()), int(edge_index.size(1)), len(z_list)
        zl, cnt_host, num_graphs, z, rows_cache = RelationalGraph._atom_counts(atomic_number, batch, z_list)
        if cnt_host[T] != 0:
            return None
        Pn = T * (T + 1) // 2
        TR, B = T * Pn, max(cnt_host[:T])
        Ns, Nt, E = T * B, TR * B, T * E0
        if TR * Ns + Nt + 8 >= 2 ** 31 or E >= 2 ** 31:
            return None
        g = RelationalGraph()
        g.num_atoms, g.T, g.device, g.uniform, g.block, g.num_graphs = NA, TR, dev, True, B, num_graphs
        g.N, g.num_src, g.triadic_pairs, g.E = Nt, Ns, Pn, E
        g.type_rowptr_host = [r * B for r in range(TR + 1)]
        g.type_rowptr = _cached_i32(tuple(g.type_rowptr_host), dev)
        i32, P = torch.int32, _lib.ptr
        e32 = lambda n: torch.empty(n, dtype=i32, device=dev)
        rows = rows_cache.get(("triadic", B))
        rows_ready = rows is not None
        if rows is None:
            rows = dict(node_order=e32(NA), row_of_node=e32(NA), z_rows=e32(Ns),
                        src_real=torch.empty(Ns, dtype=torch.float32, device=dev))
        g.src_real = g.row_real = rows["src_real"]                 # the energy read-out masks SOURCE rows
        g.row_active = torch.empty(Nt, dtype=torch.float32, device=dev)
        tgt_real = torch.empty(Nt, dtype=torch.float32, device=dev)
        g.res_row = e32(Nt)
        g.csr_rowptr, g.csr_src, g.csr_perm, g.src_id, g.tgt_id = e32(Nt + 1), e32(E), e32(E), e32(E), e32(E)
        g.csc_rowptr, g.csc_tgt, g.csc_pos = e32(TR * Ns + 1), e32(E), e32(E)
        g.out_rowptr = g.out_edges = None
        ei = edge_index.long().contiguous()
        shift = None if edge_shift is None else edge_shift.float().contiguous()
        g.shift = None if shift is None else torch.empty(E, 3, dtype=torch.float32, device=dev)
        counts_d = _cached_i32(tuple(cnt_host[:T]), dev)
        wbytes = lib.hermnet_build_triadic_workspace(NA, E0, T, B)
        work = torch.empty(wbytes, dtype=torch.uint8, device=dev)
        out = _lib.RelationsOut(P(rows["node_order"]), P(rows["row_of_node"]), P(rows["z_rows"]), P(rows["src_real"]),
                                P(g.row_active), P(g.csr_rowptr), P(g.csr_src), P(g.csr_perm), P(g.src_id), P(g.tgt_id),
                                P(g.shift), P(g.csc_rowptr), P(g.csc_tgt), P(g.csc_pos), None, None)
        if torch.is_tensor(rel_active):
            act = rel_active.to(device=dev, dtype=torch.uint8).contiguous()
        else:
            act = None if rel_active is None else _cached_u8(tuple(bool(a) for a in rel_active), dev)
        _lib.check(lib.hermnet_build_triadic(P(z), P(ei), P(shift), NA, E0, P(zl), T, B, P(counts_d), P(act), ctypes.byref(out),
                                             P(tgt_real), P(g.res_row), 1 if rows_ready else 0, P(work), wbytes, _stream()),
                   "hermnet_build_triadic")
        if not rows_ready:
            rows["z_rows64"] = rows["z_rows"].long()
            rows["row_of_node64"] = rows["row_of_node"].long()
            rows["node_order64"] = rows["node_order"].long()
            rows["batch32"] = None if batch is None else batch.to(i32).contiguous()
            if len(rows_cache) > 4:
                rows_cache.clear()
            rows_cache[("triadic", B)] = rows
        g.z_rows, g.row_of_node, g.node_order = rows["z_rows64"], rows["row_of_node64"], rows["node_order64"]
        g.batch32 = rows["batch32"]
        g.batch_rows = None
        pairs = [(p, q) for p in range(T) for q in range(p, T)]
        g.src_ranges = _cached_i32(tuple(v for _c in range(T) for (p_, q_) in pairs
                                         for v in (p_ * B, p_ * B + cnt_host[p_], (q_ * B if q_ != p_ else 0),
                                                   (q_ * B + cnt_host[q_] if q_ != p_ else 0))), dev).view(TR, 4)
        if batch is not None and g.num_graphs > 1:
            g.graph_perm = torch.argsort(batch.long(), stable=True)
            g.graph_lengths = torch.zeros(g.num_graphs, dtype=torch.long, device=dev).index_add_(
                0, batch.long(), torch.ones_like(batch.long()))
        else:
            g.graph_perm = None
            g.graph_lengths = None
        return g

    @staticmethod
    def build_triadic(atomic_number, edge_index, z_list, edge_shift=None, batch=None, rel_active=None):
        """HTNet's relation-ordered graph (DESIGN.md "HTNet"): relation rho = (centre element c, unordered pair
        {p, q} of neighbour elements), T * T(T+1)/2 of them.

        Two row spaces.  SOURCE rows: the atoms ordered by (element, id), every element padded to `block` rows (the
        HVNet order); unknown elements behind.  TARGET rows: one block of `block` rows per relation, block (c, k)
        holding the atoms of element c in the same order -- an atom is a target once per pair relation of its
        element ("virtual" rows; `res_row` maps them back to the atom's source row).  A directed edge j -> i appears
        once for every pair that contains element(j), i.e. T times: CSR by target row, CSC by (relation, SOURCE row),
        which is exactly the layout the message kernels consume with `num_src` / `res_row` set.

        `rel_active` [T P] (list or device tensor): overrides "a relation runs iff it receives an edge here" (atom shards).

        GPU tensors whose atoms are all of listed elements take the device-side build (`hermnet_build_triadic`, the
        HVNet build's counting sort over the expanded list); everything else the torch-op build below, which defines
        the result (tests/test_htnet.py compares the two)."""
        if (atomic_number.is_cuda and switches.native_relations and len(z_list) > 0
                and atomic_number.numel() > 0):
            g = RelationalGraph._build_triadic_native(atomic_number, edge_index, z_list, edge_shift, batch, rel_active)
            if g is not None:
                return g
        g = RelationalGraph()
        dev = atomic_number.device
        NA, E0, T = int(atomic_number.numel()), int(edge_index.size(1)), len(z_list)
        pairs = [(p, q) for p in range(T) for q in range(p, T)]
        P = len(pairs)
        TR = T * P
        z = atomic_number.long()
        zl = torch.tensor(list(z_list), dtype=torch.long, device=dev)
        eq = z[:, None] == zl[None, :]
        rel = torch.where(eq.any(1), eq.int().argmax(1), torch.full((NA,), T, dtype=torch.long, device=dev))
        counts = torch.zeros(T + 1, dtype=torch.long, device=dev).index_add_(0, rel, torch.ones_like(rel))
        nb = batch.long().max().reshape(1) + 1 if (batch is not None and NA > 0) else torch.ones(1, dtype=torch.long, device=dev)
        host = torch.cat([counts, nb]).cpu().tolist()          # the one host sync of the build
        cnt_host, g.num_graphs = host[:T + 1], int(host[-1])
        B = max(cnt_host[:T]) if T > 0 else 0
        Ns, Nt = T * B + cnt_host[T], TR * B
        g.num_atoms, g.T, g.device, g.uniform, g.block = NA, TR, dev, True, B
        g.N, g.num_src, g.triadic_pairs = Nt, Ns, P
        g.type_rowptr_host = [r * B for r in range(TR + 1)]
        i32 = torch.int32
        g.type_rowptr = torch.tensor(g.type_rowptr_host, dtype=i32, device=dev)

        # source rows: (element, id) order, element blocks of B rows, unknown elements behind
        g.node_order = torch.sort(rel.to(i32), stable=True).indices
        first_sorted = torch.zeros(T + 1, dtype=torch.long, device=dev)
        first_sorted[1:] = torch.cumsum(counts[:T], 0)
        rel_sorted = rel[g.node_order]
        local_sorted = torch.arange(NA, device=dev) - first_sorted[rel_sorted]
        rows_sorted = local_sorted + rel_sorted * B
        g.row_of_node = torch.empty_like(g.node_order)
        g.row_of_node[g.node_order] = rows_sorted
        local = torch.empty_like(g.node_order)
        local[g.node_order] = local_sorted
        g.z_rows = torch.zeros(Ns, dtype=torch.long, device=dev)
        g.z_rows[g.row_of_node] = z
        g.src_real = torch.zeros(Ns, dtype=torch.float32, device=dev)
        g.src_real[g.row_of_node] = 1.0
        g.row_real = g.src_real                                   # energy read-out masks SOURCE rows

        # expanded edge list: edge (j -> i) once per pair containing element(j)
        pair_of = torch.full((T + 1, T), -1, dtype=torch.long, device=dev)      # element -> its T pair indices
        for e_ in range(T):
            ks = [k for k, (p, q) in enumerate(pairs) if e_ in (p, q)]
            pair_of[e_] = torch.tensor(ks, dtype=torch.long, device=dev)
        src, tgt = edge_index[0].long(), edge_index[1].long()
        tj, ti = rel[src], rel[tgt]
        ok = (tj < T) & (ti < T)
        eid = torch.nonzero(ok).reshape(-1)
        eid_x = eid.repeat_interleave(T)                           # expanded: (edge id major, pair minor)
        k_x = pair_of[tj[eid]].reshape(-1)
        ti_x = ti[eid_x]
        rel_x = ti_x * P + k_x
        vt_x = rel_x * B + local[tgt[eid_x]]                       # virtual target row
        rs_x = g.row_of_node[src[eid_x]]                           # source row
        E = int(eid_x.numel())
        g.E = E
        big = TR * max(Ns, 1) + 1 >= 2 ** 31
        kt = torch.long if big else i32
        vt_s, csr_perm = torch.sort(vt_x.to(kt), stable=True)
        vt_s = vt_s.long()
        csr_src = rs_x[csr_perm]
        ar = lambda n: torch.arange(n + 1, device=dev)
        csr_rowptr = torch.searchsorted(vt_s, ar(Nt))
        key2 = (rel_x[csr_perm] * Ns + csr_src)
        key2_s, csc_pos = torch.sort(key2.to(kt), stable=True)
        csc_rowptr = torch.searchsorted(key2_s.long(), ar(TR * Ns))
        g.csr_rowptr, g.csr_src, g.csr_perm = csr_rowptr.to(i32), csr_src.to(i32), eid_x[csr_perm]
        g.csc_rowptr, g.csc_pos, g.csc_tgt = csc_rowptr.to(i32), csc_pos.to(i32), vt_s[csc_pos].to(i32)
        g.out_rowptr = g.out_edges = None
        g.src_id, g.tgt_id = src[g.csr_perm].to(i32), tgt[g.csr_perm].to(i32)
        g.shift = None if edge_shift is None else edge_shift[g.csr_perm].float().contiguous()
        # target rows: real = the atom exists; active = its relation has at least one edge (hermnet.py:56-57)
        rows = torch.arange(Nt, device=dev)
        r_rel, r_loc = rows // max(B, 1), rows % max(B, 1)
        r_el = r_rel // P
        cnt_d = counts[:T]
        row_real = (r_loc < cnt_d[r_el.clamp(max=max(T - 1, 0))]).float() if Nt > 0 else torch.zeros(0, device=dev)
        tn = torch.arange(TR + 1, device=dev) * B
        rel_edges = csr_rowptr[tn[1:]] - csr_rowptr[tn[:-1]] if TR > 0 else torch.zeros(0, dtype=torch.long, device=dev)
        # (`rel_active` [T P]: the caller knows better, e.g. a shard whose relation has edges on other ranks only)
        if rel_active is None:
            act = rel_edges > 0
        elif torch.is_tensor(rel_active):
            act = rel_active.to(dev).bool()
        else:
            act = torch.tensor([bool(a) for a in rel_active], dtype=torch.bool, device=dev)
        g.row_active = act[r_rel].float() * row_real if Nt > 0 else row_real
        g.res_row = (r_el * B + r_loc).to(i32)
        # relation (c; p, q) gathers sources of elements p and q only: the x_proj chain skips the other rows
        g.src_ranges = _cached_i32(tuple(v for _c in range(T) for (p_, q_) in pairs
                                         for v in (p_ * B, p_ * B + cnt_host[p_], (q_ * B if q_ != p_ else 0),
                                                   (q_ * B + cnt_host[q_] if q_ != p_ else 0))), dev).view(TR, 4)
        g.batch32 = None if batch is None else batch.to(i32).contiguous()
        g.batch_rows = None
        if batch is not None and g.num_graphs > 1:
            g.graph_perm = torch.argsort(batch.long(), stable=True)
            g.graph_lengths = torch.zeros(g.num_graphs, dtype=torch.long, device=dev).index_add_(
                0, batch.long(), torch.ones_like(batch.long()))
        else:
            g.graph_perm = None
            g.graph_lengths = None
        return g

    def as_struct(self):
        from . import _lib
        if self._cstruct is None:      # the graph is immutable: build the ctypes view once, not per launch
            self._cstruct = self._make_struct(_lib)
        return self._cstruct

    def _make_struct(self, _lib):
        return _lib.Graph(self.N, self.E, self.T, self.type_rowptr.data_ptr(), self.csr_rowptr.data_ptr(),
                          self.csr_src.data_ptr(), self.csc_rowptr.data_ptr(), self.csc_tgt.data_ptr(),
                          self.csc_pos.data_ptr(), self.num_src, None if self.res_row is None else self.res_row.data_ptr())
