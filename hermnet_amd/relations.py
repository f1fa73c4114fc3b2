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
import torch


class RelationalGraph(object):
    __slots__ = ("N", "E", "T", "node_order", "row_of_node", "type_rowptr", "type_rowptr_host",
                 "rel_edges_host", "csr_rowptr", "csr_src", "csr_perm", "csc_rowptr", "csc_tgt", "csc_pos",
                 "out_rowptr", "out_edges", "src_id", "tgt_id", "shift", "row_active", "needs_mask",
                 "batch_rows", "batch32", "num_graphs", "graph_perm", "graph_lengths", "device")

    @staticmethod
    def build(atomic_number, edge_index, z_list, edge_shift=None, batch=None, rel_active=None):
        """atomic_number [N] int, edge_index [2,E] int (row 0 = source, row 1 = target,
        `hermnet.py:135`), z_list: atomic numbers of the model's elements in module order."""
        g = RelationalGraph()
        dev = atomic_number.device
        N = int(atomic_number.numel())
        E = int(edge_index.size(1))
        T = len(z_list)
        g.N, g.E, g.T, g.device = N, E, T, dev
        z = atomic_number.long()
        zl = torch.tensor(list(z_list), dtype=torch.long, device=dev)
        eq = z[:, None] == zl[None, :]
        # relation of each atom: first matching element, T for "not in elems" (hermnet.py:53 finds none)
        rel = torch.where(eq.any(1), eq.int().argmax(1), torch.full((N,), T, dtype=torch.long, device=dev))
        ar = torch.arange(N, device=dev)
        g.node_order = torch.argsort(rel * N + ar)
        g.row_of_node = torch.empty_like(g.node_order)
        g.row_of_node[g.node_order] = ar
        counts = torch.bincount(rel, minlength=T + 1)
        type_rowptr = torch.zeros(T + 1, dtype=torch.long, device=dev)
        type_rowptr[1:] = torch.cumsum(counts[:T], 0)

        src, tgt = edge_index[0].long(), edge_index[1].long()
        rs, rt = g.row_of_node[src], g.row_of_node[tgt]
        csr_perm = torch.argsort(rt, stable=True)
        csr_src = rs[csr_perm]
        rt_s = rt[csr_perm]
        csr_rowptr = torch.zeros(N + 1, dtype=torch.long, device=dev)
        csr_rowptr[1:] = torch.cumsum(torch.bincount(rt, minlength=N), 0)

        rel_row = rel[g.node_order]                       # relation of each row
        key2 = rel_row[rt_s] * N + csr_src                # (relation(target), row(source))
        csc_pos = torch.argsort(key2, stable=True)
        csc_tgt = rt_s[csc_pos]
        csc_cnt = torch.bincount(key2, minlength=(T + 1) * N)[:T * N]
        csc_rowptr = torch.zeros(T * N + 1, dtype=torch.long, device=dev)
        csc_rowptr[1:] = torch.cumsum(csc_cnt, 0)

        out_edges = torch.argsort(csr_src, stable=True)
        out_rowptr = torch.zeros(N + 1, dtype=torch.long, device=dev)
        out_rowptr[1:] = torch.cumsum(torch.bincount(csr_src, minlength=N), 0)

        rel_edges = csc_rowptr[torch.arange(1, T + 1, device=dev) * N] - csc_rowptr[torch.arange(0, T, device=dev) * N]
        nb = batch.long().max().reshape(1) + 1 if (batch is not None and N > 0) else torch.ones(1, dtype=torch.long, device=dev)
        host = torch.cat([type_rowptr, rel_edges, nb]).cpu().tolist()   # the one host sync of the build
        g.type_rowptr_host = host[:T + 1]
        g.rel_edges_host = host[T + 1:2 * T + 1]
        g.num_graphs = int(host[-1])
        # hermnet.py:56-57: a relation without edges is skipped -> its rows stay zero; rows of
        # unknown-type atoms are zero as well (hermnet.py:51).
        # (`rel_active`: the caller knows better, e.g. a shard whose relation has edges on other ranks only)
        active = [ne > 0 for ne in g.rel_edges_host] if rel_active is None else [bool(a) for a in rel_active]
        g.needs_mask = any((not active[t]) and g.type_rowptr_host[t + 1] > g.type_rowptr_host[t] for t in range(T))
        if g.needs_mask:
            act = torch.tensor(active + [False], dtype=torch.bool, device=dev)
            g.row_active = act[rel_row].float()
        else:
            g.row_active = None

        i32 = torch.int32
        g.type_rowptr = type_rowptr.to(i32)
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
        g.batch_rows = None if batch is None else batch.long()[g.node_order]
        g.batch32 = None if batch is None else batch.to(i32).contiguous()
        # deterministic per-graph read-out: rows grouped by graph (stable) + segment lengths
        if g.batch_rows is not None and g.num_graphs > 1:
            g.graph_perm = torch.argsort(g.batch_rows, stable=True)
            g.graph_lengths = torch.bincount(g.batch_rows, minlength=g.num_graphs)
        else:
            g.graph_perm = None
            g.graph_lengths = None
        return g

    def as_struct(self):
        from . import _lib
        return _lib.Graph(self.N, self.E, self.T, self.type_rowptr.data_ptr(), self.csr_rowptr.data_ptr(),
                          self.csr_src.data_ptr(), self.csc_rowptr.data_ptr(), self.csc_tgt.data_ptr(),
                          self.csc_pos.data_ptr())
