"""Atom-sharded evaluation of one structure over the GPUs of a node.

The reference has nothing comparable (one graph lives on one device; its only collective is
DDP for training, `example/dist_train.py:25,67`).  The build shards the hot path by TARGET atom:

  * atoms are split into `world` slabs along one cell axis (equal atom counts);
  * a rank keeps every directed edge whose target it owns; sources it does not own are its
    one-hop halo;
  * layer 0 needs no exchange (x0 = embed(Z), vec0 = 0 are local); after each of the first
    L-1 layers the owners send the new (x, vec) rows of halo atoms to the ranks that need them --
    ONE variable-size all-to-all (RCCL point-to-point over xGMI) of n_halo * 4H floats per layer;
  * backward mirrors it (gradients of halo rows return to the owners and are accumulated there),
    positions use the same exchange once per step, the energy is one scalar all-reduce.

Every rank derives the complete plan from the global coordinates (the calculators have all
coordinates anyway), so there is no negotiation step.  Two planners:

  * `partition_slab` (the production path): pure geometry on the device.  Owners = equal-count slabs;
    the halo of a rank = the atoms within rc of its slab along the slab axis (periodic); the rank runs the
    cutoff neighbour search ONLY over its owned + halo atoms and keeps the edges whose target it owns.  Cost
    per rank ~ N / world; nothing of size E_global is ever built.
  * `partition` (planning from a GLOBAL edge list on the host: batches of molecules, tests): exact halos
    (only atoms that really send an edge across).
"""
import numpy as np
import torch
import torch.distributed as dist

from .data import Data


class ExchangePlan(object):
    """Index lists of one halo exchange in some local ordering.

    send_idx   [n_send]  local indices of owned entries to send, grouped by destination rank
    send_counts[world]   entries per destination
    recv_idx   [n_recv]  local indices of halo entries to fill, grouped by source rank
    recv_counts[world]
    """

    def __init__(self, send_idx, send_counts, recv_idx, recv_counts, group=None):
        self.send_idx, self.send_counts = send_idx, [int(c) for c in send_counts]
        self.recv_idx, self.recv_counts = recv_idx, [int(c) for c in recv_counts]
        self.group = group
        self._acc = None

    def accumulate_lists(self):
        """(rows [nu], ptr [nu+1], pos [n_send]) for the deterministic accumulation of returned gradients: an owned
        entry that is sent to several peers gets its contributions summed in send-list order."""
        if self._acc is None:
            idx = self.send_idx
            pos = torch.argsort(idx, stable=True)
            rows, cnt = torch.unique_consecutive(idx[pos], return_counts=True)
            ptr = torch.zeros(rows.numel() + 1, dtype=torch.long, device=idx.device)
            ptr[1:] = torch.cumsum(cnt, 0)
            self._acc = (rows.contiguous(), ptr, pos.contiguous())
        return self._acc

    def remap(self, index_map):
        """Same exchange expressed in another ordering (e.g. relation rows): idx -> index_map[idx]."""
        p = ExchangePlan(index_map.index_select(0, self.send_idx), self.send_counts,
                         index_map.index_select(0, self.recv_idx),
                         self.recv_counts, self.group)
        rows, ptr, pos = self.accumulate_lists()       # the grouping does not depend on the (injective) ordering
        p._acc = (index_map.index_select(0, rows), ptr, pos)
        return p


def _host_staged(group, t):
    """gloo has no device all-to-all: rehearsal runs (several ranks sharing one GPU) stage through the host.
    The production backend is "nccl" (= RCCL over xGMI), which takes the device buffers directly."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


class CommProbe(object):
    """Diagnostics of the halo exchanges of a sharded step (bench.py's `comm` block): HIP events on the compute
    stream around every point where it waits for a collective.  `wait(work, tag)` replaces `work.wait()`; the blocking
    exchanges are bracketed whole.  Off (None) in production: an event pair costs ~15 us of host time."""

    def __init__(self):
        self.records = []          # (tag, rows_out, rows_in, start event, end event)

    def _pair(self):
        return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def wait(self, work, tag, rows_out=0, rows_in=0):
        a, b = self._pair()
        a.record()
        work.wait()
        b.record()
        self.records.append((tag, int(rows_out), int(rows_in), a, b))

    def bracket(self, tag, rows_out, rows_in, fn):
        a, b = self._pair()
        a.record()
        out = fn()
        b.record()
        self.records.append((tag, int(rows_out), int(rows_in), a, b))
        return out

    def summary(self):
        """tag -> dict(count, ms_total, rows_out, rows_in) (synchronises)."""
        torch.cuda.synchronize()
        out = {}
        for tag, ro, ri, a, b in self.records:
            d = out.setdefault(tag, dict(count=0, ms_total=0.0, rows_out=0, rows_in=0))
            d["count"] += 1
            d["ms_total"] += a.elapsed_time(b)
            d["rows_out"] += ro
            d["rows_in"] += ri
        return out


_PROBE = None


def set_comm_probe(probe):
    global _PROBE
    _PROBE = probe


def comm_wait(work, tag, rows_out=0, rows_in=0):
    """The compute stream waits for an asynchronous exchange (`_all_to_all_rows_start`); timed when a probe is set."""
    if work is None:
        return
    if _PROBE is not None and torch.cuda.is_available():
        _PROBE.wait(work, tag, rows_out, rows_in)
    else:
        work.wait()


def _all_to_all_rows(buf, in_counts, out_counts, group):
    if _host_staged(group, buf):
        staged = lambda: _all_to_all_rows(buf.cpu(), in_counts, out_counts, group).to(buf.device)
        if _PROBE is not None:      # (rehearsal: the host-staged exchange shows up as a blocking one)
            return _PROBE.bracket("blocking", sum(in_counts), sum(out_counts), staged)
        return staged()
    out = buf.new_empty((sum(out_counts),) + tuple(buf.shape[1:]))

    def go():
        dist.all_to_all_single(out, buf.contiguous(), output_split_sizes=out_counts, input_split_sizes=in_counts,
                               group=group)
        return out
    if _PROBE is not None and buf.is_cuda:
        return _PROBE.bracket("blocking", sum(in_counts), sum(out_counts), go)
    return go()


def _all_to_all_rows_start(buf, in_counts, out_counts, group):
    """(out, work): the same exchange, started asynchronously where the backend allows it.  RCCL ("nccl") runs the
    collective on its own stream behind everything already enqueued on the current one; kernels launched next overlap
    with it, and `work.wait()` makes the current STREAM (not the host) wait for the result.  The gloo rehearsal
    (ranks sharing one GPU, CPU tests) completes it here: work = None."""
    if not buf.is_cuda or _host_staged(group, buf):
        return _all_to_all_rows(buf, in_counts, out_counts, group), None
    out = buf.new_empty((sum(out_counts),) + tuple(buf.shape[1:]))
    work = dist.all_to_all_single(out, buf.contiguous(), output_split_sizes=out_counts, input_split_sizes=in_counts,
                                  group=group, async_op=True)
    return out, work


class HaloOverlap(object):
    """An exchange of (x, vec) halo rows that is still DUE when a layer starts: the fused layer
    (`layer.FusedRelationalLayer`) runs it itself, with everything that does not wait for it in between (SURVEY 8(e):
    "run interior edges while the halo is in flight"), and mirrors that in its backward.

    plan       ExchangePlan in relation-row order
    windows    [W,2] int32 (device): row ranges that contain every halo row (one per element block)
    fwd_early  [T,2] int32 (device): per relation the TARGET rows that read no halo row -- their messages run while the
               exchange is in flight; fwd_late: the rest of the relation's row block (incl. the halo rows themselves)
    bwd_first  (device [k,2] int32, host [(lo, hi)]): the SOURCE rows whose gradients travel -- the halo windows widened to
               the node kernels' row tiles --, computed and sent first; bwd_rest: every other row, computed meanwhile.
    bwd_first_rows / bwd_rest_rows: the same cut at row granularity (the exact halo windows and their complement): what the
               "proj" form of the exchange uses (layer.py: projected rows travel, no node kernel runs on a subset of tiles)."""

    def __init__(self, plan, windows, fwd_early=None, fwd_late=None, bwd_first=None, bwd_rest=None):
        self.plan, self.windows = plan, windows
        self.fwd_early, self.fwd_late, self.bwd_first, self.bwd_rest = fwd_early, fwd_late, bwd_first, bwd_rest

    @staticmethod
    def row_windows(halo_rows, type_rowptr, num_rows):
        """One [lo, hi) row window per relation block (+ the unknown-element block) around the given rows; device
        ops only, empty blocks give (0, 0)."""
        T1 = int(type_rowptr.numel())                       # T + 1 block starts -> T + 1 blocks incl. unknown elements
        dev = halo_rows.device
        blk = torch.bucketize(halo_rows, type_rowptr.long()[1:], right=True)
        lo = torch.full((T1,), int(num_rows), dtype=torch.long, device=dev).scatter_reduce(0, blk, halo_rows, "amin")
        hi = torch.full((T1,), -1, dtype=torch.long, device=dev).scatter_reduce(0, blk, halo_rows, "amax") + 1
        lo = torch.minimum(lo, hi)                          # empty block: (0, 0)
        return torch.stack([lo, hi], dim=1).to(torch.int32).contiguous()

    @staticmethod
    def target_ranges(late_rows, type_rowptr):
        """(early, late) [T,2] int32: relation block t = [start_t, end_t) is cut at its first late row b_t into the early
        range [start_t, b_t) and the late range [b_t, end_t).  `late_rows`: rows that read a halo row or are one."""
        T = int(type_rowptr.numel()) - 1
        rp = type_rowptr.long()
        dev = rp.device
        blk = torch.bucketize(late_rows, rp[1:], right=True)               # block of every late row (T = unknown elements)
        first = torch.cat([rp[1:], rp[-1:]]).clone()                        # default: the block's end (no late row)
        first = first.scatter_reduce(0, blk, late_rows, "amin")[:T]
        first = torch.maximum(first, rp[:T])
        early = torch.stack([rp[:T], first], dim=1).to(torch.int32).contiguous()
        late = torch.stack([first, rp[1:T + 1]], dim=1).to(torch.int32).contiguous()
        return early, late

    @staticmethod
    def source_ranges(windows_host, tile_rows, num_rows, device):
        """(first, rest): the windows widened to multiples of `tile_rows` and merged, and their complement in
        [0, num_rows); each as (device [k,2] int32, host [(lo, hi)])."""
        wide = sorted((lo // tile_rows * tile_rows, min(-(-hi // tile_rows) * tile_rows, num_rows))
                      for lo, hi in windows_host if hi > lo)
        first = []
        for lo, hi in wide:
            if first and lo <= first[-1][1]:
                first[-1] = (first[-1][0], max(first[-1][1], hi))
            else:
                first.append((lo, hi))
        rest, at = [], 0
        for lo, hi in first:
            if lo > at:
                rest.append((at, lo))
            at = hi
        if at < num_rows:
            rest.append((at, num_rows))
        dv = lambda r: torch.tensor(r, dtype=torch.int32, device=device).reshape(-1, 2) if r else None
        return (dv(first), first), (dv(rest), rest)


class HaloExchange(torch.autograd.Function):
    """out = x with its halo entries replaced by the owners' current values (differentiable).  Used for the
    coordinates (once per step); index_select / index_copy_ / index_fill_ only: advanced-index assignment
    (`x[idx] = v`) costs a host synchronisation per call."""

    @staticmethod
    def forward(ctx, x, plan):
        ctx.plan = plan
        recv = _all_to_all_rows(x.detach().index_select(0, plan.send_idx), plan.send_counts, plan.recv_counts, plan.group)
        out = x.detach().clone()
        out.index_copy_(0, plan.recv_idx, recv)
        return out

    @staticmethod
    def backward(ctx, g):
        plan = ctx.plan
        back = _all_to_all_rows(g.index_select(0, plan.recv_idx), plan.recv_counts, plan.send_counts, plan.group)
        gx = _writable(g)                              # (the producer's fresh tensor where this is its only consumer)
        gx.index_fill_(0, plan.recv_idx, 0)            # the local halo values were overwritten in forward
        if back.size(0) > 0:
            rows, ptr, pos = plan.accumulate_lists()
            if rows.numel() == plan.send_idx.numel():
                # every owned atom goes to at most ONE peer (slabs thicker than the halo reach): the send list has no repeats,
                # so the returns land on distinct rows -- one launch, deterministic
                gx.index_add_(0, plan.send_idx, back)
            else:
                # an atom that is a halo atom of several peers gets its returns summed in send-list order (no atomics)
                # (unsafe=True: no host-side validation of `lengths` -- it would synchronise every step)
                seg = torch.segment_reduce(back.index_select(0, pos), "sum", lengths=ptr[1:] - ptr[:-1], unsafe=True)
                gx.index_add_(0, rows, seg)            # `rows` is unique
        return gx, None


class HaloGradReturn(torch.autograd.Function):
    """Identity on coordinates whose halo rows are ALREADY the owners' values (slab plans: every rank is handed all
    coordinates, `slab_data`); the backward sends the gradients of the halo rows to their owners and accumulates them
    there -- the return half of `HaloExchange`, without the forward collective."""

    @staticmethod
    def forward(ctx, x, plan):
        ctx.plan = plan
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return HaloExchange.backward(ctx, g)


def _writable(g):
    """A gradient this backward may update in place: the engine hands over the producer's fresh tensor when the
    forward output had a single consumer (the next layer); anything that looks shared is copied first."""
    return g if (g.is_contiguous() and g._base is None and not g.requires_grad) else g.clone(memory_format=torch.contiguous_format)


class HaloExchangeFeatures(torch.autograd.Function):
    """(x [N,H], vec [N,3,H]) of one layer -> the same tensors with the halo rows replaced by their owners'
    values: ONE all-to-all of n_halo * 4H floats, halo rows written IN PLACE (no copy of the full arrays)."""

    @staticmethod
    def forward(ctx, x, vec, plan):
        ctx.plan = plan
        H = x.size(1)
        if x.is_cuda:      # two kernels + one collective (csrc/node_kernels.hip: hermnet_halo_rows)
            from . import nodeops
            send = nodeops.halo_rows(0, x, vec, plan.send_idx)
            recv = _all_to_all_rows(send, plan.send_counts, plan.recv_counts, plan.group)
            nodeops.halo_rows(2, x, vec, plan.recv_idx, recv)
        else:              # host tensors: planning / CPU rehearsal of the exchange logic (tests)
            send = torch.cat([x.detach().index_select(0, plan.send_idx),
                              vec.detach().index_select(0, plan.send_idx).reshape(-1, 3 * H)], dim=1)
            recv = _all_to_all_rows(send, plan.send_counts, plan.recv_counts, plan.group)
            x.index_copy_(0, plan.recv_idx, recv[:, :H])
            vec.index_copy_(0, plan.recv_idx, recv[:, H:].reshape(-1, 3, H))
        ctx.mark_dirty(x, vec)
        return x, vec

    @staticmethod
    def backward(ctx, gx, gvec):
        plan = ctx.plan
        H = gx.size(1)
        if gx.is_cuda:
            from . import nodeops
            gx, gvec = _writable(gx), _writable(gvec)
            # the local halo values were overwritten in forward: their gradients go to the owners, none stays here
            gsend = nodeops.halo_rows(1, gx, gvec, plan.recv_idx)
            back = _all_to_all_rows(gsend, plan.recv_counts, plan.send_counts, plan.group)
            nodeops.halo_accumulate(gx, gvec, plan, back)               # gradients of my atoms used elsewhere
            return gx, gvec, None
        gsend = torch.cat([gx.index_select(0, plan.recv_idx),
                           gvec.index_select(0, plan.recv_idx).reshape(-1, 3 * H)], dim=1)
        back = _all_to_all_rows(gsend, plan.recv_counts, plan.send_counts, plan.group)
        gx, gvec = _writable(gx), _writable(gvec)
        gx.index_fill_(0, plan.recv_idx, 0)          # the local halo values were overwritten in forward
        gvec.index_fill_(0, plan.recv_idx, 0)
        gx.index_add_(0, plan.send_idx, back[:, :H])                      # gradients of my atoms used elsewhere
        gvec.index_add_(0, plan.send_idx, back[:, H:].reshape(-1, 3, H))
        return gx, gvec, None


class SumAcrossRanks(torch.autograd.Function):
    """E_total = sum_r E_r (all-reduce); d E_total / d E_r = 1 on every rank."""

    @staticmethod
    def forward(ctx, e, group):
        out = e.detach().clone()
        if _host_staged(group, out):
            host = out.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            return host.to(out.device)
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        return g, None


class ShardPlan(object):
    """What HVNet.forward needs to run one rank's share: see `partition`."""

    def __init__(self, rank, world, owned_global, halo_global, atom_plan, owned_mask, num_graphs, group=None):
        self.rank, self.world = rank, world
        self.num_graphs = int(num_graphs)     # of the GLOBAL batch (a rank may own no atom of some graph)
        self.owned_global = owned_global      # LongTensor: global ids of owned atoms in local order
        self.halo_global = halo_global        # LongTensor: global ids of halo atoms, ascending
        self.owned_local = None               # LongTensor: local ids of the owned atoms (set by `partition`)
        self.local_global = None              # LongTensor: global id of every local atom
        self.atom_plan = atom_plan            # ExchangePlan in local atom order
        self.owned_mask = owned_mask          # BoolTensor [N_loc]
        self.group = group
        self.z_with_in_edges = None           # atomic numbers that receive >= 1 edge GLOBALLY (hermnet.py:56-57): a set
        self.has_in_edges = None              # ... or the same as a device array [128] of 0/1 (slab plans)
        self.zz_with_in_edges = None          # (target Z, source Z) pairs joined by >= 1 edge globally: a set ...
        self.has_in_pairs = None              # ... or a device array [128 * 128] of 0/1 (slab plans)
        self._row_plan = None                 # (row_of_node tensor, atom_plan in that row order, HaloOverlap)
        self._zl_index = None                 # ((element list, device), its index tensor) for `rel_active`
        self.halo_pos_local = False           # True: Data.pos already holds the halo atoms' coordinates (slab plans)
        self.late_local = None                # BoolTensor [N_loc]: local atoms that may READ a halo atom's row (or are halo
                                              # atoms); the others' messages run while the halo exchange is in flight

    @property
    def n_owned(self):
        return int(self.owned_global.numel())

    def rel_active(self, zl):
        """Which of the model's relations (target elements `zl`) run: a list of bool (host plans) or a uint8 device
        array (slab plans; consumed by the relation build without a host read)."""
        if self.has_in_edges is not None:
            key = (tuple(zl), self.has_in_edges.device)
            if self._zl_index is None or self._zl_index[0] != key:      # (a host-to-device copy waits for the stream: once)
                self._zl_index = (key, torch.tensor([min(int(z), 127) for z in zl], dtype=torch.long, device=key[1]))
            return self.has_in_edges.index_select(0, self._zl_index[1]).to(torch.uint8)
        return [z in self.z_with_in_edges for z in zl]

    def rel_active_triadic(self, zl):
        """HTNet's relations (c; {p, q}), c-major and pairs p-major (relations.build_triadic): one runs iff an atom of
        element c receives an edge from an atom of element p or q anywhere in the structure."""
        T = len(zl)
        trip = [(c, p, q) for c in range(T) for p in range(T) for q in range(p, T)]
        if self.has_in_pairs is not None:
            dev = self.has_in_pairs.device
            key = ("triadic", tuple(zl), dev)
            if self._zl_index is None or self._zl_index[0] != key:
                cl = lambda k: min(int(zl[k]), 127)
                self._zl_index = (key, torch.tensor([[cl(c) * 128 + cl(p) for c, p, q in trip],
                                                     [cl(c) * 128 + cl(q) for c, p, q in trip]], dtype=torch.long, device=dev))
            idx = self._zl_index[1]
            return (self.has_in_pairs.index_select(0, idx[0]) | self.has_in_pairs.index_select(0, idx[1])).to(torch.uint8)
        zz = self.zz_with_in_edges
        return [((zl[c], zl[p]) in zz) or ((zl[c], zl[q]) in zz) for c, p, q in trip]

    def graph_order(self, batch):
        """(perm, lengths): the local atoms sorted by graph id (stable) and the number of local atoms of each of the
        `num_graphs` graphs -- what an ordered per-graph reduction needs; cached per `batch` tensor (one sort per plan)."""
        c = getattr(self, "_graph_order", None)
        if c is None or c[0] is not batch or c[1] != batch._version:
            perm = torch.argsort(batch, stable=True)
            lengths = torch.bincount(batch, minlength=self.num_graphs)
            c = self._graph_order = (batch, batch._version, perm, lengths)
        return c[2], c[3]

    def owned_rows(self, graph, rows):
        """[rows] float 0 / 1: the node rows (HVNet: graph.N; HTNet: its source rows) that hold an OWNED atom (halo atoms and
        padding rows: 0) -- the read-out's row mask on a shard; cached with the row layout."""
        c = getattr(self, "_owned_rows", None)
        if c is None or c[0] is not graph.row_of_node or c[1].numel() != rows:
            m = torch.zeros(rows, dtype=torch.float32, device=graph.row_of_node.device)
            m.index_copy_(0, graph.row_of_node, self.owned_mask.to(torch.float32))
            c = self._owned_rows = (graph.row_of_node, m)
        return c[1]

    @property
    def num_atoms_global(self):
        """Atoms of the whole structure (single-structure plans: every atom is owned by exactly one rank)."""
        n = getattr(self, "_n_global", None)
        if n is None:
            raise RuntimeError("this plan does not know the size of the whole structure (set plan._n_global)")
        return n

    def row_plan(self, row_of_node):
        """The atom exchange lists in the relation-row order of `row_of_node` (cached per tensor: the row layout of an
        atom set does not change along a trajectory, relations.py)."""
        if self._row_plan is None or self._row_plan[0] is not row_of_node:
            self._row_plan = (row_of_node, self.atom_plan.remap(row_of_node), None)
        return self._row_plan[1]

    def halo_overlap(self, graph, tile_rows=64):
        """`HaloOverlap` of this plan for the row layout of `graph` (cached with the row plan; one host read of the
        T + 1 halo windows when it is made).  `tile_rows`: rows per tile of the node pre kernels at the model's width."""
        plan = self.row_plan(graph.row_of_node)
        if self._row_plan[2] is None or self._row_plan[2].tile_rows != tile_rows:
            dev = graph.row_of_node.device
            if graph.num_src:      # HTNet: the exchange and the node projection live in SOURCE rows (blocks of B rows per
                Te, B = graph.T // graph.triadic_pairs, graph.block          # element), the messages land in virtual rows
                src_rowptr = torch.arange(Te + 1, dtype=torch.int32, device=dev) * B
                n_src = graph.num_src
            else:
                src_rowptr, n_src = graph.type_rowptr, graph.N
            win = HaloOverlap.row_windows(plan.recv_idx, src_rowptr, n_src)
            ov = HaloOverlap(plan, win)
            ov.tile_rows = tile_rows
            if True:
                late = self.late_local
                if late is None:                  # no classification: every owned atom may read a halo row
                    late = torch.ones(graph.row_of_node.numel(), dtype=torch.bool, device=dev)
                if plan.recv_idx.numel() == 0:    # no halo atom here: nothing is late (the rank still joins the exchange)
                    late = torch.zeros_like(late)
                late_rows = graph.row_of_node.index_select(0, torch.nonzero(late).reshape(-1))
                if graph.num_src:                 # a late atom is late in every pair relation of its element
                    P_ = graph.triadic_pairs
                    c, ii = late_rows // B, late_rows % B
                    k = torch.arange(P_, device=dev)
                    late_rows = ((c[:, None] * P_ + k[None, :]) * B + ii[:, None]).reshape(-1)
                    late_rows = late_rows[c.repeat_interleave(P_) < Te]          # (atoms of unlisted elements: no target rows)
                ov.fwd_early, ov.fwd_late = HaloOverlap.target_ranges(late_rows, graph.type_rowptr)
                el = torch.stack([ov.fwd_early, ov.fwd_late]).cpu()          # (one more small host read per plan)
                ov.early_rows = int((el[0, :, 1] - el[0, :, 0]).clamp(min=0).sum())
                ov.late_rows = int((el[1, :, 1] - el[1, :, 0]).clamp(min=0).sum())
                win_host = win.cpu().tolist()
                ov.bwd_first, ov.bwd_rest = HaloOverlap.source_ranges(win_host, tile_rows, n_src, dev)
                # the same cut at row granularity: the "proj" exchange (layer.py) runs no node kernel on a subset of the tiles
                ov.bwd_first_rows, ov.bwd_rest_rows = HaloOverlap.source_ranges(win_host, 1, n_src, dev)
            self._row_plan = self._row_plan[:2] + (ov,)
        return self._row_plan[2]

    def to(self, device):
        self.owned_global = self.owned_global.to(device)
        self.halo_global = self.halo_global.to(device)
        self.owned_mask = self.owned_mask.to(device)
        self.owned_local = self.owned_local.to(device)
        self.local_global = self.local_global.to(device)
        if self.late_local is not None:
            self.late_local = self.late_local.to(device)
        p = self.atom_plan
        self.atom_plan = ExchangePlan(p.send_idx.to(device), p.send_counts, p.recv_idx.to(device), p.recv_counts,
                                      p.group)
        return self


def slab_owner(pos, cell, world, axis=None):
    """Owner rank of every atom: equal-count slabs along `axis` (default: the longest cell vector)."""
    pos = np.asarray(pos, dtype=np.float64)
    n = pos.shape[0]
    if cell is not None:
        c = np.asarray(cell, dtype=np.float64).reshape(3, 3)
        if axis is None:
            axis = int(np.argmax(np.linalg.norm(c, axis=1)))
        frac = pos @ np.linalg.inv(c)
        coord = frac[:, axis] - np.floor(frac[:, axis])
    else:
        if axis is None:
            axis = int(np.argmax(pos.max(0) - pos.min(0)))
        coord = pos[:, axis]
    order = np.argsort(coord, kind="stable")
    owner = np.empty(n, dtype=np.int64)
    bounds = [(n * r) // world for r in range(world + 1)]
    for r in range(world):
        owner[order[bounds[r]:bounds[r + 1]]] = r
    return owner


def partition(data, rank, world, axis=None, group=None):
    """Split a global `Data` (host tensors) for `rank` of `world`.

    Returns (local_data, plan).  local_data holds the rank's owned and halo atoms in ascending GLOBAL id, i.e.
    interleaved (`plan.owned_local` / `plan.owned_mask` say which are owned): halo atoms receive no edge here, and
    with the halo rows spread through every relation's row block the message kernels' workgroups (contiguous
    row chunks) stay balanced -- appended at the end they would leave ~10 % of the workgroups empty.  Edges are
    the global edges whose target is owned, re-indexed locally; `batch`, `cell`, `edge_shift` follow.  Positions
    of halo atoms are placeholders (zeros): HVNet.forward fills them through the exchange so that force
    contributions flow back to the owners.
    """
    pos = data.pos.detach().cpu().numpy()
    cell = data.get("cell")
    cell_np = None if cell is None else cell.detach().cpu().numpy().reshape(-1, 3, 3)[0]
    owner = slab_owner(pos, cell_np, world, axis)
    ei = data.edge_index.cpu().numpy()
    src, tgt = ei[0], ei[1]
    n = pos.shape[0]

    def halo_of(r):
        """(global ids of rank r's halo atoms, their owners), sorted by (owner, id)."""
        need = np.unique(src[(owner[tgt] == r) & (owner[src] != r)])
        key = np.lexsort((need, owner[need]))
        need = need[key]
        return need, owner[need]

    owned = np.nonzero(owner == rank)[0]
    halo, halo_owner = halo_of(rank)
    local_ids = np.sort(np.concatenate([owned, halo]))
    is_owned = owner[local_ids] == rank
    g2l = np.full(n, -1, dtype=np.int64)
    g2l[local_ids] = np.arange(len(local_ids))
    emask = owner[tgt] == rank
    lsrc, ltgt = g2l[src[emask]], g2l[tgt[emask]]
    assert (lsrc >= 0).all() and (ltgt >= 0).all()

    # what I receive: my halo grouped by owner; what I send: for every other rank, its halo atoms that I own
    recv_counts = [int((halo_owner == p).sum()) for p in range(world)]
    recv_idx = g2l[halo]
    send_lists = []
    for p in range(world):
        if p == rank:
            send_lists.append(np.zeros(0, dtype=np.int64))
            continue
        hp, hp_owner = halo_of(p)
        send_lists.append(g2l[hp[hp_owner == rank]])
    send_counts = [len(s) for s in send_lists]
    send_idx = np.concatenate(send_lists) if send_lists else np.zeros(0, dtype=np.int64)

    pos_l = data.pos[torch.from_numpy(local_ids)].clone()
    pos_l[torch.from_numpy(~is_owned)] = 0.0
    kw = dict(pos=pos_l,
              atomic_number=data.atomic_number[torch.from_numpy(local_ids)],
              edge_index=torch.from_numpy(np.vstack([lsrc, ltgt])).long(),
              batch=(data.batch if data.get("batch") is not None else torch.zeros(n, dtype=torch.long))[
                  torch.from_numpy(local_ids)])
    if cell is not None:
        kw["cell"] = cell
        if data.get("edge_shift") is not None:
            kw["edge_shift"] = data.edge_shift[torch.from_numpy(np.nonzero(emask)[0])]
    local = Data(**kw)
    owned_mask = torch.from_numpy(is_owned.copy())
    num_graphs = int(data.batch.max()) + 1 if data.get("batch") is not None and n > 0 else 1
    plan = ShardPlan(rank, world, torch.from_numpy(owned), torch.from_numpy(np.sort(halo)),
                     ExchangePlan(torch.from_numpy(send_idx), send_counts, torch.from_numpy(recv_idx), recv_counts,
                                  group),
                     owned_mask, num_graphs, group)
    plan.owned_local = torch.from_numpy(np.nonzero(is_owned)[0])
    plan.local_global = torch.from_numpy(local_ids.copy())
    plan._n_global = int(n)
    late = ~is_owned                                   # halo atoms, and the owned atoms that have a halo source
    late[ltgt[~is_owned[lsrc]]] = True
    plan.late_local = torch.from_numpy(late.copy())
    z_np = data.atomic_number.cpu().numpy()
    plan.z_with_in_edges = set(int(v) for v in np.unique(z_np[tgt]))
    plan.zz_with_in_edges = set((int(a), int(b)) for a, b in np.unique(np.stack([z_np[tgt], z_np[src]], 1), axis=0))
    local._hn_shard = plan
    return local, plan


def _axis_coordinate(pos, cell, axis):
    """(coordinate along the slab axis in [0,1) for periodic cells / Cartesian for open systems, axis, margin):
    `margin` = rc-independent factor such that `rc * margin` is the cutoff expressed in that coordinate."""
    p64 = pos.detach().double()
    if cell is not None:
        c = cell.detach().double().reshape(-1, 3, 3)[0].cpu()
        if axis is None:
            axis = int(torch.argmax(c.norm(dim=1)))
        inv = torch.linalg.inv(c)
        col = inv[:, axis].to(p64.device)
        # elementwise, fixed order: every rank must get bit-identical coordinates (no BLAS here)
        frac = p64[:, 0] * col[0] + p64[:, 1] * col[1] + p64[:, 2] * col[2]
        return frac - torch.floor(frac), axis, float(inv[:, axis].norm())     # 1 / plane spacing
    if axis is None:
        ext = (p64.max(0).values - p64.min(0).values) if p64.size(0) > 0 else torch.zeros(3)
        axis = int(torch.argmax(ext))
    return p64[:, axis].clone(), axis, 1.0


def slab_owner_device(pos, cell, world, axis=None):
    """Equal-count slabs along `axis` with torch ops on `pos.device` (same definition as `slab_owner`).
    Returns (owner [N] int64, coord [N] float64, bounds (lo [world], hi [world]) of every slab, axis, margin)."""
    coord, axis, margin = _axis_coordinate(pos, cell, axis)
    n = coord.numel()
    dev = coord.device
    order = torch.argsort(coord, stable=True)
    cuts = torch.tensor([(n * r) // world for r in range(world + 1)], dtype=torch.long, device=dev)
    rank_of_sorted = torch.bucketize(torch.arange(n, device=dev), cuts[1:], right=True).clamp(max=world - 1)
    owner = torch.empty(n, dtype=torch.long, device=dev)
    owner[order] = rank_of_sorted
    cs = coord[order]
    nonempty = cuts[1:] > cuts[:-1]
    lo = torch.where(nonempty, cs[cuts[:-1].clamp(max=max(n - 1, 0))], torch.full((world,), float("nan"), dtype=cs.dtype, device=dev))
    hi = torch.where(nonempty, cs[(cuts[1:] - 1).clamp(min=0)], torch.full((world,), float("nan"), dtype=cs.dtype, device=dev))
    return owner, coord, (lo, hi), axis, margin


def _within_cutoff_of_slab(coord, lo, hi, reach, periodic):
    """Atoms whose distance along the slab axis to the interval [lo, hi] is <= reach (periodic: in fractional
    coordinates, through the cell boundary as well).  Empty slabs (lo = nan) reach nothing."""
    inside = (coord >= lo) & (coord <= hi)
    if periodic:
        up = torch.remainder(lo - coord, 1.0)        # going up from the atom to the slab's lower face
        down = torch.remainder(coord - hi, 1.0)      # going down to its upper face
        dist = torch.minimum(up, down)
    else:
        dist = torch.maximum(lo - coord, coord - hi)
    return inside | (dist <= reach)


def plan_slab(pos, atomic_number, cell, rc, rank, world, axis=None, group=None, skin=0.0):
    """The part of a slab decomposition that depends on WHERE the atoms are only to within `skin`/2: owners
    (equal-count slabs), the geometric halo (everything within rc + skin of the slab along the slab axis), the
    exchange lists.  pos [N,3] float32, atomic_number [N], cell [3,3] / [1,3,3] / None -- the same on every rank, on
    the device the step will run on (host tensors work too: CPU rehearsal).

    Local atoms = owned interior atoms, owned atoms within rc + skin of a slab face, halo atoms -- each group in
    ascending global id (so every relation's row block reads [interior | boundary | halo]: "rows that wait for the
    exchange" are T windows for the node kernels and a row range per relation for the message kernels).
    The halo is geometric, so sender and receiver derive identical exchange lists independently; it is a superset of
    the atoms that really send an edge across.  While no atom has moved further than skin/2 from `plan.pos_ref`,
    every source within rc of an owned atom is still a local atom: the plan stays valid and only the neighbour
    list is rebuilt (`slab_data`).  All peers' send lists come from ONE [world, N] mask; one host read (the
    2 x world counts)."""
    dev = pos.device
    n = pos.size(0)
    owner, coord, (lo, hi), axis, margin = slab_owner_device(pos, cell, world, axis)
    periodic = cell is not None
    reach = (float(rc) + float(skin)) * margin * (1.0 + 1e-9) + 1e-12
    mine = owner == rank
    # near[p, a]: atom a lies within reach of slab p (atoms of slab p itself included)
    near = _within_cutoff_of_slab(coord[None, :], lo[:, None], hi[:, None], reach, periodic)
    halo_mask = near[rank] & ~mine
    send_mask = near & mine[None, :]
    send_mask[rank] = False
    # receive: my halo grouped by owner (ascending id inside a group); send: for every peer, the atoms I own
    # that lie within reach of ITS slab -- the same set and order the peer derives for its receive list
    # owned atoms that can have a source outside the slab: within `reach` of one of its faces (a source within rc of an
    # atom further inside lies strictly between the faces, i.e. is owned -- also after moves of up to skin/2 each).
    # They follow the interior atoms in the local order, so every relation's row block reads
    # [interior | boundary | halo]: the messages into the interior rows run while the halo exchange is in flight.
    near_face = mine & ((coord - lo[rank] <= reach) | (hi[rank] - coord <= reach))
    interior = torch.nonzero(mine & ~near_face).reshape(-1)
    boundary = torch.nonzero(near_face).reshape(-1)
    owned = torch.cat([interior, boundary])                                         # each part in ascending global id
    halo = torch.nonzero(halo_mask).reshape(-1)                                     # ascending global id
    local_ids = torch.cat([owned, halo])
    n_owned = owned.numel()
    g2l = torch.full((n,), -1, dtype=torch.long, device=dev)
    g2l[local_ids] = torch.arange(local_ids.numel(), device=dev)
    halo_owner = owner[halo]
    recv_idx = g2l[halo[torch.argsort(halo_owner, stable=True)]]
    pairs = torch.nonzero(send_mask)                                                # (peer, atom), peer-major
    send_idx = g2l[pairs[:, 1]]
    counts = torch.stack([torch.bincount(halo_owner, minlength=world),
                          torch.bincount(pairs[:, 0], minlength=world)]).cpu().tolist()   # the host read of the plan
    is_owned = torch.zeros(local_ids.numel(), dtype=torch.bool, device=dev)
    is_owned[:n_owned] = True
    plan = ShardPlan(rank, world, owned, halo, ExchangePlan(send_idx, counts[1], recv_idx, counts[0], group),
                     is_owned, 1, group)
    plan.owned_local = torch.arange(n_owned, device=dev)
    plan.local_global = local_ids
    plan._n_global = int(n)
    plan.rc, plan.skin = float(rc), float(skin)
    plan.pos_ref = pos.detach().clone()
    plan.z_local = atomic_number[local_ids]
    plan.batch_local = torch.zeros(local_ids.numel(), dtype=torch.long, device=dev)
    plan.target_mask = is_owned.to(torch.uint8)
    plan.cell = None if cell is None else cell.detach().reshape(1, 3, 3)
    plan.halo_pos_local = True            # `slab_data` fills the halo coordinates itself
    plan.late_local = torch.ones(local_ids.numel(), dtype=torch.bool, device=dev)
    plan.late_local[:interior.numel()] = False
    return plan


def plan_self_peer(pos, atomic_number, cell, rc, virtual=2, axis=None, group=None, skin=0.0):
    """A slab plan whose halo peer is THIS rank: the structure is cut into `virtual` equal-count slabs along the slab axis
    exactly as `plan_slab` cuts it for `virtual` ranks, but all of them live here.  The rank owns every atom; an atom within
    rc + skin of ANOTHER virtual slab additionally exists as a ghost row per such slab, which is what the targets of that slab
    read (`slab_data` keeps the pair j -> i only through the instance of j that serves i's slab).  Ghost rows are halo rows in
    every respect: they are filled by the per-layer exchange -- `all_to_all_single` with send_counts = recv_counts = [k] on a
    group of ONE rank --, their gradients travel back the same way and are summed into the owners' rows in send-list order.

    Why it exists: one-GPU boxes cannot run two RCCL ranks, so without it the production
    exchange (asynchronous start, stream wait, in-place unpack, halo re-projection, early / late message ranges, reverse
    gradient exchange) never carries a row before the first 8-GPU run.  With it all of that executes over RCCL with real
    payload on one GPU, and the result must equal the unsharded evaluation.  Local order as in `plan_slab`:
    [interior | owned atoms near a face of their slab | ghosts]."""
    dev = pos.device
    n = pos.size(0)
    V = int(virtual)
    slab, coord, (lo, hi), axis, margin = slab_owner_device(pos, cell, V, axis)
    periodic = cell is not None
    reach = (float(rc) + float(skin)) * margin * (1.0 + 1e-9) + 1e-12
    if V == 1:
        # ONE slab whose neighbour through the periodic boundary is itself (what a rank of an 8-slab plan looks like, halo
        # fraction included): ghosts = the atoms within reach of the slab's faces THROUGH the boundary; a pair uses the
        # ghost of its source iff it crosses the boundary (`slab_data`: shift along the slab axis != 0)
        if not periodic:
            raise ValueError("virtual=1 cuts at the periodic boundary: it needs a cell")
        ghost = (torch.minimum(torch.remainder(lo[0] - coord, 1.0), torch.remainder(coord - hi[0], 1.0)) <= reach)[None, :]
    else:
        near = _within_cutoff_of_slab(coord[None, :], lo[:, None], hi[:, None], reach, periodic)       # [V, N]
        ghost = near & (slab[None, :] != torch.arange(V, device=dev)[:, None])
    pairs = torch.nonzero(ghost)                                                   # (served slab, atom), slab-major
    lo_a, hi_a = lo.index_select(0, slab), hi.index_select(0, slab)
    near_face = (coord - lo_a <= reach) | (hi_a - coord <= reach)
    interior = torch.nonzero(~near_face).reshape(-1)
    boundary = torch.nonzero(near_face).reshape(-1)
    owned = torch.cat([interior, boundary])
    local_ids = torch.cat([owned, pairs[:, 1]])
    g2l = torch.empty(n, dtype=torch.long, device=dev)
    g2l[owned] = torch.arange(n, device=dev)
    k = int(pairs.size(0))                                                         # the host read of the plan
    send_idx = g2l.index_select(0, pairs[:, 1])
    recv_idx = n + torch.arange(k, device=dev)
    is_owned = torch.zeros(n + k, dtype=torch.bool, device=dev)
    is_owned[:n] = True
    plan = ShardPlan(0, 1, owned, pairs[:, 1].contiguous(), ExchangePlan(send_idx, [k], recv_idx, [k], group),
                     is_owned, 1, group)
    plan.owned_local = torch.arange(n, device=dev)
    plan.local_global = local_ids
    plan._n_global = int(n)
    plan.rc, plan.skin = float(rc), float(skin)
    plan.pos_ref = pos.detach().clone()
    plan.z_local = atomic_number[local_ids]
    plan.batch_local = torch.zeros(n + k, dtype=torch.long, device=dev)
    plan.target_mask = is_owned.to(torch.uint8)
    plan.cell = None if cell is None else cell.detach().reshape(1, 3, 3)
    plan.halo_pos_local = True
    plan.late_local = torch.ones(n + k, dtype=torch.bool, device=dev)
    plan.late_local[:interior.numel()] = False
    # the virtual slab every local row SERVES: an owned row its own, a ghost row the slab it was made for
    plan.serves_slab = torch.cat([slab.index_select(0, owned), pairs[:, 0]]).contiguous()
    plan.virtual = V
    plan.wrap_axis = axis if V == 1 else None
    return plan


def partition_self_peer(pos, atomic_number, cell, rc, virtual=2, axis=None, group=None, reference_compat=False, skin=0.0):
    """`plan_self_peer` + `slab_data`: (local_data, plan) of the one rank."""
    plan = plan_self_peer(pos, atomic_number, cell, rc, virtual=virtual, axis=axis, group=group, skin=skin)
    return slab_data(plan, pos, reference_compat), plan


def block_grid(world, cell=None, pos=None):
    """A (pa, pb, pc) factorisation of `world` for `plan_blocks`: the factors go to the longest extents first, so that the
    blocks come out as cubic as the cell allows (a 36 x 36 x 900 A cell at world 8 -> (1, 1, 8): slabs; a 105 A cube ->
    (2, 2, 2))."""
    if cell is not None:
        ext = [float(v) for v in cell.detach().double().reshape(-1, 3, 3)[0].norm(dim=1).cpu()]
    else:
        p = pos.detach().double()
        ext = [float(v) for v in (p.max(0).values - p.min(0).values).cpu()] if p.size(0) else [1.0, 1.0, 1.0]
    grid, n = [1, 1, 1], int(world)
    f = 2
    primes = []
    while n > 1:
        while n % f == 0:
            primes.append(f)
            n //= f
        f += 1
    for q in sorted(primes, reverse=True):           # largest factor to the currently longest block edge
        a = max(range(3), key=lambda k: ext[k] / grid[k])
        grid[a] *= q
    return tuple(grid)


def _equal_count_cuts(coord, groups, parts):
    """Inside every group (`groups` [N] int64 group id, ids 0 .. G-1) split the atoms into `parts` equal-count pieces by
    ascending `coord` (stable).  Returns the piece index [N] in 0 .. parts-1."""
    n = coord.numel()
    dev = coord.device
    if parts == 1 or n == 0:
        return torch.zeros(n, dtype=torch.long, device=dev)
    G = int(groups.max()) + 1 if n else 1
    # sort by (group, coord): a stable sort by coord followed by a stable sort by group
    o1 = torch.argsort(coord, stable=True)
    o2 = torch.argsort(groups[o1], stable=True)
    order = o1[o2]
    gs = groups[order]
    cnt = torch.bincount(gs, minlength=G)
    start = torch.cumsum(cnt, 0) - cnt
    rank_in = torch.arange(n, device=dev) - start[gs]
    piece_sorted = (rank_in * parts) // cnt[gs].clamp(min=1)          # pieces of floor / ceil(count / parts) atoms
    piece = torch.empty(n, dtype=torch.long, device=dev)
    piece[order] = piece_sorted.clamp(max=parts - 1)
    return piece


def plan_blocks(pos, atomic_number, cell, rc, rank, world, grid=None, group=None, skin=0.0):
    """`plan_slab` in up to three dimensions ("spatial slabs (or blocks)"): the atoms are cut into pa equal-count
    slabs along the first axis, every slab into pb equal-count strips along the second, every strip into pc boxes along
    the third (grid = (pa, pb, pc), pa pb pc = world; default `block_grid`).  A rank's halo is everything within
    rc + skin of its box along EVERY cut axis (periodic in fractional coordinates) -- geometric, so sender and receiver
    derive identical lists independently, exactly as for slabs; for a near-cubic cell the halo of a box is a fraction of
    a slab's.  Same plan object, same local order (interior | near a box face | halo)."""
    dev = pos.device
    n = pos.size(0)
    if grid is None:
        grid = block_grid(world, cell, pos)
    grid = tuple(int(g_) for g_ in grid)
    if grid[0] * grid[1] * grid[2] != world:
        raise ValueError("grid %r does not multiply to the world size %d" % (grid, world))
    periodic = cell is not None
    p64 = pos.detach().double()
    coords, margins = [], []
    if periodic:
        c = cell.detach().double().reshape(-1, 3, 3)[0].cpu()
        inv = torch.linalg.inv(c)
        for a in range(3):
            col = inv[:, a].to(dev)
            # elementwise, fixed order: every rank must get bit-identical coordinates (no BLAS here)
            fr = p64[:, 0] * col[0] + p64[:, 1] * col[1] + p64[:, 2] * col[2]
            coords.append(fr - torch.floor(fr))
            margins.append(float(inv[:, a].norm()))
    else:
        for a in range(3):
            coords.append(p64[:, a].clone())
            margins.append(1.0)
    owner = torch.zeros(n, dtype=torch.long, device=dev)
    for a in range(3):                                    # hierarchical equal-count cuts
        owner = owner * grid[a] + _equal_count_cuts(coords[a], owner, grid[a])
    mine = owner == rank
    # every rank's box along the cut axes: [lo, hi] of its atoms' coordinates (nan for an empty rank: reaches nothing)
    near = torch.ones(world, n, dtype=torch.bool, device=dev)
    near_face = torch.zeros(n, dtype=torch.bool, device=dev)
    big = torch.full((world,), float("inf"), dtype=torch.float64, device=dev)
    for a in range(3):
        if grid[a] == 1:
            continue
        reach = (float(rc) + float(skin)) * margins[a] * (1.0 + 1e-9) + 1e-12
        lo = big.clone().scatter_reduce(0, owner, coords[a], "amin")
        hi = (-big).clone().scatter_reduce(0, owner, coords[a], "amax")
        empty = lo > hi
        lo = torch.where(empty, torch.full_like(lo, float("nan")), lo)
        hi = torch.where(empty, torch.full_like(hi, float("nan")), hi)
        near &= _within_cutoff_of_slab(coords[a][None, :], lo[:, None], hi[:, None], reach, periodic)
        near_face |= mine & ((coords[a] - lo[rank] <= reach) | (hi[rank] - coords[a] <= reach))
    return _finish_plan(pos, atomic_number, cell, rc, skin, rank, world, group, owner, mine, near, near_face)


def _finish_plan(pos, atomic_number, cell, rc, skin, rank, world, group, owner, mine, near, near_face):
    """Local order, exchange lists and plan object from the owners and the geometric reach masks (`plan_blocks`)."""
    dev = pos.device
    n = pos.size(0)
    halo_mask = near[rank] & ~mine
    send_mask = near & mine[None, :]
    send_mask[rank] = False
    interior = torch.nonzero(mine & ~near_face).reshape(-1)
    boundary = torch.nonzero(near_face).reshape(-1)
    owned = torch.cat([interior, boundary])
    halo = torch.nonzero(halo_mask).reshape(-1)
    local_ids = torch.cat([owned, halo])
    n_owned = owned.numel()
    g2l = torch.full((n,), -1, dtype=torch.long, device=dev)
    g2l[local_ids] = torch.arange(local_ids.numel(), device=dev)
    halo_owner = owner[halo]
    recv_idx = g2l[halo[torch.argsort(halo_owner, stable=True)]]
    pairs = torch.nonzero(send_mask)                                                # (peer, atom), peer-major
    send_idx = g2l[pairs[:, 1]]
    counts = torch.stack([torch.bincount(halo_owner, minlength=world),
                          torch.bincount(pairs[:, 0], minlength=world)]).cpu().tolist()   # the host read of the plan
    is_owned = torch.zeros(local_ids.numel(), dtype=torch.bool, device=dev)
    is_owned[:n_owned] = True
    plan = ShardPlan(rank, world, owned, halo, ExchangePlan(send_idx, counts[1], recv_idx, counts[0], group),
                     is_owned, 1, group)
    plan.owned_local = torch.arange(n_owned, device=dev)
    plan.local_global = local_ids
    plan._n_global = int(n)
    plan.rc, plan.skin = float(rc), float(skin)
    plan.pos_ref = pos.detach().clone()
    plan.z_local = atomic_number[local_ids]
    plan.batch_local = torch.zeros(local_ids.numel(), dtype=torch.long, device=dev)
    plan.target_mask = is_owned.to(torch.uint8)
    plan.cell = None if cell is None else cell.detach().reshape(1, 3, 3)
    plan.halo_pos_local = True            # `slab_data` fills the halo coordinates itself
    plan.late_local = torch.ones(local_ids.numel(), dtype=torch.bool, device=dev)
    plan.late_local[:interior.numel()] = False
    return plan


def partition_blocks(pos, atomic_number, cell, rc, rank, world, grid=None, group=None, reference_compat=False, skin=0.0):
    """`partition_slab` with boxes instead of slabs: `plan_blocks` + `slab_data`."""
    plan = plan_blocks(pos, atomic_number, cell, rc, rank, world, grid=grid, group=group, skin=skin)
    return slab_data(plan, pos, reference_compat), plan


def slab_data(plan, pos, reference_compat=False, capacity=None, want_moved=False):
    """This rank's `Data` for the current coordinates under a plan that is still valid (`plan_moved` says whether it
    is): the cutoff pairs among the local atoms whose TARGET is owned, listed directly by the neighbour search
    (`target_mask`; nothing of the other pairs is built or filtered afterwards).  Halo rows of `pos` hold the atoms'
    coordinates (every rank has them all): no coordinate exchange in the forward pass, HVNet.forward only routes the
    halo rows' force contributions back to the owners (`HaloGradReturn`).  One host read (the edge count of the
    search) -- or none: with `capacity` (periodic cells on the GPU) the list is padded to that many columns with NULL edges
    (`neighbor.neighbor_search_padded`), `local._hn_edge_count` holds (E, flags) on the device, and the caller checks
    them when it copies the step's results to the host anyway (`SlabStepper.check`).  `want_moved`: `local._hn_moved` =
    `plan_moved(plan, pos)` (on the GPU from the same launch as the relation flags)."""
    from .neighbor import neighbor_search, neighbor_search_padded
    dev = pos.device
    pos_l = pos.detach().index_select(0, plan.local_global)
    cell = plan.cell
    total = None
    if cell is not None and capacity is not None and pos_l.is_cuda:
        ei, sh, total = neighbor_search_padded(pos_l, plan.rc, cell, int(capacity), reference_compat=reference_compat,
                                               target_mask=plan.target_mask)
    elif cell is not None:
        ei, sh = neighbor_search(pos_l, plan.rc, cell, reference_compat=reference_compat, target_mask=plan.target_mask)
    else:
        ei, sh = neighbor_search(pos_l, plan.rc, None, target_mask=plan.target_mask), None
        if reference_compat:
            # `radius_graph`'s cap (data.py:16) keeps a target's 32 lowest source INDICES -- indices of the whole
            # structure.  The local order is owned-then-halo, so the cap is applied on global ids here (the capped list
            # of a shard is then exactly the unsharded capped list restricted to the owned targets).
            ei = _cap_by_global_source(ei, plan.local_global, int(plan.pos_ref.size(0)), 32)
    serves = getattr(plan, "serves_slab", None)
    if serves is not None:
        # self-peer plan: an atom near a cut exists twice among the local rows (its owned row and a ghost row per slab it
        # reaches); a pair j -> i is kept through the ONE instance of j that serves i's virtual slab (the owned row when both
        # are in the same slab, otherwise the ghost made for i's slab) -- what a rank of a real `virtual`-rank plan would list
        if total is not None:
            raise NotImplementedError("the self-peer plan searches exactly (no padded list)")
        if plan.wrap_axis is not None:      # one virtual slab: the ghost of j iff the pair crosses the periodic boundary
            keep = (~plan.owned_mask.index_select(0, ei[0])) == (sh[:, plan.wrap_axis] != 0)
        else:
            keep = serves.index_select(0, ei[0]) == serves.index_select(0, ei[1])
        ei = ei[:, keep].contiguous()
        sh = None if sh is None else sh[keep].contiguous()
    z = plan.z_local
    kw = dict(pos=pos_l, atomic_number=z, edge_index=ei, batch=plan.batch_local)
    if cell is not None:
        kw["cell"] = cell
        kw["edge_shift"] = sh
    local = Data(**kw)
    if total is not None:
        local._hn_edge_count = total
    # hermnet.py:56-57: a relation is skipped when NO atom of its element receives an edge anywhere in the structure;
    # kept on the device (the relation build takes the flags as a device array: no host read per step)
    # (flags per (target element, source element): HTNet's relations are skipped pair-wise; HVNet reads the row maxima)
    # (the NULL edges of a padded list -- endpoints -1 -- raise the spare slot behind the table)
    # (slot 128 * 128: where the NULL edges of a padded list -- endpoints -1 -- land; slot 128 * 128 + 1: "this rank's padded
    # list is incomplete", so that the SAME reduction tells every rank whether the step has to be repeated anywhere)
    moved = None
    if (pos_l.is_cuda and z.dtype == torch.long and ei.dtype == torch.long and ei.is_contiguous() and z.is_contiguous()
            and pos.dtype == torch.float32 and plan.pos_ref.dtype == torch.float32):
        # clear + mark (csrc/relation_kernels.hip: hermnet_shard_step_flags) for what is sixteen small launches below
        from . import _lib
        P = _lib.ptr
        flags = torch.empty(128 * 128 + 3, dtype=torch.int32, device=dev)
        ask = bool(want_moved) and plan.pos_ref.shape == pos.shape
        if ask:
            cur, ref = pos.detach().contiguous(), plan.pos_ref.contiguous()
        _lib.check(_lib.load().hermnet_shard_step_flags(
            P(ei), int(ei.size(1)), P(z), int(z.numel()), None if total is None else P(total),
            0 if capacity is None else int(capacity), P(flags), P(cur) if ask else None, P(ref) if ask else None,
            int(pos.size(0)) if ask else 0, float((0.5 * plan.skin) ** 2),
            torch.cuda.current_stream(dev).cuda_stream), "hermnet_shard_step_flags")
        has_in = flags[:128 * 128 + 2]
        if ask:
            moved = flags[128 * 128 + 2]
    else:
        has_in = torch.zeros(128 * 128 + 2, dtype=torch.int32, device=dev)
        if ei.size(1) > 0:
            zt, zs = z[ei[1].clamp(min=0)].clamp(max=127), z[ei[0].clamp(min=0)].clamp(max=127)
            has_in.index_fill_(0, torch.where(ei[1] >= 0, zt * 128 + zs, torch.full_like(zt, 128 * 128)), 1)
        if total is not None:
            has_in[128 * 128 + 1:] = ((total[1:] != 0) | (total[:1] > int(capacity))).to(torch.int32)
    if want_moved and moved is None:
        moved = plan_moved(plan, pos)
    if plan.world > 1 and dist.is_available() and dist.is_initialized():
        if _host_staged(plan.group, has_in):
            h = has_in.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.MAX, group=plan.group)
            has_in = h.to(dev)
        else:
            dist.all_reduce(has_in, op=dist.ReduceOp.MAX, group=plan.group)
    local._hn_list_bad = has_in[128 * 128 + 1]               # 0-d, on the device, the same on every rank
    local._hn_moved = moved
    has_in = has_in[:128 * 128]
    plan.has_in_pairs = has_in
    plan.has_in_edges = has_in.view(128, 128).amax(dim=1)
    local._hn_shard = plan
    return local


def _cap_by_global_source(edge_index, local_global, n_global, cap):
    """Keep at most `cap` in-edges per target: those whose sources have the lowest GLOBAL ids; the surviving edges stay
    in their (target, local source) order."""
    E = edge_index.size(1)
    if E == 0:
        return edge_index
    dev = edge_index.device
    key = edge_index[1] * int(n_global) + local_global.index_select(0, edge_index[0])
    order = torch.argsort(key, stable=True)
    tgt = edge_index[1].index_select(0, order)
    pos = torch.arange(E, device=dev)
    first = torch.ones(E, dtype=torch.bool, device=dev)
    first[1:] = tgt[1:] != tgt[:-1]
    start = torch.cummax(torch.where(first, pos, torch.zeros_like(pos)), 0).values
    keep = torch.zeros(E, dtype=torch.bool, device=dev)
    keep[order] = (pos - start) < cap
    return edge_index[:, keep]


def plan_moved(plan, pos):
    """0-d bool tensor: some atom is further than skin/2 from where the plan saw it (same answer on every rank: all
    ranks hold the same coordinates)."""
    d2 = ((pos.detach() - plan.pos_ref) ** 2).sum(dim=1)
    return d2.max() > (0.5 * plan.skin) ** 2 if d2.numel() > 0 else torch.zeros((), dtype=torch.bool, device=pos.device)


def partition_slab(pos, atomic_number, cell, rc, rank, world, axis=None, group=None, reference_compat=False, skin=0.0):
    """Slab decomposition of ONE structure from its coordinates alone (no global edge list): `plan_slab` +
    `slab_data`.  Returns (local_data, plan) with the contract of `partition`, except for the order of the local
    atoms (owned first, then halo; see `plan_slab`)."""
    plan = plan_slab(pos, atomic_number, cell, rc, rank, world, axis=axis, group=group, skin=skin)
    return slab_data(plan, pos, reference_compat), plan


class SlabStepper(object):
    """Per-step planning of an atom-sharded MD run: the plan is kept while it is valid under a Verlet skin.

        stepper = SlabStepper(atomic_number, cell, rc, rank, world, skin=1.0)
        local, plan = stepper(pos)          # every step; `pos` = the global coordinates, the same on every rank

    Per step: the displacement check (three elementwise launches), the slab-local neighbour search for the current
    coordinates and the has-in-edges reduction; the slab plan itself (owners, halo, exchange lists, their host read)
    only when an atom has moved further than skin/2 since it was made (the check's flag is read back first, so a
    re-plan step searches once).  The cell and the atomic numbers are watched by tensor identity and version: replace
    them or edit them in place, never through `.data` (writes through `.data` or a numpy view do not bump the version).

    `deferred=True` (round 5; periodic cells on the GPU): NO host read per step.  The step runs optimistically on the plan
    it has and on a neighbour list padded to a capacity (`slab_data(capacity=...)`); the displacement flag and the list's
    (count, flags) stay on the device, and

        ok = stepper.check()                # ONE host read -- do it when the step's results go to the host anyway

    says whether the step just taken was valid (every atom within skin/2 of the plan, list complete).  If not, the caller
    repeats the step: the next `stepper(pos)` plans again and / or searches exactly.  The first call of a plan runs the
    exact search (its count sizes the capacity).  Several steps may be taken between two checks: the flags are sticky, so
    `check()` then answers for ALL of them (False: repeat from the last checked state)."""

    def __init__(self, atomic_number, cell, rc, rank, world, skin=1.0, axis=None, group=None, reference_compat=False,
                 grid=None, deferred=False, self_peer=0):
        """`grid` = (pa, pb, pc) or "auto" (`block_grid`): boxes instead of slabs (`plan_blocks`).
        `self_peer` = V > 1 (world size 1 only): `plan_self_peer` with V virtual slabs -- the halo exchange with real rows on
        one GPU; searches exactly (no padded list)."""
        self.z, self.cell, self.rc, self.skin = atomic_number, cell, float(rc), float(skin)
        self.rank, self.world, self.axis, self.group = rank, world, axis, group
        self.grid = grid
        self.reference_compat = reference_compat
        self.self_peer = int(self_peer or 0)
        if self.self_peer and world != 1:
            raise ValueError("a self-peer plan is a plan of ONE rank (world size %d)" % world)
        self.deferred = bool(deferred) and not self.self_peer
        self.plan = None
        self.replans = 0
        self.repeats = 0                 # steps that `check` asked to be taken again
        self._cell_key = None
        self._z_key = (None, None)
        self._capacity = None            # columns of the next padded list (None: exact search)
        self._pending = None             # (moved flag [0-d bool], total [2] or None) of the step not checked yet
        self._force_replan = False

    def _replan(self, pos):
        if self.self_peer:
            self.plan = plan_self_peer(pos, self.z, self.cell, self.rc, virtual=self.self_peer, axis=self.axis,
                                       group=self.group, skin=self.skin)
        elif self.grid is not None:
            self.plan = plan_blocks(pos, self.z, self.cell, self.rc, self.rank, self.world,
                                    grid=None if self.grid == "auto" else self.grid, group=self.group, skin=self.skin)
        else:
            self.plan = plan_slab(pos, self.z, self.cell, self.rc, self.rank, self.world, axis=self.axis,
                                  group=self.group, skin=self.skin)
        self.replans += 1
        self._capacity = None
        self._force_replan = False

    def _exact(self, pos):
        from .neighbor import padded_capacity
        local = slab_data(self.plan, pos, self.reference_compat)
        self._pending = None
        if self.deferred and self.cell is not None and pos.is_cuda:
            self._capacity = padded_capacity(int(local.edge_index.size(1)))     # (the count is on the host already)
            # (another rank may be on a padded list that turns out incomplete: every rank reads the same reduced flag)
            self._pending = (plan_moved(self.plan, pos), None, local._hn_list_bad)
        return local, self.plan

    def check(self):
        """Deferred mode: was the step on the data of the last call valid -- on EVERY rank (the answer is the same
        everywhere: the displacement flag is computed from the same coordinates, the list flags are reduced over the
        ranks)?  One host read; on False every rank repeats the step (the next call plans again and / or searches
        exactly).  True when nothing is pending."""
        if self._pending is None:
            return True
        from .neighbor import padded_capacity
        moved, total, bad = self._pending
        self._pending = None
        parts = [moved.reshape(1).to(torch.long), bad.reshape(1).to(torch.long)] + ([] if total is None else [total])
        vals = torch.cat(parts).tolist()
        ok = not (vals[0] or vals[1])
        if vals[0]:
            self._force_replan = True
        if total is not None:
            E, flags = vals[2], vals[3]
            if flags & 2:
                from .neighbor import _stash_overflowed
                _stash_overflowed(total.device)
            if flags or E > self._capacity:
                self._capacity = None                            # this rank searches exactly next (its count sizes the capacity)
            elif not (E * 1.02 + 64 <= self._capacity <= E * 1.25 + 8192):
                self._capacity = padded_capacity(E)              # (still valid, but re-sized for the following steps)
        if not ok:
            self.repeats += 1
        return ok

    def __call__(self, pos):
        # (a cell that was replaced or edited in place -- NPT -- moves the slab bounds: plan again)
        cell_key = None if self.cell is None else (self.cell, self.cell._version)
        z_key = (self.z, self.z._version)          # an in-place species swap changes plan.z_local
        if self.plan is not None and (z_key[0] is not self._z_key[0] or z_key[1] != self._z_key[1]):
            self.plan = None
        self._z_key = z_key
        if self.plan is not None and (cell_key is None) != (self._cell_key is None):
            self.plan = None
        if self.plan is not None and cell_key is not None and (cell_key[0] is not self._cell_key[0] or cell_key[1] != self._cell_key[1]):
            self.plan = None
        self._cell_key = cell_key
        if self.plan is None or self.plan.pos_ref.shape != pos.shape or self._force_replan:
            self._replan(pos)
            return self._exact(pos)
        if self.deferred and self.cell is not None and pos.is_cuda:
            if self._capacity is None:
                if bool(plan_moved(self.plan, pos)):             # (an exact step reads the host anyway)
                    self._replan(pos)
                return self._exact(pos)
            # optimistic: the plan as it is, the list padded; both flags stay on the device until `check`
            local = slab_data(self.plan, pos, self.reference_compat, capacity=self._capacity, want_moved=True)
            moved, bad = local._hn_moved, local._hn_list_bad
            if self._pending is not None:
                # several steps between two `check()`s: the flags are STICKY -- an earlier step that was invalid keeps the
                # answer False (two tiny launches, only on this path; the counts are the latest list's)
                i32 = lambda t: t.reshape(()).to(torch.int32)
                moved = torch.maximum(i32(moved), i32(self._pending[0]))
                bad = torch.maximum(i32(bad), i32(self._pending[2]))
            self._pending = (moved, local.get("_hn_edge_count"), bad)
            return local, self.plan
        # one 0-d read-back decides (the search's own host read follows anyway): on a re-plan step the search and the
        # relation-flag reduction then run ONCE, on the new plan
        if bool(plan_moved(self.plan, pos)):
            self._replan(pos)
        return slab_data(self.plan, pos, self.reference_compat), self.plan
