"""Whole-step hipGraph replay: energy + forces of a FIXED neighbour list as one graph launch.

A step is ~200 kernel launches; eager enqueueing costs 2.7 ms of host time for a 3.8 ms step at 10k atoms, and is the
limit for small systems (a 1024-molecule batch: ~7 ms eager for ~2.5 ms of GPU work).  Capturing relation build +
forward + force backward once and replaying it removes the host from the loop.

Scope: the graph bakes in every launch geometry, so it is valid while `edge_index` (hence E and the relation layout)
stays the same -- repeated evaluation on one list (benchmarks, line searches, finite differences, several MD steps on a
list that is rebuilt every k steps).  The reference rebuilds the list every step (`calculator.py:49`), and beyond-cutoff
edges still send the `rbf_proj` bias, so a calculator may only reuse a graph while its list is unchanged;
`GraphedStep.matches(data)` is the check.

What made replay work (ROCm 7.2, bisected with tools/graph_probe.py): captured `hipMemsetAsync` nodes do not survive
eager memsets issued between two replays (the second replay skips / mis-addresses the fill), so nothing on the step
path uses hipMemsetAsync / hipMemcpyAsync any more (csrc/relation_kernels.hip: zero / copy / scan kernels).
"""
import torch


class GraphedStep(object):
    """energy, forces = step(pos) for a fixed graph topology.

    model: HVNet / HTNet in eval() with parameters that do not require grad; data: `Data` on the GPU with
    `edge_index` (and `cell` / `edge_shift`) already built.  Construction runs `warmup` eager steps on a side stream,
    then captures one step.  `__call__(pos=None)` copies `pos` into the static input (if given), replays, and returns
    the static output tensors (valid until the next call; clone them to keep them)."""

    def __init__(self, model, data, warmup=3):
        if not data.pos.is_cuda:
            raise RuntimeError("GraphedStep needs GPU tensors")
        if model.training:
            raise RuntimeError("GraphedStep captures the eval() path")
        self.model, self.data = model, data
        self.pos = data.pos.detach().clone().requires_grad_(True)      # static input
        data.pos = self.pos
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):                                      # caches, weights, library workspaces
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.energy, self.forces = self._eager()
        torch.cuda.synchronize()
        # the captured topology: the tensor OBJECTS (kept alive here, so their addresses cannot be handed to a rebuilt
        # list of the same size) and their versions (in-place edits); taken after the first forward, which fills in a
        # missing `batch`
        self._topo = self._topology(data)

    def _eager(self):
        e = self.model(self.data)
        f = -torch.autograd.grad(e.sum(), self.pos)[0]
        return e.detach(), f

    @staticmethod
    def _topology(data):
        keep = [data.get(k) for k in ("edge_index", "edge_shift", "cell", "atomic_number", "batch")]
        return [(t, None if t is None else t._version) for t in keep]

    def matches(self, data):
        """True while `data` carries the neighbour list (edge_index, edge_shift, cell, atomic numbers, batch) this graph
        was captured for: the same tensor objects, not modified in place since."""
        now = self._topology(data)
        return all(a is b and va == vb for (a, va), (b, vb) in zip(self._topo, now))

    def __call__(self, pos=None):
        if pos is not None:
            with torch.no_grad():
                self.pos.copy_(pos)
        self.graph.replay()
        return self.energy, self.forces


class GraphedMDStep(object):
    """energy, forces = step(pos) INCLUDING the neighbour search: cell list -> padded neighbour list -> relation build ->
    forward -> force backward as ONE hipGraph launch that stays valid across list rebuilds (the
    reference rebuilds its list on the host every step, `plugin/ase_interface/calculator.py:49`).

    What makes the launch geometry independent of the edge count: the list is padded to `capacity` columns with NULL
    edges that the relation build files behind every row (`neighbor.neighbor_search_padded`), and the search itself runs
    without a host read, library sort or memset.  model: HVNet in eval(), parameters not requiring grad; one periodic
    structure (atomic_number [N], cell [3,3], both on the GPU and unchanged for the life of the object; `pos` [N,3] gives
    the first coordinates).  `capacity` defaults to the first list's edge count + 4 %.

        step = GraphedMDStep(model, z, cell, pos0)
        e, f = step(pos)            # static outputs: valid until the next call
        ok, n_edges = step.check()  # a host read -- do it when e / f are copied to the host anyway; not ok: the list
                                    # outgrew the capacity (or left the cell by > 8 images): `step.recapture(pos)`"""

    def __init__(self, model, atomic_number, cell, pos, capacity=None, warmup=3, reference_compat=False):
        from .neighbor import neighbor_search, padded_capacity
        self.reference_compat = bool(reference_compat)      # edge conventions of the reference's own pipeline (neighbor.py)
        if not pos.is_cuda or cell is None:
            raise RuntimeError("GraphedMDStep needs GPU tensors and a periodic cell")
        if model.training:
            raise RuntimeError("GraphedMDStep captures the eval() path")
        self.model, self.z, self.cell = model, atomic_number, cell
        self.batch = torch.zeros(atomic_number.numel(), dtype=torch.long, device=pos.device)
        self.pos = pos.detach().clone().float().requires_grad_(True)      # static input
        self._warmup = warmup
        if capacity is None:
            capacity = padded_capacity(int(neighbor_search(self.pos.detach(), model.rc, cell,
                                                           reference_compat=self.reference_compat)[0].size(1)))
        self._capture(int(capacity))

    def _eager(self):
        from .data import Data
        from .neighbor import neighbor_search_padded
        ei, sh, total = neighbor_search_padded(self.pos.detach(), self.model.rc, self.cell, self.capacity,
                                               reference_compat=self.reference_compat)
        d = Data(pos=self.pos, atomic_number=self.z, batch=self.batch, cell=self.cell.reshape(1, 3, 3), edge_index=ei,
                 edge_shift=sh)
        d._hn_edge_count = total
        e = self.model(d)
        f = -torch.autograd.grad(e.sum(), self.pos)[0]
        return e.detach(), f, total

    def stale(self):
        """True once the model's derived weight copies were dropped after the capture (`load_state_dict`, `.to()`,
        `train()` / `eval()`, `invalidate_caches()`): the captured launches read the OLD copies -- capture again."""
        return self.model.__dict__.get("_cache_epoch", 0) != self._epoch

    def _capture(self, capacity):
        self.capacity = capacity
        self._epoch = self.model.__dict__.get("_cache_epoch", 0)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self._warmup):        # caches (cell on the host, element counts, row layout), library state
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.energy, self.forces, self.total = self._eager()
            # everything a caller copies to the host per step, as ONE array (`fetch`): energies | (edges, flags) | forces
            self.packed = torch.cat([self.energy.double().reshape(-1), self.total.double(), self.forces.double().reshape(-1)])
        self._host = None
        torch.cuda.synchronize()

    def __call__(self, pos=None):
        """`pos` [N,3]: a device tensor, or a float32 host tensor (uploaded straight into the captured input)."""
        if pos is not None:
            with torch.no_grad():
                self.pos.copy_(pos)
        self.graph.replay()
        return self.energy, self.forces

    def fetch(self):
        """The last call's results on the host through ONE device-to-host copy and one synchronisation (separate reads of
        the energy, the forces and the list's counters cost a round trip each -- as much as the whole replay of a small
        cell): (energy [graphs] float32 array, forces [N,3] float32 array, list complete?, edges found)."""
        import numpy as np
        from .neighbor import _stash_overflowed
        if self._host is None:
            self._host = torch.empty(self.packed.shape, dtype=torch.float64).pin_memory()
        self._host.copy_(self.packed, non_blocking=True)
        torch.cuda.current_stream(self.packed.device).synchronize()
        h = self._host.numpy()
        ng = self.energy.numel()
        n_edges, flags = int(h[ng]), int(h[ng + 1])
        if flags & 2:
            _stash_overflowed(self.packed.device)        # (the repeat gets a larger stash slot per atom)
        return h[:ng].astype(np.float32), h[ng + 2:].astype(np.float32).reshape(-1, 3), flags == 0, n_edges

    def check(self):
        """(list complete?, edges found) of the last call: one host read."""
        from .neighbor import padded_list_ok
        return padded_list_ok(self.total)

    def recapture(self, pos=None, capacity=None):
        """A new graph for a larger capacity (default: from the last edge count); returns the step's results."""
        from .neighbor import padded_capacity
        if pos is not None:
            with torch.no_grad():
                self.pos.copy_(pos)
        if capacity is None:
            capacity = padded_capacity(max(int(self.total[0]), self.capacity))
        self._capture(int(capacity))
        return self.__call__()
