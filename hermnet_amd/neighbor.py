"""Host-side cutoff neighbour search (input producer of the hot path).

Replaces `HermNet/data.py:14-24` (`neighbor_search`), which delegates to
`ase.neighborlist.primitive_neighbor_list('ijS', ...)` for periodic cells and to
`torch_cluster.radius_graph` otherwise -- neither package exists on the MI355X
image, so the build owns this step.

Conventions:
  * pair (i, j, S) is listed iff |pos[j] - pos[i] + S @ cell| < rc  (strict),
    (i, i, 0) is never listed, self-images (i, i, S != 0) are;
  * output order is canonical: lexicographic in (i, j, Sx, Sy, Sz);
  * indices are int64 (what the reference hands to the model), shifts int.

`neighbor_search` wraps the raw list in the reference's calling convention:
`edge_index = [i; j]` and an `edge_shift` such that the model-side formula
(`HermNet/hermnet.py:135-139`: pos[ei0] - pos[ei1] + edge_shift @ cell) yields
the true minimum-image vector, i.e. edge_shift = -S.  `reference_compat=True`
reproduces `data.py:19-24` literally (edge_shift = +S, the sign quirk described
in SURVEY.md section 0) for pipeline-level parity tests.
"""
import numpy as np
import os

import torch

from . import switches


def _expand_ranges(start, end):
    """Concatenate aranges [start_k, end_k) -> (flat_index, owner_k)."""
    cnt = (end - start).astype(np.int64)
    total = int(cnt.sum())
    if total == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    owner = np.repeat(np.arange(len(cnt), dtype=np.int64), cnt)
    first = np.cumsum(cnt) - cnt
    flat = np.arange(total, dtype=np.int64) - np.repeat(first, cnt) + np.repeat(start.astype(np.int64), cnt)
    return flat, owner


def neighbor_list(pos, rc, cell=None, pbc=(True, True, True), chunk=200000):
    """Cell-list neighbour search in float64.  Returns (i, j, S) canonical-sorted.

    pos [N,3]; cell [3,3] (rows are lattice vectors) or None for an open system.
    """
    pos = np.asarray(pos, dtype=np.float64).reshape(-1, 3)
    n = pos.shape[0]
    rc = float(rc)
    if n == 0:
        z = np.zeros(0, np.int64)
        return z, z, np.zeros((0, 3), np.int64)

    if cell is None:
        pts, owner, img = pos, np.arange(n, dtype=np.int64), np.zeros((n, 3), np.int64)
        wrap = np.zeros((n, 3), np.int64)
        real = pos
    else:
        cell = np.asarray(cell, dtype=np.float64).reshape(3, 3)
        pbc = np.asarray(pbc, dtype=bool).reshape(3)
        inv = np.linalg.inv(cell)
        frac = pos @ inv
        wrap = np.where(pbc, np.floor(frac), 0.0)
        fw = frac - wrap
        wrap = wrap.astype(np.int64)
        real = fw @ cell
        # plane spacings decide how many periodic images reach into the cutoff sphere
        heights = 1.0 / np.linalg.norm(inv, axis=0)
        nimg = np.where(pbc, np.ceil(rc / heights).astype(np.int64), 0)
        margin = rc / heights
        pts_l, owner_l, img_l = [], [], []
        ar = np.arange(n, dtype=np.int64)
        for sx in range(-nimg[0], nimg[0] + 1):
            for sy in range(-nimg[1], nimg[1] + 1):
                for sz in range(-nimg[2], nimg[2] + 1):
                    s = np.array([sx, sy, sz], dtype=np.float64)
                    f = fw + s
                    keep = np.all((f >= -margin - 1e-9) & (f <= 1.0 + margin + 1e-9) | ~pbc, axis=1)
                    if not keep.any():
                        continue
                    pts_l.append(f[keep] @ cell)
                    owner_l.append(ar[keep])
                    img_l.append(np.broadcast_to(np.array([sx, sy, sz], np.int64), (int(keep.sum()), 3)))
        pts = np.concatenate(pts_l)
        owner = np.concatenate(owner_l)
        img = np.concatenate(img_l)

    # uniform Cartesian bins of edge >= rc over the bounding box of all points
    lo = pts.min(axis=0) - 1e-6
    nb = np.maximum(((pts.max(axis=0) - lo) / rc).astype(np.int64) + 1, 1)
    pb = np.minimum(((pts - lo) / rc).astype(np.int64), nb - 1)
    pid = (pb[:, 0] * nb[1] + pb[:, 1]) * nb[2] + pb[:, 2]
    order = np.argsort(pid, kind="stable")
    pid_s = pid[order]
    rb = np.minimum(((real - lo) / rc).astype(np.int64), nb - 1)

    out_i, out_j, out_s = [], [], []
    rc2 = rc * rc
    for c0 in range(0, n, chunk):
        c1 = min(n, c0 + chunk)
        ids = np.arange(c0, c1, dtype=np.int64)
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                bx = rb[c0:c1, 0] + dx
                by = rb[c0:c1, 1] + dy
                ok = (bx >= 0) & (bx < nb[0]) & (by >= 0) & (by < nb[1])
                if not ok.any():
                    continue
                # the three z-neighbour bins are contiguous in the sorted id space
                z0 = np.maximum(rb[c0:c1, 2] - 1, 0)
                z1 = np.minimum(rb[c0:c1, 2] + 1, nb[2] - 1)
                base = (bx * nb[1] + by) * nb[2]
                st = np.searchsorted(pid_s, base + z0, side="left")
                en = np.searchsorted(pid_s, base + z1, side="right")
                st = np.where(ok, st, 0)
                en = np.where(ok, en, 0)
                flat, own = _expand_ranges(st, en)
                if flat.size == 0:
                    continue
                cand = order[flat]
                ii = ids[own]
                d = pts[cand] - real[ii]
                d2 = np.einsum("ij,ij->i", d, d)
                sel = d2 < rc2
                jj = owner[cand]
                ss = img[cand]
                sel &= ~((jj == ii) & np.all(ss == 0, axis=1))
                out_i.append(ii[sel])
                out_j.append(jj[sel])
                out_s.append(ss[sel])
    if not out_i:
        z = np.zeros(0, np.int64)
        return z, z, np.zeros((0, 3), np.int64)
    i = np.concatenate(out_i)
    j = np.concatenate(out_j)
    s = np.concatenate(out_s)
    # shifts were found for wrapped coordinates; express them for the caller's positions
    s = s - wrap[j] + wrap[i]
    key = np.lexsort((s[:, 2], s[:, 1], s[:, 0], j, i))
    return i[key], j[key], s[key]


_CELL_HOST = []        # [(cell tensor, version, 9 doubles)]: the cell of an MD run is the same tensor step after step


def _cell_on_host(cell):
    """The 3x3 cell as host doubles; read back once per tensor object and version (the read is a host sync)."""
    for ent in _CELL_HOST:
        if ent[0] is cell and ent[1] == cell._version:
            return ent[2]
    vals = cell.detach().double().cpu().reshape(-1, 3, 3)[0].contiguous().reshape(-1).tolist()
    _CELL_HOST.insert(0, (cell, cell._version, vals))
    del _CELL_HOST[4:]
    return vals


_STASH = {}       # device -> keys per atom of the search's stash slot (csrc/neighbor_kernels.hip)
_STASH_DEFAULT = 96       # fcc at rc = 5 A: 43 pairs per atom; 8 bytes per key and atom of workspace


def _stash_workspace_bytes(lib, N, dev):
    """Workspace of a search whose per-atom stash slot holds the current hint of this device (the slot size follows from
    the workspace size: include/hermnet_hip.h, hermnet_neighbor_workspace_for)."""
    return lib.hermnet_neighbor_workspace_for(N, _STASH.get(str(dev), _STASH_DEFAULT))


def _stash_overflowed(dev):
    """An atom had more pairs than its slot (flag bit 1): the following searches get the largest slot."""
    _STASH[str(dev)] = 160


def _neighbor_search_device(pos, rc, cell, reference_compat, target_mask=None):
    """Device cell list (`csrc/neighbor_kernels.hip`); same result as the host path, tensors stay on the GPU.
    `target_mask` [N] bool/uint8: list only the pairs whose target atom (row 1) is flagged (atom shards)."""
    import ctypes
    from . import _lib
    lib = _lib.load()
    P = _lib.ptr
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    dev = pos.device
    p32 = pos.detach().float().contiguous()
    N = int(p32.size(0))
    dbl3 = ctypes.c_double * 3
    cell_h = lo_h = hi_h = None
    if cell is not None:
        cell_h = (ctypes.c_double * 9)(*_cell_on_host(cell))
    elif N > 0:
        mm = torch.stack([p32.min(0).values, p32.max(0).values]).double().cpu().tolist()
        lo_h, hi_h = dbl3(*mm[0]), dbl3(*mm[1])
    else:
        lo_h, hi_h = dbl3(0, 0, 0), dbl3(1, 1, 1)
    mask = None
    if target_mask is not None:
        mask = target_mask if target_mask.dtype == torch.uint8 else target_mask.to(torch.uint8)
        mask = mask.contiguous()
        if mask.numel() != N or mask.device != dev:
            raise ValueError("target_mask must be [N] on the device of pos")
    ws_bytes = _stash_workspace_bytes(lib, N, dev)
    work = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    total = torch.zeros(2, dtype=torch.long, device=dev)
    args = (P(p32), N, cell_h, lo_h, hi_h, float(rc), P(work), ws_bytes)
    _lib.check(lib.hermnet_neighbor_count(*args, P(mask), P(total), stream), "hermnet_neighbor_count")
    E, flags = total.tolist()                                # the one host read of the search
    if flags & 2:
        _stash_overflowed(dev)                               # (this search finishes in its two-pass form)
    if flags & 1:                                            # an image shift beyond +-8 cells: the host path handles it
        return None
    edge_index = torch.empty(2, E, dtype=torch.long, device=dev)
    periodic = cell is not None
    shift = torch.empty(E, 3, dtype=torch.float32, device=dev) if periodic else None
    if E > 0:
        stash_ok = 0 if (flags & 2) else 1
        keys = None if stash_ok else torch.empty(E, dtype=torch.long, device=dev)
        sign = 1.0 if reference_compat else -1.0
        _lib.check(lib.hermnet_neighbor_fill(*args, E, sign, 0 if periodic else 1, stash_ok, P(keys), P(mask),
                                             P(edge_index), P(shift), stream), "hermnet_neighbor_fill")
    return (edge_index, shift) if periodic else edge_index


def neighbor_search_padded(pos, rc, cell, capacity, reference_compat=False, target_mask=None):
    """The device cell list WITHOUT its host read: `capacity` columns are provided up front, the pairs
    found fill the first E of them and the rest become NULL edges (-1, -1; shift 0), which the relation build files behind
    every row -- the model runs on the padded list unchanged, with a launch geometry that does not depend on E.

    Returns (edge_index [2, capacity] int64, edge_shift [capacity, 3] float32 or None, total [2] int64 ON THE DEVICE):
    total[0] = E, total[1] = flags; read them when the step's results are copied to the host anyway.  The list is
    complete iff `padded_list_ok(total)`; otherwise (more pairs than columns, an atom with more pairs than its stash slot,
    coordinates many cells away from the cell) repeat with `neighbor_search` and a larger capacity.  GPU tensors only;
    open systems need `reference_compat=False` (the 32-neighbour cap is a host-side filter).
    `target_mask` [N] bool / uint8 (atom shards): list only the pairs whose target atom (row 1) is flagged."""
    import ctypes
    from . import _lib
    if not pos.is_cuda:
        raise RuntimeError("neighbor_search_padded runs on the device list only")
    if cell is None and reference_compat:
        raise NotImplementedError("the reference pipeline's 32-neighbour cap needs the exact list (neighbor_search)")
    lib = _lib.load()
    P = _lib.ptr
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    dev = pos.device
    p32 = pos.detach().float().contiguous()
    N, cap = int(p32.size(0)), int(capacity)
    dbl3 = ctypes.c_double * 3
    cell_h = lo_h = hi_h = None
    if cell is not None:
        cell_h = (ctypes.c_double * 9)(*_cell_on_host(cell))
    else:       # (an open system's bounding box is a host read of its own: periodic cells are the MD case)
        mm = torch.stack([p32.min(0).values, p32.max(0).values]).double().cpu().tolist()
        lo_h, hi_h = dbl3(*mm[0]), dbl3(*mm[1])
    mask = None
    if target_mask is not None:
        mask = (target_mask if target_mask.dtype == torch.uint8 else target_mask.to(torch.uint8)).contiguous()
        if mask.numel() != N or mask.device != dev:
            raise ValueError("target_mask must be [N] on the device of pos")
    ws_bytes = _stash_workspace_bytes(lib, N, dev)
    work = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    total = torch.empty(2, dtype=torch.long, device=dev)
    edge_index = torch.empty(2, cap, dtype=torch.long, device=dev)
    periodic = cell is not None
    shift = torch.empty(cap, 3, dtype=torch.float32, device=dev) if periodic else None
    if switches.debug_poison():
        # (tests: a column the search leaves unwritten would send the relation build far out of bounds)
        edge_index.fill_(0x3f3f3f3f3f3f3f3f)
        if shift is not None:
            shift.fill_(float("nan"))
    _lib.check(lib.hermnet_neighbor_count(P(p32), N, cell_h, lo_h, hi_h, float(rc), P(work), ws_bytes, P(mask), P(total), stream),
               "hermnet_neighbor_count")
    sign = 1.0 if reference_compat else -1.0
    _lib.check(lib.hermnet_neighbor_fill_padded(N, P(work), ws_bytes, cap, sign, 0 if periodic else 1, P(edge_index), P(shift),
                                                P(total), stream), "hermnet_neighbor_fill_padded")
    return edge_index, shift, total


def padded_list_ok(total):
    """(complete?, E) of a padded list from its `total` tensor -- a host read: do it behind the step."""
    E, flags = total.tolist()
    if flags & 2:
        _stash_overflowed(total.device)           # an atom had more pairs than its stash slot: the repeat gets a larger one
    return flags == 0, int(E)


def padded_capacity(num_edges, margin=0.04, granule=4096):
    """A column count for the next steps' lists: the last count plus a margin, rounded up (a stable launch geometry)."""
    want = int(num_edges * (1.0 + margin)) + 64
    return (want + granule - 1) // granule * granule


def _cap_neighbors(edge_index, cap):
    """Keep at most `cap` in-edges per target (row 1), the ones with the lowest source index: the list is sorted
    by (target, source), so an edge's rank inside its target's run is its position minus the run's start."""
    E = edge_index.size(1)
    if E == 0:
        return edge_index
    tgt = edge_index[1]
    pos = torch.arange(E, device=tgt.device)
    first = torch.ones(E, dtype=torch.bool, device=tgt.device)
    first[1:] = tgt[1:] != tgt[:-1]
    start = torch.cummax(torch.where(first, pos, torch.zeros_like(pos)), 0).values
    return edge_index[:, (pos - start) < cap]


def neighbor_search(pos, rc, cell=None, reference_compat=False, target_mask=None):
    """Drop-in for `HermNet/data.py:14-24`.

    pos: float Tensor [N,3]; cell: Tensor [3,3] or [1,3,3] or None.
    Returns `edge_index` (open system) or `(edge_index, edge_shift)` (periodic),
    int64 [2,E] / float32 [E,3], exactly the reference's return shapes.  GPU tensors are searched
    on the GPU and the result stays there; host tensors take the numpy cell list below.

    `reference_compat=True` reproduces the reference pipeline's edge conventions instead of the true
    minimum-image ones: periodic `edge_shift = +S` (`data.py:19-24`, see the module docstring) and, for open
    systems, `radius_graph`'s default cap of 32 neighbours per atom (`data.py:16`; torch_cluster keeps an
    implementation-defined subset -- its GPU kernel the 32 lowest source indices, which is what is kept here).

    `target_mask` [N] bool (no reference counterpart; `sharding.py`): keep only the edges whose target atom
    (`edge_index[1]`) is flagged, same order -- the list of an atom shard, without building the rest.
    """
    if cell is None and reference_compat:
        return _cap_neighbors(neighbor_search(pos, rc, None, False, target_mask), 32)
    if pos.is_cuda:
        out = _neighbor_search_device(pos, rc, cell, reference_compat, target_mask)
        if out is not None:
            return out
    dev = pos.device
    p = pos.detach().cpu().numpy()
    keep = None if target_mask is None else target_mask.detach().cpu().numpy().astype(bool)
    if cell is None:
        i, j, _ = neighbor_list(p, rc, None)
        if keep is not None:
            sel = keep[i]
            i, j = i[sel], j[sel]
        # radius_graph convention: row 0 = source (neighbour), row 1 = target (centre)
        return torch.from_numpy(np.vstack([j, i])).long().to(dev)
    c = cell.detach().cpu().numpy().reshape(-1, 3, 3)[0]
    i, j, s = neighbor_list(p, rc, c)
    if keep is not None:
        sel = keep[j]
        i, j, s = i[sel], j[sel], s[sel]
    edge_index = torch.from_numpy(np.vstack([i, j])).long()
    sign = 1.0 if reference_compat else -1.0
    edge_shift = torch.from_numpy(sign * s.astype(np.float32)).float()
    return edge_index.to(dev), edge_shift.to(dev)
