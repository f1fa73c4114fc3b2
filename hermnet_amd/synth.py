"""Deterministic synthetic inputs for BASELINE.json's configs.

Everything is `numpy.random.RandomState(seed)`; there is no network for datasets
and the reference ships none.  Used by `bench.py`, the tests and the golden
generator (`tests/golden/gen_golden.py`).
"""
import numpy as np
import torch

from .data import Data
from .neighbor import neighbor_list

_DIAMOND = np.array([(0, 0, 0), (0, .5, .5), (.5, 0, .5), (.5, .5, 0),
                     (.25, .25, .25), (.25, .75, .75), (.75, .25, .75), (.75, .75, .25)], dtype=np.float64)
_FCC = np.array([(0, 0, 0), (0, .5, .5), (.5, 0, .5), (.5, .5, 0)], dtype=np.float64)


def _lattice(basis, a, reps):
    nx, ny, nz = reps
    cells = np.stack(np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij"), -1).reshape(-1, 3)
    pos = (cells[:, None, :] + basis[None, :, :]).reshape(-1, 3) * a
    cell = np.diag([nx * a, ny * a, nz * a]).astype(np.float64)
    return pos, cell


def periodic_data(pos, cell, z, rc, reference_compat=False):
    """Assemble the `Data` the reference's `transform` would build (`HermNet/data.py:27-35`)
    for one periodic structure, with `batch` (required by `hermnet.py:130`)."""
    i, j, s = neighbor_list(pos, rc, cell)
    sign = 1.0 if reference_compat else -1.0
    return Data(
        pos=torch.from_numpy(pos.astype(np.float32)),
        atomic_number=torch.from_numpy(np.asarray(z, dtype=np.int64)),
        edge_index=torch.from_numpy(np.vstack([i, j])).long(),
        edge_shift=torch.from_numpy((sign * s).astype(np.float32)),
        cell=torch.from_numpy(cell.astype(np.float32)).reshape(1, 3, 3),
        batch=torch.zeros(len(pos), dtype=torch.long),
    )


def si_diamond(reps=(2, 2, 2), a=5.43, sigma=0.05, seed=0, rc=5.0, reference_compat=False):
    """C1: jittered diamond Si, 8*prod(reps) atoms."""
    pos, cell = _lattice(_DIAMOND, a, reps)
    rs = np.random.RandomState(seed)
    pos = pos + rs.normal(scale=sigma, size=pos.shape)
    z = np.full(len(pos), 14, dtype=np.int64)
    return periodic_data(pos, cell, z, rc, reference_compat)


def fcc_alloy_atoms(reps=(10, 10, 25), a=3.6, sigma=0.05, seed=0, species=(13, 28, 29)):
    """Coordinates, cell and species of the C2/C4 alloy (no neighbour list)."""
    pos, cell = _lattice(_FCC, a, reps)
    rs = np.random.RandomState(seed)
    pos = pos + rs.normal(scale=sigma, size=pos.shape)
    pos = pos - np.floor(pos / np.diag(cell)) * np.diag(cell)
    z = np.asarray(species, dtype=np.int64)[rs.randint(0, len(species), size=len(pos))]
    return pos, cell, z


def fcc_alloy(reps=(10, 10, 25), a=3.6, sigma=0.05, seed=0, rc=5.0, species=(13, 28, 29),
              reference_compat=False, device=None):
    """C2/C4: jittered fcc alloy, 4*prod(reps) atoms, wrapped into the cell;
    species drawn uniformly after the jitter draw from the same RandomState.
    `device`: build the neighbour list with the device cell list (identical result, much faster for
    the 80k-100k atom cells) and return the `Data` on that device."""
    pos, cell, z = fcc_alloy_atoms(reps, a, sigma, seed, species)
    if device is None:
        return periodic_data(pos, cell, z, rc, reference_compat)
    from .neighbor import neighbor_search
    pos_t = torch.from_numpy(pos.astype(np.float32)).to(device)
    cell_t = torch.from_numpy(cell.astype(np.float32)).to(device)
    ei, sh = neighbor_search(pos_t, rc, cell_t, reference_compat=reference_compat)
    return Data(pos=pos_t, atomic_number=torch.from_numpy(z).to(device), edge_index=ei, edge_shift=sh,
                cell=cell_t.reshape(1, 3, 3), batch=torch.zeros(len(pos), dtype=torch.long, device=device))


def molecule_batch(num_graphs=1024, nmin=9, nmax=30, species=(1, 6, 8), radius=3.0, min_sep=0.9,
                   seed=0, rc=5.0):
    """C5: batch of open-boundary molecules (rejection-sampled points in a ball),
    edges = all ordered pairs closer than rc, one concatenated `Data` with `batch`."""
    rs = np.random.RandomState(seed)
    pos_l, z_l, b_l, ei_l = [], [], [], []
    off = 0
    for g in range(num_graphs):
        n = int(rs.randint(nmin, nmax + 1))
        pts = []
        while len(pts) < n:
            p = rs.uniform(-radius, radius, size=3)
            if np.dot(p, p) > radius * radius:
                continue
            if pts and np.min(np.linalg.norm(np.asarray(pts) - p, axis=1)) < min_sep:
                continue
            pts.append(p)
        pts = np.asarray(pts)
        d = np.linalg.norm(pts[:, None, :] - pts[None, :, :], axis=-1)
        tgt, src = np.nonzero((d < rc) & ~np.eye(n, dtype=bool))
        ei_l.append(np.vstack([src, tgt]) + off)
        pos_l.append(pts)
        z_l.append(np.asarray(species, dtype=np.int64)[rs.randint(0, len(species), size=n)])
        b_l.append(np.full(n, g, dtype=np.int64))
        off += n
    return Data(
        pos=torch.from_numpy(np.concatenate(pos_l).astype(np.float32)),
        atomic_number=torch.from_numpy(np.concatenate(z_l)),
        edge_index=torch.from_numpy(np.concatenate(ei_l, axis=1)).long(),
        batch=torch.from_numpy(np.concatenate(b_l)),
    )


def synth_state_dict(reference_state_dict, seed=0):
    """Deterministic, construction-order-independent weights for parity runs.

    Takes any state_dict with the reference's key layout
    and returns one with the same keys/shapes whose values depend only on
    (sorted key position, shape, seed).  Linear weights/biases ~ U(-1/sqrt(fan_in), +),
    LayerNorm affine is made non-trivial, buffers such as
    `radial_basis.rbf.offset` are kept (they are defined by hyper-parameters).
    """
    out = {}
    for n, key in enumerate(sorted(reference_state_dict.keys())):
        v = reference_state_dict[key]
        g = torch.Generator().manual_seed(seed * 100003 + n)
        if key.endswith("rbf.offset"):
            out[key] = v.clone()
        elif key == "embed.weight":
            out[key] = torch.randn(v.shape, generator=g)
        elif "x_layernorm" in key:
            r = 0.1 * torch.randn(v.shape, generator=g)
            out[key] = (1.0 + r) if key.endswith("weight") else r
        elif v.dim() == 2:
            bound = 1.0 / (v.shape[1] ** 0.5)
            out[key] = (torch.rand(v.shape, generator=g) * 2 - 1) * bound
        elif v.dim() == 1:
            # bias: the fan-in of the matching weight
            w = reference_state_dict.get(key[:-4] + "weight")
            bound = 1.0 / (w.shape[1] ** 0.5) if w is not None and w.dim() == 2 else 0.1
            out[key] = (torch.rand(v.shape, generator=g) * 2 - 1) * bound
        else:
            out[key] = v.clone()
    return out
