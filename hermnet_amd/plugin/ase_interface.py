"""ASE calculator adapter: the intended API of `plugin/ase_interface/calculator.py:10-98`.

The reference bodies cannot run as written (SURVEY.md section 8(b) lists the defects: missing
`model.rc`, `torch.from_numpy` on a Tensor, `cell[0]` on a 3x3 cell, no `batch`, the Voigt vector
taking `[0,1]` for `xx`); this module reproduces the intent, not the defects.
"""
import numpy as np
import torch

from ..data import Data, neighbor_search
from ..elements import atomic_numbers
from ..utils import virial_calc

try:  # ASE is optional on the MI355X image
    from ase.calculators.calculator import Calculator as _Base, all_changes
except Exception:  # pragma: no cover - exercised on the GPU image
    all_changes = ['positions', 'numbers', 'cell', 'pbc', 'initial_charges', 'initial_magmoms']

    class _Base(object):
        """Minimal stand-in for `ase.calculators.calculator.Calculator`."""

        def __init__(self, **kwargs):
            self.results = {}
            self.atoms = None

        def calculate(self, atoms=None, properties=('energy',), system_changes=all_changes):
            self.atoms = atoms


_SPECIES = {}      # device -> (numpy Z, z tensor, batch tensor) of the last call


def _species_tensors(elements, device):
    """Atomic numbers and the (single-graph) batch vector as device tensors, REUSED while the species list is
    unchanged: along a trajectory only the coordinates change, and the relation build skips its one host sync
    (element counts) when it is handed the same tensor objects again (`relations._COUNT_CACHE`)."""
    elements = np.asarray(elements)
    key = str(device)
    hit = _SPECIES.get(key)
    if hit is None or hit[0].shape != elements.shape or not np.array_equal(hit[0], elements):
        z = torch.from_numpy(elements.copy()).long()
        batch = torch.zeros(elements.shape[0], dtype=torch.long)
        if device is not None:
            z, batch = z.to(device), batch.to(device)
        hit = _SPECIES[key] = (elements.copy(), z, batch)
    return hit[1], hit[2]


_CELLS = {}      # device -> (numpy cell, tensor) of the last call


def _cell_tensor(cell, device):
    """The cell as a [3,3] float32 tensor on `device`, REUSED while its values are unchanged (NVT / NVE: the neighbour
    search then skips reading it back, `neighbor._cell_on_host`, and a captured step keeps seeing the same tensor)."""
    c = np.asarray(cell, dtype=np.float64).reshape(3, 3)
    key = str(device)
    hit = _CELLS.get(key)
    if hit is None or not np.array_equal(hit[0], c):
        hit = _CELLS[key] = (c.copy(), torch.from_numpy(c).float().to(device))
    return hit[1]


def build_graph(cell, elements, pos, rc, device=None, reference_compat=False, capacity=None):
    """`calculator.py:10-27`: numpy (cell [3,3] or None, Z [N], pos [N,3]) -> `Data` with the cutoff graph.
    `capacity` (periodic cells on the GPU): the neighbour list is built WITHOUT its host read, padded to that many columns
    (`neighbor.neighbor_search_padded`); `data._hn_edge_count` then holds (E, flags) on the device for a check behind
    the step (`NNCalculator.calculate` does it when the results are copied to the host anyway).
    With `device` set to the GPU the coordinates are uploaded first and the neighbour search runs on the
    device (`csrc/neighbor_kernels.hip`) instead of the host -- the reference rebuilds the list every step.
    `reference_compat`: the edge conventions of the reference's own pipeline (`neighbor.neighbor_search`),
    for checkpoints trained through it on periodic data."""
    pos_t = torch.from_numpy(np.asarray(pos)).float()
    if device is not None:
        pos_t = pos_t.to(device)
    z, batch = _species_tensors(elements, device)
    data = Data(atomic_number=z, pos=pos_t, batch=batch)
    if cell is None or not np.any(np.asarray(cell)):
        data.edge_index = neighbor_search(pos=pos_t, rc=rc, reference_compat=reference_compat)
    else:
        cell_t = _cell_tensor(cell, pos_t.device)
        if capacity is not None and pos_t.is_cuda:
            from ..neighbor import neighbor_search_padded
            data.edge_index, data.edge_shift, data._hn_edge_count = neighbor_search_padded(
                pos_t, rc, cell_t, capacity, reference_compat=reference_compat)
        else:
            data.edge_index, data.edge_shift = neighbor_search(pos=pos_t, rc=rc, cell=cell_t,
                                                                reference_compat=reference_compat)
        data.cell = cell_t.reshape(1, 3, 3)
    return data


def _evaluate(model, data, device, pbc, want_virial, trn_mean=0.0):
    """One energy + force evaluation -> (energy [graphs] tensor, forces [N,3] tensor, W [3,3] tensor in energy units
    or None).  W = -dE/d(strain) = sum_i pos_i (x) F_i - cell^T dE/dcell, symmetrised (`utils.virial_calc` with unit
    factor 1); the cell gradient comes from the edge geometry kernel's backward."""
    device = torch.device(device)
    data = data.to(device)
    data.pos.requires_grad = True
    periodic = bool(pbc) and data.get('cell') is not None
    if want_virial and periodic:
        data.cell.requires_grad = True
    if model.training:      # (not unconditionally: `eval()` walks every submodule -- 0.85 ms per call for a 5-layer model)
        model.eval()
    energy = model(data) + trn_mean
    forces = -torch.autograd.grad(energy.sum(), data.pos, retain_graph=want_virial and periodic)[0]
    w = None
    if want_virial:
        w = virial_calc(cell=data.get('cell'), pos=data.pos.detach(), forces=forces, energy=energy, units='lj',
                        pbc=periodic).detach()
    return energy.detach(), forces.detach(), w


def _evaluate_finite(model, data, device, pbc, want_virial, trn_mean=0.0):
    """`_evaluate` + the energy on the host (every caller copies it there anyway), never NaN: the stale-cache guard
    (`guard.ParamGuard`) answers a write through `.data` behind the cached kernel-ready weights with a NaN step, and ONE NaN
    force step handed to an integrator corrupts a trajectory for good.  A non-finite energy therefore drops the caches,
    evaluates again with the current weights (warning once per model), and raises if the result is still not finite."""
    for attempt in (0, 1):
        energy, forces, w = _evaluate(model, data, device, pbc, want_virial, trn_mean)
        e_host = energy.cpu()
        if bool(torch.isfinite(e_host).all()):
            return e_host, forces, w
        if attempt == 0 and hasattr(model, "invalidate_caches"):
            _warn_repaired(model)
            model.invalidate_caches()
    raise FloatingPointError("hermnet_amd: the energy is not finite (%r) also after rebuilding the cached weights: "
                             "refusing to hand NaN energies / forces to the caller" % (e_host.tolist(),))


def _warn_repaired(model):
    if not model.__dict__.get("_warned_nan_repair"):
        model.__dict__["_warned_nan_repair"] = True
        import warnings
        warnings.warn("hermnet_amd: a step came back NaN (weights written through `.data` behind the cached copies?); the "
                      "caches were rebuilt and the step re-evaluated.  Call model.invalidate_caches() after such writes.",
                      RuntimeWarning, stacklevel=3)


def model_calc(model, data, device, pbc, ensemble='NVT', trn_mean=0.0, units='metal'):
    """`calculator.py:59-98` / `lmp_calc.py:36-85`: (energy float, forces [N,3] float32, virial [6]).

    virial (NPT only) is the symmetrised pressure*volume tensor of `virial_calc` in the order
    [xx, yy, zz, xy, xz, yz] -- what LAMMPS `fix client/md` expects (`lmp_calc.py:58-67,232-235`); zeros for NVT like
    the reference.  This is the LAMMPS packing; ASE's `stress` is made from the same tensor in
    `NNCalculator.calculate` (`stress_from_virial`)."""
    from ..utils import _NKTV2P
    if units not in _NKTV2P:
        raise ValueError('Illegal units command')
    npt = ensemble.lower() == 'npt'
    energy, forces, w = _evaluate_finite(model, data, device, pbc, npt, trn_mean)
    if npt:
        v = (w * _NKTV2P[units]).cpu().numpy()
        virial = np.array([v[0, 0], v[1, 1], v[2, 2], v[0, 1], v[0, 2], v[1, 2]])
    else:
        virial = np.zeros(6, dtype=np.float32)
    return energy.item(), forces.cpu().numpy().reshape(-1, 3), virial


def stress_from_virial(w, volume):
    """ASE's contract for `results['stress']`: sigma = (1/V) dE/d(strain) = -W / V in eV/A^3 (W in eV), Voigt order
    [xx, yy, zz, yz, xz, xy] (`ase.calculators.calculator`: cell filters and NPT dynamics consume it in that form).
    The reference hands ASE the LAMMPS virial instead (`calculator.py:85-97`: pressure*volume units, LAMMPS order,
    opposite sign, `[0,1]` where `[0,0]` is meant) -- SURVEY 8(b) lists it under "reproduce the intent"."""
    s = -np.asarray(w, dtype=np.float64) / float(volume)
    return np.array([s[0, 0], s[1, 1], s[2, 2], s[1, 2], s[0, 2], s[0, 1]])


class NNCalculator(_Base):
    """`calculator.py:30-57`.  `model_path=None` keeps the weights already in `model`."""
    implemented_properties = ['energy', 'free_energy', 'forces', 'stress']

    def __init__(self, model, model_path, trn_mean, device_='cuda', ensemble='NVT', reference_compat=False,
                 graph_replay=False):
        """`graph_replay=True` (GPU, periodic cell, forces without stress): neighbour search + relation build + forward +
        force backward are captured ONCE as a hipGraph and replayed every call (`graph.GraphedMDStep`; the same kernels, no
        per-launch host work: 2-3x on cells of a few hundred atoms, where the eager step is bound by launch overhead).
        The capture is renewed by itself when the species, the cell, the atom count or the model's weights
        (`load_state_dict`, `.to()`, `invalidate_caches()`) change or the list outgrows its capacity; weights written
        through `.data` need `model.invalidate_caches()` as everywhere else.  Calls that need the stress, open systems and
        CPU runs take the eager path."""
        super(NNCalculator, self).__init__()
        self.reference_compat = reference_compat     # see `build_graph`
        self.graph_replay = bool(graph_replay)
        self._graphed = None              # ((z tensor, cell tensor, N), GraphedMDStep)
        self.graph_captures = 0           # how often a step was captured (diagnostics / tests)
        self.device_ = device_
        device = torch.device(device_)
        self.model = model.to(device)
        if model_path is not None:
            self.model.load_state_dict(torch.load(model_path, map_location=device))
        for p in self.model.parameters():       # energy/force evaluation only
            p.requires_grad_(False)
        self.trn_mean = trn_mean
        self.ensemble = ensemble
        self._edge_capacity = None        # columns of the padded neighbour list of the next call (None: exact search)

    def _replayed(self, cell, elems, positions):
        """(energy tensor, forces tensor) from the captured step, or None when this call has to run eagerly."""
        from ..graph import GraphedMDStep
        dev = torch.device(self.device_)
        z, _batch = _species_tensors(elems, dev)
        cell_t = _cell_tensor(cell, dev)
        pos_t = torch.from_numpy(np.ascontiguousarray(positions, dtype=np.float32))     # host: copied into the captured input
        g = self._graphed
        if (g is None or g[0][0] is not z or g[0][1] is not cell_t or g[0][2] != pos_t.size(0) or g[1].stale()
                or g[1].model is not self.model):
            if self.model.training:
                self.model.eval()
            step = GraphedMDStep(self.model, z, cell_t, pos_t.to(dev), reference_compat=self.reference_compat)
            self._graphed = g = ((z, cell_t, pos_t.size(0)), step)
            self.graph_captures += 1
        g[1](pos_t)
        e, f, ok, _n = g[1].fetch()            # ONE device-to-host copy: energy, forces and the list's counters
        if not ok:                             # the list outgrew its columns: a larger capture, this step again
            g[1].recapture(pos_t)
            self.graph_captures += 1
            e, f, ok, _n = g[1].fetch()
            if not ok:                         # (coordinates many images outside the cell ...): the eager path decides
                self._graphed = None
                return None
        if not np.isfinite(e[0]):
            # The captured step contains the stale-cache guard's check-and-poison kernel, and a replay cannot read its flag:
            # after a write through `.data` (an EMA swap) EVERY replay would return NaN.  The results are on the host here:
            # a NaN drops the caches and the capture, and the eager path evaluates this call again (and raises if the
            # energy is still not finite) -- NaN forces never reach ASE.
            _warn_repaired(self.model)
            self.model.invalidate_caches()
            self._graphed = None
            return None
        return float(np.float32(e[0]) + np.float32(self.trn_mean)), f

    def calculate(self, atoms, properties=('energy',), system_changes=all_changes):
        super(NNCalculator, self).calculate(atoms, properties, system_changes)
        pbc = bool(np.any(atoms.pbc))
        cell = np.asarray(atoms.cell if not hasattr(atoms, "todict") else atoms.todict()['cell']) if pbc else None
        # (`atoms.numbers` is what ASE keeps; the symbols -- `calculator.py:46` -- cost a dictionary lookup per atom and call)
        numbers = getattr(atoms, "numbers", None)
        elems = (np.asarray(numbers) if numbers is not None
                 else np.array([atomic_numbers[s] for s in atoms.get_chemical_symbols()]))
        dev = self.device_ if torch.device(self.device_).type == 'cuda' else None
        if (self.graph_replay and dev is not None and pbc and cell is not None and np.any(cell)
                and self.ensemble.lower() != 'npt' and 'stress' not in tuple(properties)):
            out = self._replayed(cell, elems, atoms.positions)
            if out is not None:
                self.results['energy'] = out[0]
                self.results['free_energy'] = out[0]
                self.results['forces'] = out[1]
                self.results.pop('stress', None)      # (not computed: see the end of this method)
                return
        # periodic cells on the GPU: from the second call on the neighbour list is built without its host read, padded to a
        # capacity taken from the last edge count; count and flags are checked behind the step, where the results are
        # copied to the host anyway (an overflow repeats the step on an exact list)
        cap = self._edge_capacity if (dev is not None and cell is not None) else None
        data = build_graph(cell=cell, elements=elems, pos=atoms.positions, rc=self.model.rc, device=dev,
                           reference_compat=self.reference_compat, capacity=cap)
        # stress whenever ASE asks for it (or the ensemble is NPT) on a periodic cell; an open system has none
        want = pbc and cell is not None and (self.ensemble.lower() == 'npt' or 'stress' in tuple(properties))
        energy, forces, w = _evaluate_finite(self.model, data, self.device_, pbc, want, self.trn_mean)
        energy = energy.item()
        if dev is not None and cell is not None:
            from ..neighbor import padded_capacity, padded_list_ok
            if cap is not None:
                ok, n_edges = padded_list_ok(data._hn_edge_count)
                if not ok or n_edges > cap:            # the padded list was incomplete: this step again, exactly
                    self._edge_capacity = None
                    return self.calculate(atoms, properties, system_changes)
            else:
                n_edges = int(data.edge_index.size(1))
            # (kept while it still fits with a margin and is not wastefully large: a stable launch geometry)
            if cap is None or not (n_edges * 1.02 + 64 <= cap <= n_edges * 1.25 + 8192):
                cap = padded_capacity(n_edges)
            self._edge_capacity = cap
        self.results['energy'] = energy
        self.results['free_energy'] = energy
        self.results['forces'] = forces.cpu().numpy().reshape(-1, 3)
        if want:
            volume = abs(float(np.linalg.det(np.asarray(cell, dtype=np.float64).reshape(3, 3))))
            self.results['stress'] = stress_from_virial(w.cpu().numpy(), volume)
        elif not (pbc and cell is not None):
            self.results['stress'] = np.zeros(6)          # an open system has no stress: zeros, as the reference stores
        else:
            # Periodic cell, stress not asked for in this call: NOT stored.  ASE's `calculation_required` only looks whether
            # a property's name is in `results`, so a cached zero vector would be handed to a later `atoms.get_stress()`
            # (cell filters, NPT dynamics) without a recomputation; absent, ASE calls `calculate(['stress'])` again.
            self.results.pop('stress', None)

    def model_calc(self, data, device, pbc, ensemble='NVT'):
        """`calculator.py:59-98`: (energy, forces [N,3], LAMMPS-packed virial [6]) -- see the module-level function."""
        return model_calc(self.model, data, device, pbc, ensemble, self.trn_mean)
