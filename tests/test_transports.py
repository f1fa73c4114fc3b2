"""The calculator transports of the reference (`plugin/lmp_interface/lmp_calc.py:135-238`,
`plugin/i-pi_interface/ipi_calc.py:5-18`) driven end to end against test doubles of the
third-party layers that are absent here (LAMMPS' `cslib`, `ase.calculators.socketio`):

  * a fake `CSlib` plays the LAMMPS `fix client/md` client: handshake -> SETUP -> 3 x STEP -> exit,
    and records every reply; the payloads must equal `model_calc` on the same coordinates;
  * a stub `SocketClient` plays i-PI: it asks the attached calculator for energy and forces.

CPU flavour: kernels replaced by tests/ref_ops.py (host logic + protocol); `-m gpu` flavour: the
real HIP path through the same loop.
"""
import sys
import types

import numpy as np
import pytest
import torch

import ref_ops
from helpers import Golden, rel_err
from hermnet_amd.plugin import lmp_interface as L
from hermnet_amd.plugin import ase_interface as A


class FakeCSlib(object):
    """The subset of LAMMPS' `cslib.CSlib` that `fix client/md` <-> server traffic uses, scripted:
    `script` = list of (msg_id, {field_id: value}); replies are recorded in `self.replies`."""

    instances = []

    def __init__(self, csflag, mode, ptr, comm):
        assert csflag == 1 and isinstance(mode, bytes) and isinstance(ptr, bytes) and comm is None
        self.mode, self.ptr = mode, ptr
        self.script = list(FakeCSlib.script)
        self.cur = None
        self.replies = []
        self.open_reply = None
        FakeCSlib.instances.append(self)

    # --- client -> server
    def recv(self):
        msg_id, fields = self.script.pop(0)
        self.cur = fields
        ids = list(fields.keys())
        return msg_id, len(ids), ids, [0] * len(ids), [0] * len(ids)

    def unpack_string(self, fid):
        return self.cur[fid]

    def unpack_int(self, fid):
        return int(self.cur[fid])

    def unpack(self, fid, tflag):
        assert tflag == 1
        return list(self.cur[fid])

    # --- server -> client
    def send(self, msg_id, nfield):
        self.open_reply = {"msg_id": msg_id, "nfield": nfield, "fields": {}}
        self.replies.append(self.open_reply)

    def pack(self, fid, ftype, flen, data):
        assert ftype == 4 and len(data) == flen and all(isinstance(v, float) for v in data)
        self.open_reply["fields"][fid] = np.asarray(data, dtype=np.float64)

    def pack_double(self, fid, value):
        assert isinstance(value, float)
        self.open_reply["fields"][fid] = value


def _cpu_ops(monkeypatch):
    import hermnet_amd.hermnet as hmod
    import hermnet_amd.layer as lmod
    monkeypatch.setattr(hmod.HVNet, "_require_device", staticmethod(lambda pos: None))
    monkeypatch.setattr(hmod, "EdgeGeometry", ref_ops.RefEdgeGeometry)
    for fn in ["energy_head_fwd", "energy_head_bwd", "layernorm_fwd", "layernorm_bwd", "ssilu_fwd", "ssilu_bwd",
               "update_mid", "update_out", "update_out_bwd", "update_mid_bwd", "node_pre_fwd", "node_pre_bwd", "node_update_fwd",
               "node_update_bwd", "node_update_pre_fwd", "node_pre_fwd16", "node_pre_bwd16"]:
        monkeypatch.setattr(lmod.nodeops, fn, getattr(ref_ops, fn))
    monkeypatch.setattr(lmod, "_msg_fwd", ref_ops.msg_fwd)
    monkeypatch.setattr(lmod, "_msg_bwd", ref_ops.msg_bwd)


def _lammps_session(g, device, ensemble, monkeypatch):
    """Run `serve` against the scripted client; returns (client, expected replies from model_calc)."""
    elems = g.elems
    a = g.arrays
    cell = a["cell"].reshape(3, 3).astype(np.float64)
    z = a["atomic_number"]
    types = [elems.index({13: "Al", 28: "Ni", 29: "Cu", 14: "Si"}[int(v)]) + 1 for v in z]
    n = len(z)
    rs = np.random.RandomState(5)
    frames = [a["pos"].astype(np.float64)]
    for _ in range(3):
        frames.append(frames[-1] + rs.normal(scale=0.02, size=frames[-1].shape))
    box2 = cell * 1.01                                     # NPT-style box change in the last STEP
    script = [(0, {1: b"md"}),
              (L.SETUP, {L.DIM: 3, L.PERIODICITY: [1, 1, 1], L.ORIGIN: [0.0, 0.0, 0.0], L.BOX: cell.reshape(-1).tolist(),
                         L.NATOMS: n, L.NTYPES: len(elems), L.TYPES: types, L.COORDS: frames[0].reshape(-1).tolist()}),
              (L.STEP, {L.COORDS: frames[1].reshape(-1).tolist()}),
              (L.STEP, {L.COORDS: frames[2].reshape(-1).tolist()}),
              (L.STEP, {L.COORDS: frames[3].reshape(-1).tolist(), L.ORIGIN: [0.0, 0.0, 0.0],
                        L.BOX: box2.reshape(-1).tolist()}),
              (-1, {})]
    FakeCSlib.script, FakeCSlib.instances = script, []
    fake = types_module("cslib", CSlib=FakeCSlib)
    model = g.model().to(device)
    rc = g.model_kw.get("rc", 5.0)
    argv = ["-m", "file", "-p", "tmp.couple", "-d", device, "-f", "unused.pt", "-s", "-1.25", "-r", str(rc),
            "-c", "True", "-t"] + elems + ["-e", ensemble]
    served = L.serve(argv, model=model, cslib=fake)
    assert served == 4
    cs = FakeCSlib.instances[0]
    assert (cs.mode, cs.ptr) == (b"file", b"tmp.couple")
    expect = []
    dev = device if device != "cpu" else None
    for k, pos in enumerate(frames):
        c = box2 if k == 3 else cell
        data = A.build_graph(c, np.asarray(z), pos, rc, device=dev)
        expect.append(A.model_calc(model, data, device, True, ensemble, -1.25, "metal"))
    return cs, expect, n


def types_module(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


def _check_lammps_replies(cs, expect, n):
    # handshake ack, one reply per SETUP/STEP, final ack
    assert [r["msg_id"] for r in cs.replies] == [0, L.SETUP, L.STEP, L.STEP, L.STEP, 0]
    assert cs.replies[0]["nfield"] == 0 and cs.replies[-1]["nfield"] == 0
    for rep, (e, f, v) in zip(cs.replies[1:-1], expect):
        assert rep["nfield"] == 3 and set(rep["fields"]) == {L.FORCES, L.ENERGY, L.VIRIAL}
        assert rep["fields"][L.FORCES].shape == (3 * n,)
        assert rel_err(torch.from_numpy(rep["fields"][L.FORCES]), torch.from_numpy(f.reshape(-1).astype(np.float64))) < 1e-6
        assert abs(rep["fields"][L.ENERGY] - e) <= 1e-6 * abs(e)
        assert np.allclose(rep["fields"][L.VIRIAL], v, rtol=1e-5, atol=1e-5 * max(1.0, np.abs(v).max()))
    # the frames differ, so must the replies (a server stuck on the SETUP coordinates would pass the above)
    assert not np.allclose(cs.replies[1]["fields"][L.FORCES], cs.replies[2]["fields"][L.FORCES])


@pytest.mark.parametrize("ensemble", ["NVT", "NPT"])
def test_lammps_server_loop_against_fake_cslib_cpu(ensemble, monkeypatch):
    _cpu_ops(monkeypatch)
    g = Golden("alloy108")
    cs, expect, n = _lammps_session(g, "cpu", ensemble, monkeypatch)
    _check_lammps_replies(cs, expect, n)
    if ensemble == "NPT":
        assert np.abs(cs.replies[1]["fields"][L.VIRIAL]).max() > 0
    else:
        assert np.abs(cs.replies[1]["fields"][L.VIRIAL]).max() == 0


@pytest.mark.gpu
@pytest.mark.parametrize("ensemble", ["NVT", "NPT"])
def test_lammps_server_loop_against_fake_cslib_gpu(ensemble, monkeypatch):
    g = Golden("alloy108")
    cs, expect, n = _lammps_session(g, "cuda", ensemble, monkeypatch)
    _check_lammps_replies(cs, expect, n)
    # first frame = the golden structure: the wire payload equals the reference's forces
    f0 = torch.from_numpy(cs.replies[1]["fields"][L.FORCES]).float().reshape(-1, 3)
    assert rel_err(f0, g.forces) < 1e-5


def test_lammps_server_rejects_bad_handshake_and_unknown_message():
    model = object.__new__(torch.nn.Module)
    torch.nn.Module.__init__(model)
    argv = ["-f", "x", "-s", "0", "-r", "5", "-c", "True", "-t", "Si"]
    for script, msg in [([(3, {})], "handshake"), ([(0, {1: b"xx"})], "protocol"),
                        ([(0, {1: b"md"}), (7, {})], "unrecognized")]:
        FakeCSlib.script, FakeCSlib.instances = script, []
        with pytest.raises(SystemExit) as ei:
            L.serve(argv, model=model, cslib=types_module("cslib", CSlib=FakeCSlib))
        assert msg in str(ei.value)


class _FakeAtoms(object):
    def __init__(self, g):
        a = g.arrays
        sym = {13: "Al", 28: "Ni", 29: "Cu", 14: "Si", 1: "H", 6: "C", 8: "O"}
        self.positions = a["pos"].astype(np.float64)
        self.cell = a["cell"].reshape(3, 3).astype(np.float64) if "cell" in a else np.zeros((3, 3))
        self.pbc = np.array([True] * 3 if "cell" in a else [False] * 3)
        self._sym = [sym[int(v)] for v in a["atomic_number"]]
        self.calc = None

    def get_chemical_symbols(self):
        return self._sym


def _ipi_session(g, device, monkeypatch):
    """`ipi_communicate` with ASE's two entry points replaced: `read_vasp` returns the fixture structure,
    `SocketClient.run(atoms)` does what ASE's client does per i-PI request (POSDATA -> calculate -> FORCEREADY)."""
    log = {}

    class SocketClient(object):
        def __init__(self, host=None, port=None, unixsocket=None):
            log["ctor"] = dict(host=host, port=port, unixsocket=unixsocket)

        def run(self, atoms):
            calc = atoms.calc
            for k in range(2):
                atoms.positions = atoms.positions + 0.01 * k
                calc.calculate(atoms, ["energy", "forces", "stress"])
                log.setdefault("results", []).append({k_: np.copy(v) if isinstance(v, np.ndarray) else v
                                                      for k_, v in calc.results.items()})

    monkeypatch.setitem(sys.modules, "ase", types_module("ase"))
    monkeypatch.setitem(sys.modules, "ase.calculators", types_module("ase.calculators"))
    monkeypatch.setitem(sys.modules, "ase.calculators.socketio", types_module("ase.calculators.socketio",
                                                                                SocketClient=SocketClient))
    monkeypatch.setitem(sys.modules, "ase.io", types_module("ase.io"))
    monkeypatch.setitem(sys.modules, "ase.io.vasp", types_module("ase.io.vasp", read_vasp=lambda path: _FakeAtoms(g)))
    from hermnet_amd.plugin.ipi_interface import ipi_communicate
    calc = A.NNCalculator(g.model(), None, trn_mean=0.5, device_=device)
    ipi_communicate("POSCAR", calc, host="hermnet", port=12345, mode="unix")
    assert log["ctor"] == dict(host=None, port=None, unixsocket="hermnet")
    ipi_communicate("POSCAR", calc, host="127.0.0.1", port=12345, mode="inet")
    assert log["ctor"] == dict(host="127.0.0.1", port=12345, unixsocket=None)
    with pytest.raises(AssertionError):
        ipi_communicate("POSCAR", calc, mode="tcp")
    return log["results"]


def test_ipi_client_against_stub_socketclient_cpu(monkeypatch):
    _cpu_ops(monkeypatch)
    g = Golden("alloy108")
    res = _ipi_session(g, "cpu", monkeypatch)
    assert abs(res[0]["energy"] - (float(g.energy[0]) + 0.5)) < 5e-6 * abs(float(g.energy[0]))
    assert rel_err(torch.from_numpy(res[0]["forces"]), g.forces) < 1e-5
    assert res[0]["free_energy"] == res[0]["energy"] and res[0]["stress"].shape == (6,)


@pytest.mark.gpu
def test_ipi_client_against_stub_socketclient_gpu(monkeypatch):
    g = Golden("alloy108")
    res = _ipi_session(g, "cuda", monkeypatch)
    assert abs(res[0]["energy"] - (float(g.energy[0]) + 0.5)) < 5e-6 * abs(float(g.energy[0]))
    assert rel_err(torch.from_numpy(res[0]["forces"]), g.forces) < 1e-5


def test_ase_calculator_never_caches_a_stress_it_did_not_compute(monkeypatch):
    """ASE's `calculation_required` only asks whether a property's NAME is in `results`.  A periodic call
    that was not asked for the stress (NVT, properties=('energy',)) used to store zeros there, so a following
    `atoms.get_stress()` on the same atoms got the cached zeros without a recomputation.  Now the key is absent until the
    stress has been computed; an open system keeps its zeros (it has none)."""
    _cpu_ops(monkeypatch)
    g = Golden("alloy32_h256")              # (32 atoms: the dense CPU restatement of the message kernel costs seconds per call)
    calc = A.NNCalculator(g.model(), None, trn_mean=0.0, device_="cpu")
    atoms = _FakeAtoms(g)
    calc.calculate(atoms, ["energy"])
    assert "stress" not in calc.results and "energy" in calc.results and "forces" in calc.results
    calc.calculate(atoms, ["stress"])                      # (what ASE does next, the name being absent)
    st = np.asarray(calc.results["stress"])
    assert st.shape == (6,) and np.abs(st).max() > 0
    calc.calculate(atoms, ["energy", "forces"])            # and a later call without it drops the stale value again
    assert "stress" not in calc.results
    open_atoms = _FakeAtoms(g)               # the same atoms as an open system: no stress, zeros stay (as the reference stores)
    open_atoms.pbc = np.array([False] * 3)
    calc.calculate(open_atoms, ["energy"])
    assert np.abs(calc.results["stress"]).max() == 0
