"""LAMMPS `fix client/md` server: CLI flags, protocol constants and reply packing of
`plugin/lmp_interface/lmp_calc.py:88-239`.  The transport (`cslib.CSlib`, part of LAMMPS, not
vendored anywhere) is imported only in `serve`; `calculator` is transport-free."""
import argparse

import numpy as np
import torch

from ..elements import atomic_numbers
from .ase_interface import build_graph, model_calc

# lmp_calc.py:136-138
SETUP, STEP = 1, 2
DIM, PERIODICITY, ORIGIN, BOX, NATOMS, NTYPES, TYPES, COORDS, UNITS, CHARGE = range(1, 11)
FORCES, ENERGY, VIRIAL, ERROR = 1, 2, 3, 4


def calculator(data, model, trn_mean, device, pbc, units, ensemble='NVT'):
    """`lmp_calc.py:36-85`: energy, forces flattened to 3N, virial [6]."""
    e, f, v = model_calc(model, data, device, pbc, ensemble, trn_mean, units)
    return e, f.reshape(-1), v


def parse_args(argv=None):
    """The nine flags of `lmp_calc.py:89-127`."""
    p = argparse.ArgumentParser(description="HermNet works as a server for LAMMPS.")
    p.add_argument('-m', '--mode', choices=['file', 'zmq'], required=True)
    p.add_argument('-p', '--ptr', required=True)
    p.add_argument('-d', '--device', default='cuda')
    p.add_argument('-f', '--model', required=True)
    p.add_argument('-s', '--mean', type=float, default=0.0)
    p.add_argument('-r', '--rc', type=float, default=5.0)
    p.add_argument('-c', '--periodic', choices=['True', 'False'], default='True')
    p.add_argument('-u', '--units', default='metal')
    p.add_argument('-t', '--elems', nargs='+', required=True)
    p.add_argument('-e', '--ensemble', choices=['NVT', 'NPT', 'nvt', 'npt'], default='NVT')
    return p.parse_args(argv)


def lammps_types_to_numbers(types, elems):
    """`lmp_calc.py:220-222`: LAMMPS type ids 1..n -> atomic numbers of `elems`."""
    table = np.array([0] + [atomic_numbers[e] for e in elems])
    return table[np.asarray(types, dtype=np.int64)]


def box_to_cell(box):
    """9 doubles from LAMMPS (`lmp_calc.py:196-201`) -> [3,3] rows = lattice vectors."""
    return np.asarray(box, dtype=np.float64).reshape(3, 3)


def pack_reply(cs, msg_id, forces, energy, virial):
    """`lmp_calc.py:232-235`."""
    cs.send(msg_id, 3)
    cs.pack(FORCES, 4, len(forces), [float(x) for x in forces])
    cs.pack_double(ENERGY, float(energy))
    cs.pack(VIRIAL, 4, 6, [float(x) for x in virial])


def serve(argv=None):  # pragma: no cover - needs LAMMPS' cslib
    import torch
    from cslib import CSlib
    from ..hermnet import HVNet
    a = parse_args(argv)
    model = HVNet(a.elems, rc=a.rc, intensive=False).to(a.device)
    model.load_state_dict(torch.load(a.model, map_location=a.device))
    cs = CSlib(1, a.mode, a.ptr, None)
    msg_id, nfield, fieldid, fieldtype, fieldlen = cs.recv()
    if msg_id != 0 or cs.unpack_string(1) != "md":
        raise SystemExit("HermNet server: unexpected protocol")
    cs.send(0, 0)
    types = coords = box = None
    pbc = a.periodic == 'True'
    while True:
        msg_id, nfield, fieldid, fieldtype, fieldlen = cs.recv()
        if msg_id < 0:
            break
        for f in fieldid:
            if f == TYPES:
                types = cs.unpack(TYPES, 1)
            elif f == COORDS:
                coords = cs.unpack(COORDS, 1)
            elif f == BOX:
                box = cs.unpack(BOX, 1)
        z = lammps_types_to_numbers(types, a.elems)
        pos = np.asarray(coords, dtype=np.float64).reshape(-1, 3)
        # GPU runs: upload the coordinates first, neighbour search on the device (the reference rebuilds the list
        # on the host every step, lmp_calc.py:224)
        dev = a.device if torch.device(a.device).type == 'cuda' else None
        data = build_graph(box_to_cell(box) if pbc else None, z, pos, a.rc, device=dev)
        e, f, v = calculator(data, model, a.mean, a.device, pbc, a.units, a.ensemble)
        pack_reply(cs, msg_id, f, e, v)
    cs.send(0, 0)


if __name__ == '__main__':  # pragma: no cover
    serve()
