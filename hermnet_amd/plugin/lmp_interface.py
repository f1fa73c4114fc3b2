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
    """The flags of `lmp_calc.py:89-127` with the reference's option names, defaults and required set
    (`-f -s -r -c -t` are required there too).  Two deliberate differences: `--device` defaults to `cuda`
    (the reference: `cpu`; this engine has no CPU path and refuses it loudly), and `--reference-compat`
    (not in the reference) makes the neighbour search reproduce the reference pipeline's edge conventions
    (`data.py:16,19-24`: edge_shift = +S, 32-neighbour cap for open systems) for checkpoints that were trained
    through that pipeline.  `--mean` / `--rc` are accepted as aliases of `--stats` / `--radius`."""
    p = argparse.ArgumentParser(description="HermNet works as a server for LAMMPS.")
    p.add_argument('-m', '--mode', help='The mode for exchange messages', type=str, choices=['file', 'zmq'],
                   default='zmq')
    p.add_argument('-p', '--ptr', help='Filename or socket ID', type=str, default='tmp.couple')
    p.add_argument('-d', '--device', help='Device to allocate HermNet', type=str, choices=['cpu', 'cuda'],
                   default='cuda')
    p.add_argument('-f', '--model', help='The path that saves trained model', type=str, required=True)
    p.add_argument('-s', '--stats', '--mean', dest='stats', type=float, required=True,
                   help='The mean value of trainset that shifts the output of model')
    p.add_argument('-r', '--radius', '--rc', dest='radius', help='Cutoff radius', type=float, required=True)
    p.add_argument('-c', '--periodic', help='If the system is PBC or not', type=str, choices=['True', 'False'],
                   required=True)
    p.add_argument('-u', '--units', help='Units', type=str, default='metal')
    p.add_argument('-t', '--elems', help='Elements. The order should be the same with data file', type=str,
                   nargs='*', required=True)
    p.add_argument('-e', '--ensemble', help='Ensemble', type=str, default='NVT')
    p.add_argument('--reference-compat', action='store_true',
                   help="edge conventions of the reference's own data pipeline (see the docstring)")
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


def serve(argv=None, model=None, cslib=None):
    """The server loop of `lmp_calc.py:135-238`: handshake ("md" protocol), then SETUP / STEP messages until
    LAMMPS sends a negative message id; every message is answered with FORCES (3N doubles), ENERGY, VIRIAL (6).

    `cslib` is LAMMPS' python wrapper module (imported here if not given: it is not vendored anywhere);
    `model` overrides the checkpoint load (tests, ensembles).  Returns the number of evaluations served."""
    if cslib is None:
        import cslib            # noqa: F811  (LAMMPS: lib/message/cslib/src/cslib.py)
    from ..hermnet import HVNet
    a = parse_args(argv)
    device = torch.device(a.device)
    if model is None:
        model = HVNet(elems=a.elems, rc=a.radius, intensive=False).to(device)
        model.load_state_dict(torch.load(a.model, map_location=device))
    for p_ in model.parameters():
        p_.requires_grad_(False)
    cs = cslib.CSlib(1, a.mode.encode('ascii'), a.ptr.encode('ascii'), None)
    msg_id, nfield, fieldid, fieldtype, fieldlen = cs.recv()
    if msg_id != 0:
        raise SystemExit('Error: Bad initial client/server handshake')
    if cs.unpack_string(1) not in (b'md', 'md'):
        raise SystemExit('Error: Mismatch in client/server protocol')
    cs.send(0, 0)
    types = coords = box = natoms = None
    pbc = a.periodic == 'True'             # (the reference: eval(args.periodic), lmp_calc.py:227)
    served = 0
    while True:
        msg_id, nfield, fieldid, fieldtype, fieldlen = cs.recv()
        if msg_id < 0:
            break
        if msg_id == SETUP:                # beginning of each run: box, types, coordinates, counts
            for f in fieldid:
                if f == BOX:
                    box = cs.unpack(BOX, 1)
                elif f == NATOMS:
                    natoms = cs.unpack_int(NATOMS)
                elif f == TYPES:
                    types = cs.unpack(TYPES, 1)
                elif f == COORDS:
                    coords = cs.unpack(COORDS, 1)
                elif f in (DIM, NTYPES):
                    cs.unpack_int(f)
                elif f in (PERIODICITY, ORIGIN):
                    cs.unpack(f, 1)
        elif msg_id == STEP:               # every timestep: coordinates, optionally a new box
            for f in fieldid:
                if f == COORDS:
                    coords = cs.unpack(COORDS, 1)
                elif f == BOX:
                    box = cs.unpack(BOX, 1)
                elif f == ORIGIN:
                    cs.unpack(ORIGIN, 1)
        else:
            raise SystemExit('Error: HermNet wrapper received unrecognized message')
        z = lammps_types_to_numbers(types, a.elems)
        pos = np.asarray(coords, dtype=np.float64).reshape(-1, 3)
        if natoms is not None and pos.shape[0] != natoms:
            raise SystemExit('Error: COORDS does not hold NATOMS atoms')
        # GPU runs: upload the coordinates first, neighbour search on the device (the reference rebuilds the list
        # on the host every step, lmp_calc.py:224)
        dev = a.device if device.type == 'cuda' else None
        data = build_graph(box_to_cell(box) if pbc else None, z, pos, a.radius, device=dev,
                           reference_compat=a.reference_compat)
        e, f, v = calculator(data, model, a.stats, a.device, pbc, a.units, a.ensemble)
        pack_reply(cs, msg_id, f, e, v)
        served += 1
    cs.send(0, 0)
    return served


if __name__ == '__main__':  # pragma: no cover
    serve()
