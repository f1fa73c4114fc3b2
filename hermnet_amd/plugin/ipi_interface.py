"""i-PI client (`plugin/i-pi_interface/ipi_calc.py:5-18`): wraps an `NNCalculator` in ASE's
SocketClient.  Needs `ase` (imported lazily); the transport itself is all ASE."""


def ipi_communicate(poscar, calc, host='localhost', port=8888, mode='unix'):
    from ase.calculators.socketio import SocketClient
    from ase.io.vasp import read_vasp
    atoms = read_vasp(poscar)
    atoms.calc = calc
    assert mode in ['inet', 'unix']
    client = SocketClient(host=host, port=port) if mode == 'inet' else SocketClient(unixsocket=host)
    client.run(atoms)
