"""`virial_calc` with the reference's signature and unit factors (`HermNet/utils.py:138-160`)."""
import torch

_NKTV2P = {"metal": 1.6021765e6, "lj": 1.0, "si": 1.0, "cgs": 1.0, "micro": 1.0, "nano": 1.0,
           "real": 68568.415, "electron": 2.94210108e13}


def virial_calc(cell, pos, forces, energy, units='metal', pbc=False):
    """Symmetrised virial  sum_i pos_i (x) F_i  -  cell^T dE/dcell  (periodic) in LAMMPS pressure*volume units.

    `cell` must require grad and be the tensor the model saw (its gradient comes from the edge
    geometry kernel's backward, hermnet_amd/ops.py)."""
    if units not in _NKTV2P:
        raise ValueError('Illegal units command')
    nktv2p = _NKTV2P[units]
    if pbc:
        assert cell.requires_grad
        gcell = torch.autograd.grad(energy.sum(), cell)[0].reshape(3, 3)
        virial = torch.einsum('ij, ik->jk', pos, forces) - cell.reshape(3, 3).T @ gcell
        virial = (virial + virial.T) / 2 * nktv2p
    else:
        virial = torch.einsum('ij, ik->jk', pos, forces) * nktv2p
        virial = (virial + virial.T) / 2
    return virial
