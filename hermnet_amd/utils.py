"""`virial_calc` with the reference's signature and unit factors (`HermNet/utils.py:138-160`), and the
GEMM-selection table for the node-level library GEMMs."""
import os

import torch

TUNED_GEMMS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned", "gemm_gfx950.csv")


def prefer_rocblas():
    """Route library GEMMs that have no entry in the tuned table through rocBLAS: on the node-GEMM shapes of this
    path it is as fast on the GPU as hipBLASLt's default heuristic and costs ~9 us less host time per call
    (50 GEMMs per step: 0.45 ms of enqueue time, which matters once the step is host-bound, e.g. sharded runs)."""
    try:
        torch.backends.cuda.preferred_blas_library("cublas")      # "cublas" = rocBLAS on ROCm
    except Exception:
        pass


def enable_tuned_gemms(path=TUNED_GEMMS, online=False):
    """Use the rocBLAS / hipBLASLt solutions recorded in `tuned/gemm_gfx950.csv` (PyTorch TunableOp, tuned on
    MI355X for the node GEMM shapes of BASELINE config 2).  Tables recorded with other library versions are
    rejected by their validators; shapes that are not in the table fall back to rocBLAS's default choice --
    unless `online=True`: then TunableOp also times the candidates of every NEW shape the first time it is seen
    (a few seconds in total; meant for warm-up steps, e.g. the rank-local shapes of an atom-sharded run, where the
    default choice is up to 2x slower on the skinny K = 128/256 GEMMs of this path).  Call `freeze_gemm_tuning()`
    afterwards so that nothing is tuned inside a timed or production region.  Returns True when the table was
    accepted.  Same arithmetic (fp32 MFMA GEMMs) either way: only the tile / solution choice changes."""
    import tempfile
    import torch.cuda.tunable as tunable
    prefer_rocblas()
    tunable.enable(True)
    tunable.tuning_enable(bool(online))
    ok = False
    try:
        # TunableOp writes its table back at exit: keep that out of the working directory
        tunable.set_filename(os.path.join(tempfile.gettempdir(), "hermnet_tunableop_%d.csv" % os.getpid()))
        ok = os.path.exists(path) and bool(tunable.read_file(path))
    except Exception:
        ok = False
    return ok


def freeze_gemm_tuning():
    """Stop timing new GEMM shapes (keep using what is known): end of the warm-up."""
    import torch.cuda.tunable as tunable
    tunable.tuning_enable(False)


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
