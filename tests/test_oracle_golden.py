"""CPU: the oracle (oracle/hermnet_oracle.py) against the golden vectors produced by the
reference's own code (tests/golden/gen_golden.py).  Tolerance 1e-6 relative (fp32, same
operation order)."""
import pytest
import torch

from helpers import Golden, SMALL_CASES, NONGAUSS_CASES, TRAIN_CASES, rel_err
from oracle import hermnet_oracle as orc

TOL = 1e-6


@pytest.mark.parametrize("name", SMALL_CASES + NONGAUSS_CASES)
@pytest.mark.parametrize("mode", ["faithful", "vectorised"])
def test_oracle_matches_reference(name, mode):
    g = Golden(name)
    sd = g.model().state_dict()
    e, f = orc.energy_and_forces(sd, g.elems, g.data(), mode=mode, **g.oracle_kwargs())
    assert rel_err(e, g.energy) < TOL, (e, g.energy)
    assert rel_err(f, g.forces) < 5 * TOL


@pytest.mark.parametrize("name", ["c1_si64", "alloy108"])
def test_oracle_intermediates(name):
    g = Golden(name)
    sd = g.model().state_dict()
    d = g.data()
    kw = g.oracle_kwargs()
    e, inter = orc.hvnet_energy(sd, g.elems, d.pos, d.atomic_number, d.edge_index, d.batch,
                                d.get("edge_shift"), d.get("cell"), return_intermediates=True, **kw)
    a = g.arrays
    assert rel_err(inter["edge_dist"], torch.from_numpy(a["edge_dist"])) < 1e-7
    assert rel_err(inter["edge_vec"], torch.from_numpy(a["edge_vec"])) < 1e-6
    for l in range(kw["num_layers"]):
        assert rel_err(inter["x"][l], torch.from_numpy(a["x_l%d" % l])) < TOL
        assert rel_err(inter["vec"][l], torch.from_numpy(a["vec_l%d" % l])) < TOL


def test_oracle_forces_match_fp64_finite_differences():
    g = Golden("c1_si64")
    sd64 = {k: v.double() for k, v in g.model().state_dict().items()}
    d = g.data()
    kw = g.oracle_kwargs()
    pos = d.pos.double()

    def energy(p):
        return orc.hvnet_energy(sd64, g.elems, p, d.atomic_number, d.edge_index, d.batch,
                                d.edge_shift.double(), d.cell.double(), **kw).sum()

    p = pos.clone().requires_grad_(True)
    f = -torch.autograd.grad(energy(p), p)[0]
    h = 1e-5
    for (i, c) in [(0, 0), (17, 1), (63, 2)]:
        pp, pm = pos.clone(), pos.clone()
        pp[i, c] += h
        pm[i, c] -= h
        fd = -(energy(pp) - energy(pm)) / (2 * h)
        assert abs(float(fd - f[i, c])) < 1e-6 * max(1.0, abs(float(fd)))
    # and the fp32 golden forces agree with fp64 to fp32 accuracy
    assert rel_err(g.forces.double(), f) < 2e-5


def test_physics_invariants_of_the_oracle():
    g = Golden("alloy108")
    sd = g.model().state_dict()
    d = g.data()
    kw = g.oracle_kwargs()
    e, f = orc.energy_and_forces(sd, g.elems, d, **kw)
    assert float(f.sum(0).abs().max()) < 1e-4          # momentum conservation
    d2 = g.data()
    d2.pos = d2.pos + torch.tensor([0.3, -0.2, 0.1])
    e2, f2 = orc.energy_and_forces(sd, g.elems, d2, **kw)
    assert rel_err(e2, e) < 1e-5 and rel_err(f2, f) < 1e-4   # translation invariance


@pytest.mark.parametrize("name", TRAIN_CASES)
@pytest.mark.parametrize("mode", ["faithful", "vectorised"])
def test_oracle_training_step_matches_reference(name, mode):
    """Loss of `example/dist_train.py:86-99` (force term through create_graph=True) and the gradient of
    every parameter, against the reference's own backward pass."""
    g = Golden(name)
    sd = g.model().state_dict()
    y, ftgt, gamma, (loss, e_loss, f_loss), grads = g.training()
    l, le, lf, og = orc.training_loss_and_grads(sd, g.elems, g.data(), y, ftgt, gamma, mode=mode, **g.oracle_kwargs())
    assert abs(float(le) - e_loss) < 2e-5 * max(1.0, e_loss)
    assert abs(float(lf) - f_loss) < 2e-5 * max(1.0, f_loss)
    assert abs(float(l) - loss) < 2e-5 * max(1.0, loss)
    assert set(k for k, v in og.items() if v is not None) == set(grads.keys())
    gmax = max(float(v.abs().max()) for v in grads.values())
    for k, ref in grads.items():
        # per tensor, relative to the tensor's largest entry (floor: 1e-3 of the largest gradient anywhere)
        scale = max(float(ref.abs().max()), 1e-3 * gmax)
        assert float((og[k] - ref).abs().max()) / scale < 2e-5, k
