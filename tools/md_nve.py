#!/usr/bin/env python3
"""Velocity-Verlet NVE run on the device: neighbour list (rebuilt every step, as the reference's calculators do,
`plugin/ase_interface/calculator.py:49`) + HVNet energy/forces, everything resident on the GPU.  An end-to-end check
that the forces are the gradient of the energy the model reports: the total energy must stay put while kinetic and
potential energy trade places.

    python tools/md_nve.py [--reps 6 6 6] [--steps 200] [--dt 0.5] [--temp 300]

Units: eV, Angstrom, amu, fs (the synthetic model is random-initialised: the numbers mean nothing physically)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import hermnet_amd as hn  # noqa: E402
from hermnet_amd import synth  # noqa: E402
from hermnet_amd.data import Data  # noqa: E402
from hermnet_amd.neighbor import neighbor_search  # noqa: E402

AMU_A2_FS2_TO_EV = 103.642696562         # 1 amu A^2 / fs^2 in eV
KB = 8.617333262e-5                       # eV / K
MASS = {13: 26.9815, 28: 58.6934, 29: 63.546}


def energy_forces(model, pos, z, cell, rc, fixed=None):
    ei, sh = fixed if fixed is not None else neighbor_search(pos.detach(), rc, cell)
    d = Data(pos=pos.detach().clone().requires_grad_(True), atomic_number=z, edge_index=ei, edge_shift=sh,
             cell=cell.reshape(1, 3, 3), batch=torch.zeros(z.numel(), dtype=torch.long, device=pos.device))
    e = model(d)
    f = -torch.autograd.grad(e.sum(), d.pos)[0]
    return float(e.detach()[0]), f


def run(reps=(6, 6, 6), steps=200, dt=0.5, temp=300.0, seed=0, rc=5.0, log=None, state_seed=11, fixed_list=False):
    dev = torch.device("cuda:0")
    pos_np, cell_np, z_np = synth.fcc_alloy_atoms(reps=reps, seed=seed)
    pos = torch.from_numpy(pos_np).float().to(dev)
    cell = torch.from_numpy(cell_np).float().to(dev)
    z = torch.from_numpy(z_np).to(dev)
    model = hn.HVNet(["Al", "Ni", "Cu"], rc=rc, num_layers=3, hidden_channels=128, num_rbf=128).eval()
    model.load_state_dict(synth.synth_state_dict(model.state_dict(), state_seed))
    model = model.to(dev)
    for p in model.parameters():
        p.requires_grad_(False)
    m = torch.tensor([MASS[int(v)] for v in z_np], device=dev)[:, None] * AMU_A2_FS2_TO_EV    # eV fs^2 / A^2
    gen = torch.Generator(device="cpu").manual_seed(seed + 1)
    v = (torch.randn(pos.shape, generator=gen).to(dev) * torch.sqrt(KB * temp / m))
    v -= (v * m).sum(0) / m.sum()                      # no centre-of-mass drift
    box = torch.diagonal(cell)
    # fixed_list: the step-0 neighbour list for the whole run (coordinates then stay unwrapped, its image shifts stay
    # valid).  The model's energy is NOT continuous when an edge enters or leaves the list -- rbf_proj's bias is outside
    # the envelope (rmnet.py:45,55), the reference has the same jump -- so only a fixed list conserves energy
    fixed = neighbor_search(pos, rc, cell) if fixed_list else None
    e_pot, f = energy_forces(model, pos, z, cell, rc, fixed)
    hist = []
    for step in range(steps + 1):
        e_kin = float(0.5 * (m * v * v).sum())
        hist.append((e_pot, e_kin))
        if log and step % log == 0:
            print("step %4d  E_pot %+.6f  E_kin %.6f  E_tot %+.6f" % (step, e_pot, e_kin, e_pot + e_kin))
        if step == steps:
            break
        v = v + 0.5 * dt * f / m
        pos = pos + dt * v
        if not fixed_list:
            pos = pos - torch.floor(pos / box) * box   # wrap (orthorhombic cell)
        e_pot, f = energy_forces(model, pos, z, cell, rc, fixed)
        v = v + 0.5 * dt * f / m
    return hist


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, nargs=3, default=[6, 6, 6])
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--dt", type=float, default=0.5)
    ap.add_argument("--temp", type=float, default=300.0)
    ap.add_argument("--fixed-list", action="store_true")
    a = ap.parse_args()
    h = run(tuple(a.reps), a.steps, a.dt, a.temp, log=max(a.steps // 10, 1), fixed_list=a.fixed_list)
    et = [p + k for p, k in h]
    ek = [k for _, k in h]
    print("atoms %d  steps %d  dt %.2f fs:  E_tot drift %.3e eV (max |E_tot - E_tot0|),  E_kin range %.3e eV"
          % (4 * a.reps[0] * a.reps[1] * a.reps[2], a.steps, a.dt, max(abs(x - et[0]) for x in et), max(ek) - min(ek)))
