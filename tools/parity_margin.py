#!/usr/bin/env python3
"""How much of the 1e-5 parity tolerance the HIP path uses: relative error of energies and forces (forces relative to max |F|)
against every committed golden of the reference's own code (tests/golden/*.npz):   python tools/parity_margin.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import SMALL_CASES, Golden, rel_err  # noqa: E402

dev = torch.device("cuda")
worst = (0.0, 0.0)
for name in SMALL_CASES:
    g = Golden(name)
    model = g.model().to(dev).eval()
    d = g.data().to(dev)
    d.pos.requires_grad_(True)
    e = model(d)
    f = -torch.autograd.grad(e.sum(), d.pos)[0]
    ee, fe = float(rel_err(e.detach().cpu(), g.energy)), float(rel_err(f.cpu(), g.forces))
    worst = (max(worst[0], ee), max(worst[1], fe))
    print("%-28s energy %.2e   forces %.2e" % (name, ee, fe))
print("worst: energy %.2e, forces %.2e  (tolerance 1e-5)" % worst)
