#!/usr/bin/env python3
"""One GPU, one million atoms: fcc 50 x 50 x 100 (1,000,000 atoms, 43 M directed edges) through the configs[1] model --
size-independent properties (finite, sum of forces ~ 0, energy per atom equal to the 10k cell's within 2 %, bit
reproducible), peak memory and step time.  Not part of the test suite (a fresh box's first import + 1M-atom lattice is
slow); run once per round:   python tools/million_atoms.py [nx ny nz]"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hermnet_amd as hn  # noqa: E402
from hermnet_amd import synth  # noqa: E402

reps = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (50, 50, 100)
dev = torch.device("cuda:0")
kw = dict(rc=5.0, num_layers=5, hidden_channels=128, num_rbf=128)
model = hn.HVNet(["Al", "Ni", "Cu"], **kw).eval()
model.load_state_dict(synth.synth_state_dict(model.state_dict(), 10))
model = model.to(dev)
for p in model.parameters():
    p.requires_grad_(False)
t0 = time.perf_counter()
d = synth.fcc_alloy(reps=reps, seed=0, device=dev)
torch.cuda.synchronize()
t_list = time.perf_counter() - t0
N, E = d.pos.size(0), d.edge_index.size(1)


def step():
    d.pos.requires_grad_(True)
    e = model(d)
    return e.detach(), -torch.autograd.grad(e.sum(), d.pos)[0]


torch.cuda.reset_peak_memory_stats()
e, f = step()
torch.cuda.synchronize()
e2, f2 = step()
ts = []
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
small = synth.fcc_alloy(reps=(10, 10, 25), seed=0, device=dev)
e_small = model(small)
out = {"atoms": N, "edges": E, "lattice_and_device_list_s": t_list, "ms_per_step": min(ts) * 1e3, "atom_steps_per_s": N / min(ts),
       "peak_memory_GB": torch.cuda.max_memory_allocated() / 1e9,
       "finite": bool(torch.isfinite(e).all() and torch.isfinite(f).all()),
       "sum_forces_over_max": float(f.sum(0).abs().max() / f.abs().max()),
       "energy_per_atom": float(e[0]) / N, "energy_per_atom_10k_cell": float(e_small[0]) / small.pos.size(0),
       "bit_reproducible": bool(torch.equal(e, e2) and torch.equal(f, f2))}
print(json.dumps(out))
