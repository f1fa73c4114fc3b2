#!/usr/bin/env python3
"""The captured MD step (graph.GraphedMDStep) of a SMALL alloy cell, replayed N times: where a launch-bound step spends its
time.   rocprofv3 --kernel-trace --stats -d out -- python3 tools/small_cell_probe.py [reps_x reps_y reps_z] [replays]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hermnet_amd as hn  # noqa: E402
from hermnet_amd import synth  # noqa: E402
from hermnet_amd.graph import GraphedMDStep  # noqa: E402

reps = tuple(int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2, 2, 4)
n = int(sys.argv[4]) if len(sys.argv) > 4 else 200
dev = torch.device("cuda:0")
model = hn.HVNet(["Al", "Ni", "Cu"], rc=5.0, num_layers=5, hidden_channels=128, num_rbf=128).eval()
model.load_state_dict(synth.synth_state_dict(model.state_dict(), 10))
model = model.to(dev)
for p in model.parameters():
    p.requires_grad_(False)
pos, cell, z = synth.fcc_alloy_atoms(reps=reps, seed=0)
step = GraphedMDStep(model, torch.from_numpy(z).to(dev), torch.from_numpy(cell.astype(np.float32)).to(dev),
                     torch.from_numpy(pos.astype(np.float32)).to(dev))
for _ in range(10):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    step()
torch.cuda.synchronize()
print("atoms %d: %.3f ms per replayed MD step" % (len(z), (time.perf_counter() - t0) / n * 1e3))
