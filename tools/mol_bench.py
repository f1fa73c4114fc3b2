#!/usr/bin/env python3
"""BASELINE configs[4] in eval(): the 1024-molecule batch, energy + forces per step (rocprofv3 --kernel-trace --stats -- python3
tools/mol_bench.py for the per-kernel list)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hermnet_amd as hn  # noqa: E402
from hermnet_amd import synth  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda")
m = hn.HVNet(["H", "C", "O"], rc=5.0, num_layers=5, hidden_channels=128, num_rbf=128).eval()
m.load_state_dict(synth.synth_state_dict(m.state_dict(), 12))
m = m.to(dev)
for p in m.parameters():
    p.requires_grad_(False)
d = synth.molecule_batch(num_graphs=1024).to(dev)


def one():
    d.pos.requires_grad_(True)
    e = m(d)
    return e, -torch.autograd.grad(e.sum(), d.pos)[0]


for _ in range(5):
    one()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    one()
torch.cuda.synchronize()
print(json.dumps({"workload": "configs[4] 1024-molecule batch, eval", "atoms": int(d.pos.size(0)), "edges": int(d.edge_index.size(1)),
                  "ms_per_step": (time.perf_counter() - t0) / steps * 1e3}))
