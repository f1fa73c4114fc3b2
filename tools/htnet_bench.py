#!/usr/bin/env python3
"""BASELINE configs[2]: HTNet (18 triadic relations) on the 10,000-atom alloy cell, energy + forces per step.
    python tools/htnet_bench.py [steps]     (rocprofv3 --kernel-trace --stats -- python3 tools/htnet_bench.py for the profile)"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hermnet_amd as hn  # noqa: E402
from hermnet_amd import synth  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
if os.environ.get("HN_OPTIONS") or os.environ.get("HN_SWITCHES"):      # (A/B loops: tools/_opts.py)
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from _opts import apply_option_env
    apply_option_env()
dev = torch.device("cuda")
model = hn.HTNet(["Al", "Ni", "Cu"], rc=5.0, num_layers=5, hidden_channels=128, num_rbf=128).eval()
model.load_state_dict(synth.synth_state_dict(model.state_dict(), 10))
model = model.to(dev)
for p in model.parameters():
    p.requires_grad_(False)
d = synth.fcc_alloy(reps=(10, 10, 25), seed=0, device=dev)


def one():
    d.pos.requires_grad_(True)
    e = model(d)
    return e, -torch.autograd.grad(e.sum(), d.pos)[0]


for _ in range(5):
    one()
torch.cuda.synchronize()
ts = []
for _ in range(steps):
    t0 = time.perf_counter()
    one()
    torch.cuda.synchronize()
    ts.append(time.perf_counter() - t0)
ts.sort()
print(json.dumps({"workload": "configs[2] HTNet 10k atoms", "ms_per_step": ts[len(ts) // 2] * 1e3, "ms_per_step_min": ts[0] * 1e3,
                  "atom_steps_per_s": d.pos.size(0) / ts[len(ts) // 2]}))
