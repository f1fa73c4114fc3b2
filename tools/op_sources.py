#!/usr/bin/env python3
"""Which host lines launch the small torch kernels of one energy+forces step (aten::copy_, aten::to, aten::sum, sort, cat ...):
    python tools/op_sources.py [htnet]     (configs[1] / configs[2], one profiled step after warm-up)"""
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hermnet_amd as hn  # noqa: E402
from hermnet_amd import synth  # noqa: E402

dev = torch.device("cuda")
Model = hn.HTNet if (len(sys.argv) > 1 and sys.argv[1] == "htnet") else hn.HVNet
model = Model(["Al", "Ni", "Cu"], rc=5.0, num_layers=5, hidden_channels=128, num_rbf=128).eval()
model.load_state_dict(synth.synth_state_dict(model.state_dict(), 10))
model = model.to(dev)
for p in model.parameters():
    p.requires_grad_(False)
data = synth.fcc_alloy(reps=(10, 10, 25), seed=0, device=dev)


def step():
    data.pos.requires_grad_(True)
    e = model(data)
    return e, -torch.autograd.grad(e.sum(), data.pos)[0]


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
import collections
agg = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_time_total > 0 and ev.name.startswith("aten::") and (ev.cpu_parent is None or not ev.cpu_parent.name.startswith("aten::")):
        parent = ev.cpu_parent.name if ev.cpu_parent is not None else "-"
        k = (ev.name, parent, tuple(str(s_) for s_ in (ev.input_shapes or []))[:3])
        agg[k][0] += 1
        agg[k][1] += ev.device_time_total
for (name, parent, shapes), (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-24s x%-3d %8.1f us   in %-40s %s" % (name, n, t, parent[:40], shapes))
