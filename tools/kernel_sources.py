#!/usr/bin/env python3
"""Every device kernel / memcpy of one energy+forces step in launch order with the host op that issued it:
    python tools/kernel_sources.py [htnet | shard]      (shard: the atom-sharded step on ONE rank over RCCL)"""
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
stepper = None
if len(sys.argv) > 1 and sys.argv[1] == "shard":
    import numpy as np
    import torch.distributed as dist
    from hermnet_amd.sharding import SlabStepper
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29543")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    pos, cell, z = synth.fcc_alloy_atoms(reps=(10, 10, 25), seed=0)
    stepper = SlabStepper(torch.from_numpy(z).to(dev), torch.from_numpy(cell.astype(np.float32)).to(dev), 5.0, 0, 1, skin=1.0,
                          group=dist.group.WORLD, deferred=True)
    gpos = torch.from_numpy(pos.astype(np.float32)).to(dev)
    data, _plan = stepper(gpos)
else:
    data = synth.fcc_alloy(reps=(10, 10, 25), seed=0, device=dev)


def step():
    global data
    if stepper is not None:          # as bench.py's sharded step: planning under the skin + neighbour list, every step
        data, _p = stepper(gpos)
    data.pos.requires_grad_(True)
    e = model(data)
    return e, -torch.autograd.grad(e.sum(), data.pos)[0]


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
rows = []
for ev in prof.events():
    if not ev.kernels:
        continue
    # leaf ops only: an op whose child also owns these kernels is skipped
    if any(ch.kernels for ch in (ev.cpu_children or [])):
        continue
    chain, p = [], ev.cpu_parent
    while p is not None and len(chain) < 3:
        chain.append(p.name[:40])
        p = p.cpu_parent
    stack = [fr for fr in (ev.stack or []) if "hermnet_amd" in fr or "bench" in fr or "tools/" in fr]
    for k in ev.kernels:
        rows.append((ev.time_range.start, k.name[:70], k.duration, ev.name[:30], " < ".join(chain), stack[0][-60:] if stack else ""))
for t, kn, dur, op, chain, st in sorted(rows):
    print("%-70s %7.1f us  %-30s %-60s %s" % (kn, dur, op, chain, st))
