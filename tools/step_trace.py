#!/usr/bin/env python3
"""The device kernels of ONE energy + forces step of configs[1], in launch order, with their durations (torch.profiler):
    python tools/step_trace.py [reps_z]
What the step is made of besides this library's own kernels: torch glue (copies, fills, elementwise ops, reductions)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hermnet_amd as hn  # noqa: E402
from hermnet_amd import synth  # noqa: E402


def main():
    rz = int(sys.argv[1]) if len(sys.argv) > 1 else 25
    dev = torch.device("cuda")
    kw = dict(rc=5.0, num_layers=5, hidden_channels=128, num_rbf=128)
    model = hn.HVNet(["Al", "Ni", "Cu"], **kw).eval()
    model.load_state_dict(synth.synth_state_dict(model.state_dict(), 10))
    model = model.to(dev)
    for p in model.parameters():
        p.requires_grad_(False)
    data = synth.fcc_alloy(reps=(10, 10, rz), seed=0, device=dev)

    def step():
        data.pos.requires_grad_(True)
        e = model(data)
        return -torch.autograd.grad(e.sum(), data.pos)[0]

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=False, with_stack=False) as prof:
        step()
        torch.cuda.synchronize()
    evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]
    evs.sort(key=lambda e: e.time_range.start)
    tot = 0.0
    glue = 0.0
    for e in evs:
        own = "anonymous namespace" in e.name or "hermnet" in e.name
        tot += e.device_time
        if not own:
            glue += e.device_time
        print("%8.1f us  %s%s" % (e.device_time, "" if own else "[torch] ", e.name[:110]))
    print("kernels %d, device time %.1f us, of it torch glue %.1f us" % (len(evs), tot, glue))
    # which host-side ops launch the glue
    print(prof.key_averages().table(sort_by="self_cuda_time_total", row_limit=40, max_name_column_width=60))


if __name__ == "__main__":
    main()
