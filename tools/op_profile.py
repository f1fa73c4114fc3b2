"""Which aten ops launch the glue kernels of one step (torch.profiler, one step)."""
import sys, torch
sys.path.insert(0, '.')
import hermnet_amd as hn
from hermnet_amd import synth
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
kw = dict(rc=5.0, num_layers=5, hidden_channels=128, num_rbf=128)
model = hn.HVNet(["Al", "Ni", "Cu"], **kw).eval().to(dev)
for p in model.parameters(): p.requires_grad_(False)
data = synth.fcc_alloy(device=dev)
def step():
    data.pos.requires_grad_(True)
    e = model(data)
    return -torch.autograd.grad(e.sum(), data.pos)[0]
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
rows = []
for ev in prof.key_averages():
    if ev.device_time_total > 0 and ev.key.startswith("aten::"):
        rows.append((ev.device_time_total, ev.count, ev.key))
for t, c, k in sorted(rows, reverse=True)[:28]:
    print("%8.1f us  x%3d  %s" % (t, c, k))
print("---- device kernels ----")
krows = {}
for ev in prof.events():
    if getattr(ev, "device_type", None) is not None and str(ev.device_type).endswith("CUDA"):
        k = ev.name[:70]
        t, c = krows.get(k, (0.0, 0))
        krows[k] = (t + ev.device_time, c + 1)
tot = sum(t for t, _ in krows.values())
for k, (t, c) in sorted(krows.items(), key=lambda kv: -kv[1][0])[:40]:
    print("%8.1f us  x%3d  %s" % (t, c, k))
print("total device time %.1f us" % tot)
