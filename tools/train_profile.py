"""Where the training step (train() mode, differentiable device ops) spends its time: torch.profiler, one step."""
import sys, torch
sys.path.insert(0, '.')
import torch.nn.functional as F
import hermnet_amd as hn
from hermnet_amd import synth
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
kw = dict(rc=5.0, num_layers=5, hidden_channels=128, num_rbf=128)
d = synth.molecule_batch(num_graphs=1024).to(dev)
model = hn.HVNet(["H", "C", "O"], **kw)
model.load_state_dict(synth.synth_state_dict(model.state_dict(), 12))
model = model.to(dev).train()
gen = torch.Generator().manual_seed(0)
y = torch.randn(1024, generator=gen).to(dev)
ftgt = (0.5 * torch.randn(d.pos.shape, generator=gen)).to(dev)
def step():
    model.zero_grad()
    d.pos.requires_grad_(True)
    e = model(d)
    f = -torch.autograd.grad(e.sum(), d.pos, create_graph=True)[0]
    loss = 0.2 * F.mse_loss(e, y) + 0.8 * F.mse_loss(f, ftgt)
    loss.backward()
    return loss
import time
for _ in range(2): step()
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
print("training step (forward + create_graph force pass + backward, no optimiser): median %.1f ms, min %.1f ms" % (sorted(ts)[2] * 1e3, min(ts) * 1e3))
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = [(ev.device_time_total, ev.count, ev.key) for ev in prof.key_averages() if ev.device_time_total > 0 and ev.key.startswith("aten::")]
for t, c, k in sorted(rows, reverse=True)[:30]:
    print("%9.1f us  x%4d  %s" % (t, c, k))
import collections
kern = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if str(ev.device_type).endswith("CUDA"):
        kern[ev.name[:90]][0] += 1
        kern[ev.name[:90]][1] += ev.device_time_total if ev.device_time_total else (ev.time_range.end - ev.time_range.start)
tot = sum(v[1] for v in kern.values())
print("device kernels: %.1f ms in %d launches" % (tot / 1e3, sum(v[0] for v in kern.values())))
for name, (n, t) in sorted(kern.items(), key=lambda kv: -kv[1][1])[:60]:
    print("%9.1f us  x%4d  %s" % (t, n, name))
byshape = collections.defaultdict(lambda: [0, 0.0])
for ev in prof.events():
    if ev.device_time_total > 0 and ev.name.startswith("aten::") and not any((c.name or "").startswith("aten::") and c.device_time_total > 0 for c in (ev.cpu_children or [])):
        k = (ev.name, str([tuple(s_) for s_ in (ev.input_shapes or []) if s_][:2]))
        byshape[k][0] += 1
        byshape[k][1] += ev.device_time_total
print("leaf aten ops by input shapes:")
for (name, shp), (n, t) in sorted(byshape.items(), key=lambda kv: -kv[1][1])[:45]:
    print("%9.1f us  x%4d  %-28s %s" % (t, n, name, shp))
