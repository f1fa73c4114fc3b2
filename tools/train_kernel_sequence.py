"""The device launches of one training step in order (name, duration), per phase: for reading the step like a timeline."""
import sys
sys.path.insert(0, '.')
import torch
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
state = {}
def fwd():
    model.zero_grad(); d.pos.requires_grad_(True)
    state["e"] = model(d)
def force():
    state["f"] = -torch.autograd.grad(state["e"].sum(), d.pos, create_graph=True)[0]
def bwd():
    loss = 0.2 * F.mse_loss(state["e"], y) + 0.8 * F.mse_loss(state["f"], ftgt)
    loss.backward()
for _ in range(2):
    fwd(); force(); bwd()
torch.cuda.synchronize()
for name, fn in (("forward", fwd), ("force pass", force), ("backward", bwd)):
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        fn(); torch.cuda.synchronize()
    rows = []
    for ev in prof.events():
        if str(ev.device_type).endswith("CUDA") or not ev.kernels:
            continue
        if any(c.kernels for c in (ev.cpu_children or [])):
            continue
        shp = str([tuple(s_) for s_ in (ev.input_shapes or []) if s_][:3])
        for k in ev.kernels:
            rows.append((ev.time_range.start, k.duration, ev.name, k.name[:70], shp))
    rows.sort()
    print("==== %s: %d launches, %.2f ms" % (name, len(rows), sum(r[1] for r in rows) / 1e3))
    for _, dur, op, kn, shp in rows:
        print("%8.1f  %-28s %-70s %s" % (dur, op[:28], kn, shp[:90]))
