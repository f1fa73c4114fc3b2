"""Host side of the training step (train() mode): per phase (forward, create_graph force pass, loss.backward) the wall time with a
device sync, the time the host needs to enqueue it, the device launches; then cProfile of a few steps (own time)."""
import cProfile, pstats, sys, io, time, collections
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
sync = torch.cuda.synchronize


def phases(record=None):
    out = []
    def timed(name, fn):
        sync(); t0 = time.perf_counter(); r = fn(); t1 = time.perf_counter(); sync(); t2 = time.perf_counter()
        out.append((name, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
        return r
    model.zero_grad()
    d.pos.requires_grad_(True)
    e = timed("forward", lambda: model(d))
    f = timed("force pass", lambda: -torch.autograd.grad(e.sum(), d.pos, create_graph=True)[0])
    loss = timed("loss", lambda: 0.2 * F.mse_loss(e, y) + 0.8 * F.mse_loss(f, ftgt))
    timed("backward", lambda: loss.backward())
    return out


def step():
    model.zero_grad()
    d.pos.requires_grad_(True)
    e = model(d)
    f = -torch.autograd.grad(e.sum(), d.pos, create_graph=True)[0]
    loss = 0.2 * F.mse_loss(e, y) + 0.8 * F.mse_loss(f, ftgt)
    loss.backward()
    return loss


for _ in range(3): step()
sync()
t0 = time.perf_counter()
for _ in range(5): step()
t1 = time.perf_counter(); sync(); t2 = time.perf_counter()
print("5 steps back to back: host returns after %.1f ms per step, device done after %.1f ms per step" % ((t1 - t0) * 200, (t2 - t0) * 200))
for _ in range(2): ph = phases()
print("phase            host enqueue ms   wall ms (synced)")
for name, h, w in ph:
    print("%-16s %12.2f %14.2f" % (name, h, w))
# launches per phase
for name, fn in (("forward", None),):
    pass
counts = collections.OrderedDict()
model.zero_grad(); d.pos.requires_grad_(True)
def launches(fn):
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        r = fn(); sync()
    n = sum(1 for ev in prof.events() if str(ev.device_type).endswith("CUDA"))
    t = sum((ev.device_time_total or 0) for ev in prof.events() if str(ev.device_type).endswith("CUDA"))
    return r, n, t
e, n, t = launches(lambda: model(d)); print("forward     : %4d launches, %.2f ms of kernels" % (n, t / 1e3))
f, n, t = launches(lambda: -torch.autograd.grad(e.sum(), d.pos, create_graph=True)[0]); print("force pass  : %4d launches, %.2f ms of kernels" % (n, t / 1e3))
loss = 0.2 * F.mse_loss(e, y) + 0.8 * F.mse_loss(f, ftgt)
_, n, t = launches(lambda: loss.backward()); print("backward    : %4d launches, %.2f ms of kernels" % (n, t / 1e3))
pr = cProfile.Profile()
sync()
pr.enable()
for _ in range(5): step()
pr.disable()
sync()
for key in ("tottime", "cumtime"):
    st = io.StringIO()
    pstats.Stats(pr, stream=st).sort_stats(key).print_stats(40)
    print(st.getvalue()[:7000])
