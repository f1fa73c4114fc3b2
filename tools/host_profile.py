"""Host-side (Python) cost of one energy+forces step: cProfile over a few steps, top entries by own time."""
import cProfile, pstats, sys, io
sys.path.insert(0, '.')
import torch
import hermnet_amd as hn
from hermnet_amd import synth
from hermnet_amd.utils import enable_tuned_gemms
enable_tuned_gemms()
dev = torch.device('cuda:0')
kw = dict(rc=5.0, num_layers=5, hidden_channels=128, num_rbf=128)
model = hn.HVNet(["Al", "Ni", "Cu"], **kw).eval().to(dev)
for p in model.parameters(): p.requires_grad_(False)
data = synth.fcc_alloy(device=dev)
def step():
    data.pos.requires_grad_(True)
    e = model(data)
    return -torch.autograd.grad(e.sum(), data.pos)[0]
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20): step()
pr.disable()
torch.cuda.synchronize()
st = io.StringIO()
pstats.Stats(pr, stream=st).sort_stats("tottime").print_stats(35)
print(st.getvalue()[:6000])
