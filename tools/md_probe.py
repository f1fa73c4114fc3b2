import os, sys, time, torch
sys.path.insert(0, ".")
import hermnet_amd as hn
from hermnet_amd import synth
from hermnet_amd.neighbor import neighbor_search
dev = torch.device("cuda")
kw = dict(rc=5.0, num_layers=5, hidden_channels=128, num_rbf=128)
model = hn.HVNet(["Al", "Ni", "Cu"], **kw).eval()
model.load_state_dict(synth.synth_state_dict(model.state_dict(), 10))
model = model.to(dev)
for p in model.parameters(): p.requires_grad_(False)
data = synth.fcc_alloy(reps=(10, 10, 25), seed=0, device=dev)
pos0, cell0 = data.pos.detach(), data.cell
def step(d):
    d.pos.requires_grad_(True)
    e = model(d)
    return -torch.autograd.grad(e.sum(), d.pos)[0]
def md():
    ei, sh = neighbor_search(pos0, 5.0, cell0)
    d = hn.Data(pos=pos0.clone(), atomic_number=data.atomic_number, batch=data.batch, cell=cell0, edge_index=ei, edge_shift=sh)
    return step(d)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t)/n*1e3
def host_only(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(n): fn()
    h=(time.perf_counter()-t)/n*1e3
    torch.cuda.synchronize(); return h
print(os.environ.get("TAG",""), "plain %.3f ms (host enqueue %.3f)  md %.3f ms" % (timeit(lambda: step(data)), host_only(lambda: step(data)), timeit(md)))
