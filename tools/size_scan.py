import sys, time, torch
sys.path.insert(0, '.')
import hermnet_amd as hn
from hermnet_amd import synth
dev = torch.device('cuda:0')
kw = dict(rc=5.0, num_layers=5, hidden_channels=128, num_rbf=128)
model = hn.HVNet(["Al", "Ni", "Cu"], **kw).eval()
model.load_state_dict(synth.synth_state_dict(model.state_dict(), 10))
model = model.to(dev)
for p in model.parameters(): p.requires_grad_(False)
for nz in (25, 34, 66, 129, 250):
    d = synth.fcc_alloy(reps=(10, 10, nz), seed=0, device=dev)
    def step():
        d.pos.requires_grad_(True)
        e = model(d)
        return -torch.autograd.grad(e.sum(), d.pos)[0]
    for _ in range(6): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 20
    for _ in range(n): step()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / n * 1e3
    N = d.pos.size(0)
    print("atoms %6d  %.3f ms/step  %.2f M atom-steps/s  (%.1f ns/atom)" % (N, ms, N / ms / 1e3, ms * 1e6 / N))
