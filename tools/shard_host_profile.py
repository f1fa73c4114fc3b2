"""Host-side cost of the atom-sharded step on ONE rank (world 1, RCCL): cProfile over 20 steps.
Run: MASTER_ADDR=127.0.0.1 MASTER_PORT=29541 RANK=0 WORLD_SIZE=1 python tools/shard_host_profile.py"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, '.')
import numpy as np, torch, torch.distributed as dist
import hermnet_amd as hn
from hermnet_amd import synth
from hermnet_amd.sharding import SlabStepper
dev = torch.device('cuda:0'); torch.cuda.set_device(dev)
dist.init_process_group('nccl', device_id=dev)
kw = dict(rc=5.0, num_layers=5, hidden_channels=128, num_rbf=128)
model = hn.HVNet(['Al', 'Ni', 'Cu'], **kw).eval()
model.load_state_dict(synth.synth_state_dict(model.state_dict(), 10)); model = model.to(dev)
for p in model.parameters(): p.requires_grad_(False)
pos, cell, z = synth.fcc_alloy_atoms(reps=(10, 10, 25), seed=0)
gpos = torch.from_numpy(pos.astype(np.float32)).to(dev); gcell = torch.from_numpy(cell.astype(np.float32)).to(dev); gz = torch.from_numpy(z).to(dev)
st = SlabStepper(gz, gcell, 5.0, 0, 1, skin=1.0, group=dist.group.WORLD)
data, plan = st(gpos)
def step(d):
    d.pos.requires_grad_(True)
    e = model(d)
    return e, -torch.autograd.grad(e.sum(), d.pos)[0]
for _ in range(5): step(data)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step(data)
th = time.perf_counter() - t0
torch.cuda.synchronize()
print("host enqueue %.3f ms/step, wall %.3f ms/step" % (th / 20 * 1e3, (time.perf_counter() - t0) / 20 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step(data)
pr.disable(); torch.cuda.synchronize()
ps = pstats.Stats(pr); ps.sort_stats('tottime').print_stats(28)
ps.sort_stats('cumulative').print_stats(45)
dist.destroy_process_group()
