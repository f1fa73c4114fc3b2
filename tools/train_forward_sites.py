"""Device launches of the training step's FORWARD by function (record_function scopes around the train()-mode functions)."""
import sys, collections, functools
sys.path.insert(0, '.')
import torch
import hermnet_amd as hn
from hermnet_amd import synth, rmnet, trainops, hermnet
from torch.profiler import profile, ProfilerActivity, record_function
dev = torch.device('cuda:0')
kw = dict(rc=5.0, num_layers=5, hidden_channels=128, num_rbf=128)
d = synth.molecule_batch(num_graphs=1024).to(dev)
model = hn.HVNet(["H", "C", "O"], **kw)
model.load_state_dict(synth.synth_state_dict(model.state_dict(), 12))
model = model.to(dev).train()


def scoped(owner, name, label=None):
    fn = getattr(owner, name)
    @functools.wraps(fn)
    def w(*a, **k):
        with record_function("SCOPE:" + (label or name)):
            return fn(*a, **k)
    raw = owner.__dict__.get(name) if hasattr(owner, "__dict__") else None
    setattr(owner, name, staticmethod(w) if isinstance(raw, staticmethod) else w)


scoped(hermnet.HVNet, "_build_graph")
scoped(hermnet.HVNet, "_edge_geometry_autograd")
scoped(rmnet.RadialBasis, "bucketed")
scoped(trainops.BucketedBasis, "project")
scoped(trainops, "message_scatter_generic")
scoped(rmnet, "message_scatter_generic")
scoped(rmnet, "_relational_layer_batched")
scoped(trainops, "tall_bmm")
scoped(trainops, "_row_keys")


def run():
    model.zero_grad(); d.pos.requires_grad_(True)
    return model(d)


for _ in range(2): run()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    run(); torch.cuda.synchronize()
by = collections.defaultdict(lambda: collections.defaultdict(int))
tot = collections.defaultdict(int)
for ev in prof.events():
    if str(ev.device_type).endswith("CUDA") or not ev.kernels:
        continue
    if any(c.kernels for c in (ev.cpu_children or [])):
        continue
    scope, p = "<top>", ev.cpu_parent
    while p is not None:
        if p.name.startswith("SCOPE:"):
            scope = p.name[6:]
            break
        p = p.cpu_parent
    by[scope][ev.name] += len(ev.kernels)
    tot[scope] += len(ev.kernels)
print("forward: %d launches" % sum(tot.values()))
for scope, n in sorted(tot.items(), key=lambda kv: -kv[1]):
    print("== %-28s %4d   " % (scope, n) + ", ".join("%s x%d" % (k.replace("aten::", ""), v) for k, v in sorted(by[scope].items(), key=lambda kv: -kv[1])))
