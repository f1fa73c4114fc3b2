"""cProfile of the training step's forward alone (host side): own time and cumulative time."""
import cProfile, pstats, sys, io
sys.path.insert(0, '.')
import torch
import hermnet_amd as hn
from hermnet_amd import synth
dev = torch.device('cuda:0')
kw = dict(rc=5.0, num_layers=5, hidden_channels=128, num_rbf=128)
d = synth.molecule_batch(num_graphs=1024).to(dev)
model = hn.HVNet(["H", "C", "O"], **kw)
model.load_state_dict(synth.synth_state_dict(model.state_dict(), 12))
model = model.to(dev).train()
def fwd():
    model.zero_grad(); d.pos.requires_grad_(True)
    return model(d)
for _ in range(3): fwd()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    fwd()
    torch.cuda.synchronize()
pr.disable()
for key in ("tottime", "cumtime"):
    st = io.StringIO()
    pstats.Stats(pr, stream=st).sort_stats(key).print_stats(45)
    print(st.getvalue()[:9000])
