"""Three training steps and nothing else: the program behind `rocprofv3 --pmc ... -- python3 tools/train_three_steps.py` (counter
passes of the training kernels; tools/collect_counters.py folds the CSVs)."""
import sys
sys.path.insert(0, '.')
import torch, torch.nn.functional as F
import hermnet_amd as hn
from hermnet_amd import synth
dev = torch.device('cuda:0')
kw = dict(rc=5.0, num_layers=5, hidden_channels=128, num_rbf=128)
d = synth.molecule_batch(num_graphs=1024).to(dev)
model = hn.HVNet(["H", "C", "O"], **kw)
model.load_state_dict(synth.synth_state_dict(model.state_dict(), 12))
model = model.to(dev).train()
gen = torch.Generator().manual_seed(0)
y = torch.randn(1024, generator=gen).to(dev); ftgt = (0.5 * torch.randn(d.pos.shape, generator=gen)).to(dev)
for _ in range(3):
    model.zero_grad(); d.pos.requires_grad_(True)
    e = model(d); f = -torch.autograd.grad(e.sum(), d.pos, create_graph=True)[0]
    (0.2 * F.mse_loss(e, y) + 0.8 * F.mse_loss(f, ftgt)).backward()
torch.cuda.synchronize()
