"""120 training steps with Adam on the bench's molecule batch: the loss falls, the allocator's figures stay put (no per-step growth)."""
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
opt = torch.optim.Adam(model.parameters(), lr=3e-4)
gen = torch.Generator().manual_seed(0)
y = torch.randn(1024, generator=gen).to(dev); ftgt = (0.5 * torch.randn(d.pos.shape, generator=gen)).to(dev)
for i in range(120):
    opt.zero_grad(); d.pos.requires_grad_(True)
    e = model(d); f = -torch.autograd.grad(e.sum(), d.pos, create_graph=True)[0]
    loss = 0.2 * F.mse_loss(e, y) + 0.8 * F.mse_loss(f, ftgt)
    loss.backward(); opt.step()
    if i % 20 == 0 or i == 119:
        torch.cuda.synchronize()
        print(i, "loss %.5f" % float(loss), "allocated %.2f GB, reserved %.2f GB, peak %.2f GB" % (torch.cuda.memory_allocated() / 2**30, torch.cuda.memory_reserved() / 2**30, torch.cuda.max_memory_allocated() / 2**30))
