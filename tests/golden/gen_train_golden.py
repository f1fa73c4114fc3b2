#!/usr/bin/env python3
"""Golden vectors for the TRAINING step.  RUN ONLY IN THE BUILD
CONTAINER (needs /root/reference); same harness and stand-ins as `gen_golden.py`.

Executes the reference's own HVNet in train() mode through the loss of
`example/dist_train.py:86-99` (MSE energy + MSE force with create_graph=True, gamma = 0.8) and stores
the loss terms and the gradient of EVERY parameter.  Targets are seeded synthetic numbers
(`training_targets`); weights are `synth_state_dict(seed)` as in the other fixtures.

    python tests/golden/gen_train_golden.py
"""
import hashlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import gen_golden as gg  # noqa: E402  (sets up sys.path for the reference and the stand-ins)
from hermnet_amd import synth  # noqa: E402

GAMMA = 0.8


def training_targets(pred_e, num_atoms, seed):
    """Deterministic regression targets: y [num_graphs] = the untrained prediction + N(0, 0.5) (so that the
    energy and the force term of the loss -- first- and second-order gradients -- have similar weight),
    forces [num_atoms, 3] ~ N(0, 0.5).  Both are stored in the fixture."""
    g = torch.Generator().manual_seed(1000 + seed)
    return pred_e + 0.5 * torch.randn(pred_e.numel(), generator=g), 0.5 * torch.randn(num_atoms, 3, generator=g)


def cases():
    return [
        ("train_alloy108_h64", synth.fcc_alloy(reps=(3, 3, 3)), ["Al", "Ni", "Cu"],
         dict(num_layers=2, rc=5.0, hidden_channels=64, num_rbf=32), 11),
        ("train_mol8_h64", synth.molecule_batch(num_graphs=8), ["H", "C", "O"],
         dict(num_layers=2, rc=5.0, hidden_channels=64, num_rbf=32), 12),
        ("train_si64_intensive_h64", synth.si_diamond(), ["Si"],
         dict(num_layers=2, rc=5.0, hidden_channels=64, num_rbf=32, intensive=True), 13),
        # width 128 (the width of BASELINE configs[1] and of the training kernels' tuned instances); one relation keeps the
        # fixture at 1.5 MB (every parameter's gradient is stored)
        ("train_si64_h128", synth.si_diamond(), ["Si"],
         dict(num_layers=2, rc=5.0, hidden_channels=128, num_rbf=32), 15),
    ]


def run_reference(d, elems, model_kw, seed):
    ref = gg.ref_hermnet
    torch.manual_seed(0)
    model = ref.HVNet(elems, **model_kw)
    sd = synth.synth_state_dict(model.state_dict(), seed)
    model.load_state_dict(sd)
    model.train()
    model.eval()
    with torch.no_grad():
        y, ftgt = training_targets(model(gg.to_ref(d)).detach(), d.pos.size(0), seed)
    model.train()
    crit = torch.nn.MSELoss()
    orig = ref.HVNet.with_edge
    ref.HVNet.with_edge = gg._with_edge_out_of_place      # see gen_golden.py: in-place write breaks autograd
    try:
        rd = gg.to_ref(d)
        rd.pos.requires_grad = True
        pred_e = model(rd)                                                       # dist_train.py:89
        e_loss = crit(pred_e, y)                                                 # :90
        pred_f = -torch.autograd.grad(pred_e.sum(), rd.pos, create_graph=True)[0]   # :92-94
        f_loss = crit(pred_f, ftgt)                                              # :95
        loss = (1 - GAMMA) * e_loss + GAMMA * f_loss                             # :97
        loss.backward()                                                          # :99
    finally:
        ref.HVNet.with_edge = orig
    grads = {k: (None if p.grad is None else p.grad.detach().clone()) for k, p in model.named_parameters()}
    return sd, y, ftgt, loss.item(), e_loss.item(), f_loss.item(), pred_e.detach(), pred_f.detach(), grads


def main():
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    for name, d, elems, kw, seed in cases():
        if only and only != name:
            continue
        sd, y, ftgt, loss, e_loss, f_loss, e, f, grads = run_reference(d, elems, kw, seed)
        meta = dict(name=name, elems=elems, model_kw=kw, weight_seed=seed, sd_sha256=gg.sd_checksum(sd), gamma=GAMMA,
                    num_edges=int(d.edge_index.size(1)), torch=torch.__version__,
                    no_grad_params=sorted(k for k, g in grads.items() if g is None),
                    edge_index_sha256=hashlib.sha256(d.edge_index.numpy().tobytes()).hexdigest())
        arrays = dict(pos=d.pos.numpy(), atomic_number=d.atomic_number.numpy(), batch=d.batch.numpy(),
                      edge_index=d.edge_index.numpy().astype(np.int32), y=y.numpy(), force_target=ftgt.numpy(),
                      energy=e.numpy(), forces=f.numpy(), loss=np.array([loss, e_loss, f_loss], dtype=np.float64),
                      meta=np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8))
        if d.get("cell") is not None:
            arrays["cell"] = d.cell.numpy()
            arrays["edge_shift"] = d.edge_shift.numpy().astype(np.int8)
        for k, g in grads.items():
            if g is not None:
                arrays["grad:" + k] = g.numpy()
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **arrays)
        gmax = max(float(g.abs().max()) for g in grads.values() if g is not None)
        print("%-28s N=%-5d E=%-6d loss=%.6f (e %.6f, f %.6f) max|grad|=%.3e  params without grad: %d  %d KB" % (
            name, d.pos.size(0), d.edge_index.size(1), loss, e_loss, f_loss, gmax, len(meta["no_grad_params"]),
            os.path.getsize(path) // 1024))


if __name__ == "__main__":
    main()
