"""Training step: `HVNet` in train() mode on MI355X against the
reference's own backward pass (`tests/golden/train_*.npz`, generator `gen_train_golden.py`) and the
oracle; DistributedDataParallel over whole graphs (`example/dist_train.py:63`).

Tolerance: loss terms 2e-5 relative; parameter gradients 5e-5 of the tensor's largest entry (floor:
1e-3 of the largest gradient of the model) -- second derivatives in fp32, different summation order."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

import hermnet_amd as hn
from helpers import Golden, TRAIN_CASES
from oracle import hermnet_oracle as orc

HERE = os.path.dirname(os.path.abspath(__file__))
GRAD_TOL = 5e-5


def training_step(model, data, y, ftgt, gamma):
    """`example/dist_train.py:86-99`."""
    data.pos.requires_grad_(True)
    pred_e = model(data)
    e_loss = F.mse_loss(pred_e, y)
    pred_f = -torch.autograd.grad(pred_e.sum(), data.pos, create_graph=True)[0]
    f_loss = F.mse_loss(pred_f, ftgt)
    loss = (1 - gamma) * e_loss + gamma * f_loss
    loss.backward()
    return loss.detach(), e_loss.detach(), f_loss.detach()


def assert_grads_close(named_grads, ref, tol=GRAD_TOL):
    gmax = max(float(v.abs().max()) for v in ref.values())
    assert set(k for k, v in named_grads.items() if v is not None) == set(ref.keys())
    worst = 0.0
    for k, r in ref.items():
        scale = max(float(r.abs().max()), 1e-3 * gmax)
        err = float((named_grads[k].detach().cpu() - r).abs().max()) / scale
        worst = max(worst, err)
        assert err < tol, (k, err)
    return worst


@pytest.mark.gpu
@pytest.mark.parametrize("name", TRAIN_CASES)
def test_training_step_matches_reference_golden(name):
    dev = torch.device("cuda:0")
    g = Golden(name)
    model = g.model().to(dev).train()
    y, ftgt, gamma, (loss, e_loss, f_loss), grads = g.training()
    l, le, lf = training_step(model, g.data().to(dev), y.to(dev), ftgt.to(dev), gamma)
    # e_loss = mean((E - y)^2) with |E - y| ~ 0.5: an energy error of 1e-5 |E| moves it by ~1e-5 |E|
    emax = max(1.0, float(g.energy.abs().max()))
    assert abs(float(le) - e_loss) < 2e-5 * emax * max(1.0, e_loss)
    assert abs(float(lf) - f_loss) < 2e-5 * max(1.0, f_loss)
    assert abs(float(l) - loss) < 2e-5 * emax * max(1.0, loss)
    assert_grads_close({k: p.grad for k, p in model.named_parameters()}, grads)


@pytest.mark.gpu
def test_training_step_matches_oracle_default_width():
    """H = R = 128, 3 layers, molecule batch (the shape of `dist_train.py`'s MD17 batches): oracle on the CPU."""
    dev = torch.device("cuda:0")
    g = Golden("mol16")
    model = g.model().to(dev).train()
    d = g.data()
    gen = torch.Generator().manual_seed(5)
    y = g.energy + 0.5 * torch.randn(g.energy.numel(), generator=gen)
    ftgt = 0.5 * torch.randn(d.pos.shape, generator=gen)
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    lo, leo, lfo, og = orc.training_loss_and_grads(sd, g.elems, d, y, ftgt, 0.8, **g.oracle_kwargs())
    l, le, lf = training_step(model, d.to(dev), y.to(dev), ftgt.to(dev), 0.8)
    assert abs(float(l) - float(lo)) < 2e-5 * max(1.0, float(g.energy.abs().max())) * max(1.0, float(lo))
    assert_grads_close({k: p.grad for k, p in model.named_parameters()}, {k: v for k, v in og.items() if v is not None})


@pytest.mark.gpu
def test_eval_mode_after_training_uses_fused_path_and_same_numbers():
    """train() and eval() are the same function of (weights, data): energies and forces agree."""
    dev = torch.device("cuda:0")
    g = Golden("train_mol8_h64")
    model = g.model().to(dev)
    out = {}
    for mode in ("train", "eval"):
        getattr(model, mode)()
        d = g.data().to(dev)
        d.pos.requires_grad_(True)
        e = model(d)
        f = -torch.autograd.grad(e.sum(), d.pos)[0]
        out[mode] = (e.detach(), f)
    assert float((out["train"][0] - out["eval"][0]).abs().max()) < 1e-5 * float(out["eval"][0].abs().max())
    assert float((out["train"][1] - out["eval"][1]).abs().max()) < 1e-5 * float(out["eval"][1].abs().max())


@pytest.mark.gpu
def test_train_and_eval_agree_with_an_atom_of_an_unknown_element():
    """The train() path sizes its edge arrays by the number of edges with a KNOWN target: with every element in the model that is
    the host's edge count (no device read, RelationalGraph.rel_edge_total); an atom of an element the model does not have
    (its row stays zero, hermnet.py:51) takes the read-back route.  Both against eval() on the same data."""
    dev = torch.device("cuda:0")
    g = Golden("train_mol8_h64")
    model = g.model().to(dev)
    for unknown in (False, True):
        out = {}
        for mode in ("train", "eval"):
            getattr(model, mode)()
            d = g.data().to(dev)
            if unknown:
                z = d.atomic_number.clone()
                z[3] = 79                                # gold: not an element of this model
                d.atomic_number = z
            d.pos.requires_grad_(True)
            e = model(d)
            f = -torch.autograd.grad(e.sum(), d.pos)[0]
            out[mode] = (e.detach(), f, d._hn_graph._all_known)
        assert out["train"][2] == (not unknown)
        assert float((out["train"][0] - out["eval"][0]).abs().max()) < 1e-5 * float(out["eval"][0].abs().max())
        assert float((out["train"][1] - out["eval"][1]).abs().max()) < 1e-5 * float(out["eval"][1].abs().max())


def test_training_mode_refuses_host_tensors():
    g = Golden("train_mol8_h64")
    with pytest.raises(RuntimeError):
        g.model().train()(g.data())            # host tensors: no CPU path, in training mode either


def _ddp_worker(rank, world, name, port, out):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)      # ranks share the one GPU of the box
    try:
        dev = torch.device("cuda:0")
        g = Golden(name)
        model = torch.nn.parallel.DistributedDataParallel(g.model().to(dev).train())
        y, ftgt, gamma, _, _ = g.training()
        d = g.data()
        # rank r trains on graphs r, r + world, ... (whole graphs per rank, dist_train.py:57)
        keep_g = torch.arange(int(d.batch.max()) + 1)[rank::world]
        amask = torch.isin(d.batch, keep_g)
        new_id = torch.full((d.pos.size(0),), -1, dtype=torch.long)
        new_id[amask] = torch.arange(int(amask.sum()))
        emask = amask[d.edge_index[0]]
        local = hn.Data(pos=d.pos[amask], atomic_number=d.atomic_number[amask],
                        batch=torch.searchsorted(keep_g.contiguous(), d.batch[amask]), edge_index=new_id[d.edge_index[:, emask]])
        training_step(model, local.to(dev), y[keep_g].to(dev), ftgt[amask].to(dev), gamma)
        out[rank] = {k: p.grad.detach().cpu().numpy() for k, p in model.module.named_parameters()}
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_ddp_gradients_are_the_rank_average():
    """DDP (gloo here; `nccl` = RCCL on a multi-GPU node) averages the per-rank gradients: both ranks end
    with the same gradients, equal to the mean of the two single-process half-batch gradients."""
    name, world = "train_mol8_h64", 2
    port = 33500 + os.getpid() % 2000
    out = mp.Manager().dict()
    mp.spawn(_ddp_worker, args=(world, name, port, out), nprocs=world, join=True)
    for k in out[0]:
        assert np.array_equal(out[0][k], out[1][k]), k
    # single-process reference of the same two half batches
    dev = torch.device("cuda:0")
    g = Golden(name)
    y, ftgt, gamma, _, _ = g.training()
    d = g.data()
    acc = None
    for rank in range(world):
        model = g.model().to(dev).train()
        keep_g = torch.arange(int(d.batch.max()) + 1)[rank::world]
        amask = torch.isin(d.batch, keep_g)
        new_id = torch.full((d.pos.size(0),), -1, dtype=torch.long)
        new_id[amask] = torch.arange(int(amask.sum()))
        emask = amask[d.edge_index[0]]
        local = hn.Data(pos=d.pos[amask], atomic_number=d.atomic_number[amask],
                        batch=torch.searchsorted(keep_g.contiguous(), d.batch[amask]), edge_index=new_id[d.edge_index[:, emask]])
        training_step(model, local.to(dev), y[keep_g].to(dev), ftgt[amask].to(dev), gamma)
        gr = {k: p.grad.detach().cpu() / world for k, p in model.named_parameters()}
        acc = gr if acc is None else {k: acc[k] + gr[k] for k in gr}
    # (index_add in the training path uses atomics: run-to-run differences of a few 1e-6)
    assert_grads_close({k: torch.from_numpy(v) for k, v in out[0].items()}, acc)
