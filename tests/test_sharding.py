"""Atom-sharded path on CPU with gloo, world_size 2 and 3:
partition + halo exchange + energy all-reduce must reproduce the single-process result.
The kernels are replaced by their PyTorch restatements (tests/ref_ops.py), as in
test_host_logic.py -- what is tested here is the distributed plumbing."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import Golden, rel_err

HERE = os.path.dirname(os.path.abspath(__file__))


def _patch_cpu_ops():
    import hermnet_amd.hermnet as hmod
    import hermnet_amd.layer as lmod
    import ref_ops
    hmod.HVNet._require_device = staticmethod(lambda pos: None)
    hmod.EdgeGeometry = ref_ops.RefEdgeGeometry
    for fn in ["energy_head_fwd", "energy_head_bwd", "layernorm_fwd", "layernorm_bwd", "ssilu_fwd", "ssilu_bwd", "update_mid", "update_out", "update_out_bwd",
               "update_mid_bwd"]:
        setattr(lmod.nodeops, fn, getattr(ref_ops, fn))
    lmod._msg_fwd = ref_ops.msg_fwd
    lmod._msg_bwd = ref_ops.msg_bwd


def _worker(rank, world, name, port, out):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        _patch_cpu_ops()
        from hermnet_amd.sharding import partition
        g = Golden(name)
        model = g.model()
        for p in model.parameters():
            p.requires_grad_(False)
        local, plan = partition(g.data(), rank, world)
        local.pos.requires_grad_(True)
        e = model(local)
        f_local = -torch.autograd.grad(e.sum(), local.pos)[0]
        # forces of owned atoms arrive complete at their owner; halo rows carry nothing
        assert float(f_local[~plan.owned_mask].abs().max()) == 0.0 if f_local.size(0) > plan.n_owned else True
        out[rank] = (e.detach().numpy(), plan.owned_global.numpy(), f_local[plan.owned_local].numpy(),
                     int(plan.halo_global.numel()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("alloy108", 2), ("alloy108", 3), ("mol16", 2), ("c1_si64", 2)])
def test_sharded_energy_and_forces_match_single_process(name, world):
    port = 29500 + (os.getpid() + hash((name, world))) % 2000
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, name, port, out), nprocs=world, join=True)
    g = Golden(name)
    forces = np.zeros_like(g.forces.numpy())
    seen = np.zeros(forces.shape[0], dtype=int)
    for r in range(world):
        e, owned, f, nhalo = out[r]
        assert rel_err(torch.from_numpy(e), g.energy) < 5e-6, (r, e, g.energy)
        forces[owned] = f
        seen[owned] += 1
        if name != "mol16":
            assert nhalo > 0
    assert (seen == 1).all()                       # every atom owned exactly once
    assert rel_err(torch.from_numpy(forces), g.forces) < 2e-5


def test_partition_plans_are_consistent():
    from hermnet_amd.sharding import partition
    g = Golden("alloy108")
    d = g.data()
    world = 4
    parts = [partition(g.data(), r, world) for r in range(world)]
    n_edges = sum(p[0].edge_index.size(1) for p in parts)
    assert n_edges == d.edge_index.size(1)          # every edge lives on exactly one rank (its target's)
    for r, (loc, plan) in enumerate(parts):
        assert plan.atom_plan.recv_counts[r] == 0 and plan.atom_plan.send_counts[r] == 0
        for p in range(world):
            # what r sends to p is what p expects from r, in the same order
            send_idx = plan.atom_plan.send_idx[sum(plan.atom_plan.send_counts[:p]):sum(plan.atom_plan.send_counts[:p + 1])]
            assert bool(plan.owned_mask[send_idx].all())           # only owned atoms are sent
            sent = plan.local_global[send_idx]
            q = parts[p][1]
            off = sum(q.atom_plan.recv_counts[:r])
            recv_idx = q.atom_plan.recv_idx[off:off + q.atom_plan.recv_counts[r]]
            assert not bool(q.owned_mask[recv_idx].any())           # only halo atoms are received
            assert torch.equal(sent, q.local_global[recv_idx])
        # owned and halo atoms are interleaved in ascending global id
        assert torch.equal(plan.local_global, torch.sort(plan.local_global).values)
        assert torch.equal(plan.local_global[plan.owned_local], plan.owned_global)


def _gpu_worker(rank, world, name, port, out):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)     # exchange staged through the host
    try:
        from hermnet_amd.sharding import partition
        dev = torch.device("cuda:0")                                    # ranks share the one GPU of the box
        g = Golden(name)
        model = g.model().to(dev)
        for p in model.parameters():
            p.requires_grad_(False)
        local, plan = partition(g.data(), rank, world)
        local = local.to(dev)
        local.pos.requires_grad_(True)
        e = model(local)
        f_local = -torch.autograd.grad(e.sum(), local.pos)[0]
        out[rank] = (e.detach().cpu().numpy(), plan.owned_global.cpu().numpy(),
                     f_local[plan.owned_local].cpu().numpy(), int(plan.halo_global.numel()))
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("name,world", [("alloy108", 2), ("mol16", 3)])
def test_sharded_hip_path_matches_reference_golden(name, world):
    """The HIP kernels in atom-sharded mode (2-3 ranks sharing the GPU, host-staged exchange) vs the
    reference's energies/forces; the RCCL exchange differs only in the collective call."""
    port = 31500 + (os.getpid() + hash((name, world))) % 2000
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_gpu_worker, args=(world, name, port, out), nprocs=world, join=True)
    g = Golden(name)
    forces = np.zeros_like(g.forces.numpy())
    for r in range(world):
        e, owned, f, nhalo = out[r]
        assert rel_err(torch.from_numpy(e), g.energy) < 1e-5
        forces[owned] = f
    assert rel_err(torch.from_numpy(forces), g.forces) < 1e-5


def _rccl_worker(rank, world, name, port, out):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)      # "nccl" = RCCL on ROCm
    try:
        from hermnet_amd.sharding import partition
        g = Golden(name)
        model = g.model().to(dev)
        for p in model.parameters():
            p.requires_grad_(False)
        local, plan = partition(g.data(), rank, world)
        local = local.to(dev)
        local.pos.requires_grad_(True)
        e = model(local)
        f_local = -torch.autograd.grad(e.sum(), local.pos)[0]
        dist.barrier()
        out[rank] = (e.detach().cpu().numpy(), f_local[plan.owned_local].cpu().numpy(), plan.owned_global.cpu().numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_backend_single_rank_smoke():
    """The production backend ("nccl" = RCCL) on the one GPU of the box: world size 1, so the exchanges are empty,
    but every collective of the sharded path (variable-size all_to_all_single on device buffers, all_reduce,
    barrier) goes through RCCL with the same calls a multi-GPU run makes."""
    name = "alloy108"
    port = 35500 + os.getpid() % 2000
    out = mp.Manager().dict()
    mp.spawn(_rccl_worker, args=(1, name, port, out), nprocs=1, join=True)
    g = Golden(name)
    e, f, owned = out[0]
    forces = np.zeros_like(g.forces.numpy())
    forces[owned] = f
    assert rel_err(torch.from_numpy(e), g.energy) < 1e-5
    assert rel_err(torch.from_numpy(forces), g.forces) < 1e-5
