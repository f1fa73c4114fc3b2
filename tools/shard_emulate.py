"""Per-rank cost of the atom-sharded step WITHOUT peers: one rank's share of an N-way slab decomposition of the
weak-scaling cell (fcc 10 x 10 x 25N) runs alone, with the halo exchange replaced by a local stand-in of the same
shape (zeros for what the neighbours would send).  Everything else -- halo rows in the node GEMMs, pack / unpack /
accumulate kernels, owned-atom read-out -- is the real sharded code path, so  t(unsharded) / t(this)  bounds the
weak-scaling efficiency from above (the RCCL latency of ~10 small all-to-alls per step comes on top).

    python tools/shard_emulate.py [world=8] [rank=3]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import hermnet_amd as hn  # noqa: E402
from hermnet_amd import sharding, synth  # noqa: E402
from hermnet_amd.utils import enable_tuned_gemms  # noqa: E402


def main():
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    rank = int(sys.argv[2]) if len(sys.argv) > 2 else world // 2
    dev = torch.device("cuda:0")
    mode = os.environ.get("EMUL_GEMM", "tuned")
    if mode == "tuned":
        enable_tuned_gemms()
    elif mode == "rocblas":
        torch.backends.cuda.preferred_blas_library("cublas")
    elif mode == "hipblaslt":
        torch.backends.cuda.preferred_blas_library("cublaslt")
    elif mode == "online":      # TunableOp tunes every new shape during the warm-up steps
        enable_tuned_gemms(online=True)
    kw = dict(rc=5.0, num_layers=5, hidden_channels=128, num_rbf=128)
    model = hn.HVNet(["Al", "Ni", "Cu"], **kw).eval()
    model.load_state_dict(synth.synth_state_dict(model.state_dict(), 10))
    model = model.to(dev)
    for p in model.parameters():
        p.requires_grad_(False)

    def timed(data, steps=40, warm=8):
        def step():
            data.pos.requires_grad_(True)
            e = model(data)
            return -torch.autograd.grad(e.sum(), data.pos)[0]
        for _ in range(warm):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / steps * 1e3

    single = synth.fcc_alloy(reps=(10, 10, 25), seed=0, device=dev)
    t1 = timed(single)
    g = synth.fcc_alloy(reps=(10, 10, 25 * world), seed=0, device=dev)
    local, plan = sharding.partition(g.to("cpu"), rank, world)
    del g
    local = local.to(dev)
    # stand-ins for the collectives: same shapes, no peers
    sharding._all_to_all_rows = lambda buf, in_counts, out_counts, group: buf.new_zeros((sum(out_counts),) + tuple(buf.shape[1:]))
    sharding.SumAcrossRanks.forward = staticmethod(lambda ctx, e, group: e.detach().clone())
    tN = timed(local)
    print("unsharded 10k-atom step: %.3f ms" % t1)
    print("rank %d of %d: %d owned + %d halo atoms, %d edges: %.3f ms  -> weak-scaling efficiency bound %.1f %%"
          % (rank, world, plan.n_owned, int(plan.halo_global.numel()), local.edge_index.size(1), tN, 100 * t1 / tN))


if __name__ == "__main__":
    main()
