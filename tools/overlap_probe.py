#!/usr/bin/env python3
"""Does an asynchronous RCCL collective really run BESIDE the kernels of the compute stream on this box?

PyTorch runs a process group's collectives on an internal stream of its own.  HIP multiplexes streams over a small number of
hardware queues (GPU_MAX_HW_QUEUES, default 4); two streams that share a queue execute in submission order -- an "asynchronous"
all_to_all then sits in FRONT of the kernels it was meant to hide behind.  The kernel trace of the self-peer step
(profiles/r06_selfpeer_trace.md) shows exactly that with the model on the default stream.

One rank over RCCL (world size 1, self send/recv), per candidate compute stream (the default stream and a few pool streams):
  t_comm     k asynchronous all_to_all_single of `mb` MB each, alone
  t_compute  a fixed chain of matmuls, alone
  t_both     the collectives started first (async_op=True), then the matmuls, then work.wait()
  overlap = (t_comm + t_compute - t_both) / min(t_comm, t_compute)     1: fully concurrent, 0: serialised
Usage: python tools/overlap_probe.py [GPU_MAX_HW_QUEUES]      (the variable has to be set before HIP starts: argv, not code)"""
import json
import os
import sys
import time

if len(sys.argv) > 1:
    os.environ["GPU_MAX_HW_QUEUES"] = sys.argv[1]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29577")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch
import torch.distributed as dist

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
sys.stdout.flush()
saved = os.dup(1)
os.dup2(2, 1)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
dist.barrier(device_ids=[0])
torch.cuda.synchronize()
os.dup2(saved, 1)

mb, k = 16, 12
src = torch.randn(mb * (1 << 20) // 4, device=dev)
dst = torch.empty_like(src)
a = torch.randn(4096, 4096, device=dev)
b = torch.randn(4096, 4096, device=dev)


def comm():
    works = [dist.all_to_all_single(dst, src, output_split_sizes=[src.numel()], input_split_sizes=[src.numel()], async_op=True)
             for _ in range(k)]
    return works


def compute():
    c = a
    for _ in range(6):
        c = c @ b
    return c


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def only_comm():
    for w in comm():
        w.wait()


def both():
    ws = comm()
    compute()
    for w in ws:
        w.wait()


out = {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES", "(default)"), "streams": []}
cands = [("default", torch.cuda.default_stream(dev))] + [("pool%d" % i, torch.cuda.Stream(dev)) for i in range(6)]
for name, s in cands:
    with torch.cuda.stream(s):
        t_comm, t_comp, t_both = timed(only_comm), timed(compute), timed(both)
    out["streams"].append({"stream": name, "t_comm_ms": round(t_comm, 3), "t_compute_ms": round(t_comp, 3),
                           "t_both_ms": round(t_both, 3),
                           "overlap": round((t_comm + t_comp - t_both) / min(t_comm, t_comp), 3)})
print(json.dumps(out, indent=1))
dist.destroy_process_group()
