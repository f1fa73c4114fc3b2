#!/usr/bin/env python3
"""Device neighbour search of the configs[1] cell, timed (wall, synchronised) and profiled:  python tools/nbr_bench.py [reps]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hermnet_amd import synth  # noqa: E402
from hermnet_amd.neighbor import neighbor_search  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda")
d = synth.fcc_alloy(reps=(10, 10, 25), seed=0, device=dev)
pos, cell = d.pos.detach(), d.cell
for _ in range(3):
    ei, sh = neighbor_search(pos, 5.0, cell)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    ei, sh = neighbor_search(pos, 5.0, cell)
torch.cuda.synchronize()
print("neighbor_search: %.3f ms per call, E = %d" % ((time.perf_counter() - t0) / reps * 1e3, ei.size(1)))
