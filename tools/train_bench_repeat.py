"""bench.py's training secondary three times in one process (tuned / default GEMM solutions per repetition): run-to-run spread.
    python tools/train_bench_repeat.py [timed steps]"""
import sys, json
sys.path.insert(0, '.')
import torch, bench
import hermnet_amd as hn
from hermnet_amd import synth
dev = torch.device('cuda:0')
kw = dict(rc=5.0, num_layers=5, hidden_channels=128, num_rbf=128)
for i in range(3):
    r = bench.training_secondary(hn, synth, dev, kw, steps=int(sys.argv[1]) if len(sys.argv) > 1 else 3)
    print(round(r["ms_per_step"], 2), round(r["ms_per_step_default_gemm_solutions"], 2), r["gemm_selection"][-40:])
