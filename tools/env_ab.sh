#!/bin/bash
# interleaved A/B of an environment switch on the bench's headline step:  bash tools/env_ab.sh VAR ROUNDS value...
V=$1; R=$2; shift; shift
for rep in $(seq $R); do
  for x in "$@"; do
    echo "$V=$x: $(env $V=$x python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); k=d["kernels"]; print("%.3f ms  fwd %.1f us  fwd_l0 %.1f us  E %s" % (d["ms_per_step"], k["message_scatter_fwd"]["avg_ms"]*1e3, k["message_scatter_fwd_l0"]["avg_ms"]*1e3, d["energy"]))')"
  done
done
