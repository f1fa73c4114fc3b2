#!/bin/bash
# interleaved A/B of the layer-boundary modes (HN_SWITCHES=boundary_mode=: 0 none, 4 backward fused, 3 forward fused, 1 both) on
# the bench's headline step:   bash tools/boundary_ab.sh [rounds]
R=${1:-2}
for rep in $(seq $R); do
  for m in 4 1 3 0 2; do
    echo "mode $m: $(HN_SWITCHES=boundary_mode=$m python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print("%.3f ms  E %s  mfma %s" % (d["ms_per_step"], d.get("energy_check", d.get("energy")), d.get("mfma", {}).get("mfma_util")))')"
  done
done
