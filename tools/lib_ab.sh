#!/bin/bash
# interleaved A/B of library variants on the bench's headline step:  bash tools/lib_ab.sh ROUNDS name...   ("default" = the tree's library)
R=$1; shift
for rep in $(seq $R); do
  for v in "$@"; do
    if [ $v = default ]; then L=""; else L="hermnet_amd/csrc/variants/libhermnet_$v.so"; fi
    echo "$v: $(HERMNET_LIB_PATH=$L python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-secondary 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); print("%.3f ms  mfma %s" % (d["ms_per_step"], d.get("mfma", {}).get("mfma_util")))')"
  done
done
