#!/bin/bash
# same box, same job, interleaved -- the bench's headline step and the isolated message kernels (tools/kbench.py)
# with the tree's library and with the round-4 message kernels inside the current library (tools/build_r04msg.sh).
#   bash tools/msg_regress_ab.sh ROUNDS > log
R=${1:-3}
V=hermnet_amd/csrc/variants/libhermnet_r04msg.so
line() { python -c 'import json,sys; d=json.loads(sys.stdin.read().strip().split("\n")[-1]); k=d["kernels"]; print("%.3f ms/step  bwd %.1f us  fwd %.1f us  fwd_l0 %.1f  bwd_l0 %.1f  chains %.3f ms" % (d["ms_per_step"], k["message_scatter_bwd"]["avg_ms"]*1e3, k["message_scatter_fwd"]["avg_ms"]*1e3, k["message_scatter_fwd_l0"]["avg_ms"]*1e3, k["message_scatter_bwd_l0"]["avg_ms"]*1e3, d["mfma"]["ms_per_step"]))'; }
for rep in $(seq $R); do
  echo "HEAD    in-step : $(python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-secondary 2>/dev/null | line)"
  echo "r04msg  in-step : $(HERMNET_LIB_PATH=$V HERMNET_ALLOW_STALE_LIB=1 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-secondary 2>/dev/null | line)"
done
for rep in $(seq 2); do
  echo "HEAD    isolated:"; python tools/kbench.py 40 2>/dev/null | grep message_
  echo "r04msg  isolated:"; HERMNET_LIB_PATH=$V HERMNET_ALLOW_STALE_LIB=1 python tools/kbench.py 40 2>/dev/null | grep message_
done
