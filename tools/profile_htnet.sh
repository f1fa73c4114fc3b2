#!/bin/bash
# rocprofv3 kernel stats of the HTNet step (BASELINE configs[2]):  bash tools/profile_htnet.sh  -> gpurun_out/prof_htnet/
set -e
export TMPDIR=/tmp
O=$PWD/gpurun_out/prof_htnet
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 tools/htnet_bench.py 10 > $O/stats.log 2>&1
cp $(ls $O/stats/*/*_kernel_stats.csv | head -1) $O/kernel_stats.csv
python3 tools/htnet_bench.py 20 > $O/bench.json 2> $O/bench.err
