#!/bin/bash
# Regenerates the per-round profile artefacts on the MI355X box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r01_v6      -> gpurun_out/prof_r01_v6/{kernel_stats.csv,pmc_*.csv,traffic.json,bench.json}
set -e
TAG=${1:-rXX}
export TMPDIR=/tmp
O=$PWD/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc/FETCH_SIZE -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc/WRITE_SIZE -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary > $O/write.log 2>&1
python tools/collect_traffic.py $O/pmc $O/traffic.json > /dev/null
cp $O/traffic.json profiles/traffic.json      # bench.py reads roofline.traffic from here
cp $(ls $O/stats/*/*_kernel_stats.csv | head -1) $O/kernel_stats.csv
cp $(ls $O/pmc/FETCH_SIZE/*/*_counter_collection.csv | head -1) $O/pmc_FETCH_SIZE.csv
cp $(ls $O/pmc/WRITE_SIZE/*/*_counter_collection.csv | head -1) $O/pmc_WRITE_SIZE.csv
python bench.py > $O/bench.json 2> $O/bench.err
tail -c 400 $O/bench.json
