#!/bin/bash
# Regenerates the per-round profile artefacts on the MI355X box (run through gpurun from the repo root):
#   bash tools/profile_round.sh r02_v1   -> gpurun_out/prof_r02_v1/{kernel_stats.csv,pmc_*.csv,traffic.json,counters.json,bench.json}
# Counter passes are separate runs with --pmc only (never combined with a trace domain).
set -e
TAG=${1:-rXX}
export TMPDIR=/tmp
O=$PWD/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
B="python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc/FETCH_SIZE -- $B > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc/WRITE_SIZE -- $B > $O/write.log 2>&1
# issue / stall anatomy of every kernel of the step (SQ block: 8 counters per pass)
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY \
  --output-format csv -d $O/sq/a -- $B > $O/sq_a.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM \
  --output-format csv -d $O/sq/b -- $B > $O/sq_b.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE \
  --output-format csv -d $O/sq/c -- $B > $O/sq_c.log 2>&1 || echo "mfma counter pass failed (names differ?): see $O/sq_c.log"
python tools/collect_traffic.py $O/pmc $O/traffic.json > /dev/null
python tools/collect_counters.py $O/sq $O/counters.json > $O/counters.txt
cp $O/traffic.json profiles/traffic.json      # bench.py reads roofline.traffic from here
cp $(ls $O/stats/*/*_kernel_stats.csv | head -1) $O/kernel_stats.csv
cp $(ls $O/pmc/FETCH_SIZE/*/*_counter_collection.csv | head -1) $O/pmc_FETCH_SIZE.csv
cp $(ls $O/pmc/WRITE_SIZE/*/*_counter_collection.csv | head -1) $O/pmc_WRITE_SIZE.csv
python bench.py > $O/bench.json 2> $O/bench.err
tail -c 400 $O/bench.json
