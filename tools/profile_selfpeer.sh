# kernel trace of the self-peer sharded step (one rank, halo rows exchanged with itself over RCCL) and of the unsharded step
# on the same cell: tools/profile_selfpeer.sh <tag> [reps]      (summaries under gpurun_out/<tag>_*)
set -e
export TMPDIR=/tmp
TAG=${1:-r06_selfpeer}
REPS=${2:-10,10,31}
O=$PWD/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/sp -- python3 bench.py --self-peer 1 --config c2 --reps $REPS --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/sp.json 2> $O/sp.err
cp $(ls $O/sp/*/*_kernel_stats.csv | head -1) $O/selfpeer_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/un -- python3 bench.py --config c2 --reps $REPS --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > $O/un.json 2> $O/un.err
cp $(ls $O/un/*/*_kernel_stats.csv | head -1) $O/unsharded_kernel_stats.csv
rm -rf $O/sp $O/un
