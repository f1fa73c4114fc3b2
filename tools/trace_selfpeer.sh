# per-dispatch kernel trace of a few self-peer steps (which launch takes how long, in order): tools/trace_selfpeer.sh <tag> [reps]
set -e
export TMPDIR=/tmp
TAG=${1:-r06_sptrace}
REPS=${2:-10,10,31}
O=$PWD/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/sp -- python3 bench.py --self-peer 1 --config c2 --reps $REPS --steps 2 --warmup 2 --no-cpu-baseline --no-secondary > $O/sp.json 2> $O/sp.err
cp $(ls $O/sp/*/*_kernel_trace.csv | head -1) $O/selfpeer_kernel_trace.csv
rm -rf $O/sp
