set -e
export TMPDIR=/tmp
O=$PWD/gpurun_out/prof_shard1
rm -rf $O; mkdir -p $O
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29555 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --shard-anyway --config c2 --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > $O/stats.log 2>&1
cp $(ls $O/stats/*/*_kernel_stats.csv | head -1) $O/kernel_stats.csv
