# usage: bash tools/variants_run.sh name...   -- kbench each library variant built by tools/build_variant.sh
mkdir -p gpurun_out
for v in "$@"; do
  echo "== $v"
  L=hermnet_amd/csrc/variants/libhermnet_$v.so
  [ "$v" = base ] && L=hermnet_amd/csrc/libhermnet_hip.so
  HERMNET_LIB_PATH=$L timeout -k 10 120 python tools/kbench.py 20 2>&1 | grep "scatter_\|checksum" || exit 1
done
