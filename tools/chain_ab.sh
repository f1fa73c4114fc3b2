#!/bin/bash
# tools/chain_bench.py on library variants (tools/build_chain_variant.sh), one process each:  bash tools/chain_ab.sh name...
for v in "$@"; do
  echo "== $v"
  HERMNET_LIB_PATH=hermnet_amd/csrc/variants/libhermnet_$v.so python tools/chain_bench.py 10000 128 3 200 2>&1 | grep -v "amdgpu.ids\|^rows"
done
