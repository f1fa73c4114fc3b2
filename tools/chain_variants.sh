#!/bin/bash
# usage: bash tools/chain_variants.sh OUT name...   -- tools/chain_bench.py (10k atoms, H 128) on the default library and on each
# variant built by tools/build_chain_variant.sh; one process per library (the library is chosen at import)
OUT=$1; shift
mkdir -p "$(dirname "$OUT")"
echo "== default" > $OUT
python tools/chain_bench.py 10000 128 3 200 >> $OUT 2>&1
for v in "$@"; do
  echo "== $v" >> $OUT
  HERMNET_LIB_PATH=hermnet_amd/csrc/variants/libhermnet_$v.so python tools/chain_bench.py 10000 128 3 200 >> $OUT 2>&1
done
grep -v "^rows" $OUT
