#!/bin/bash
# usage: bash tools/chain_variants.sh OUT name...   -- tools/chain_bench.py (10k atoms, H 128) on the default library and on each
# variant built by tools/build_chain_variant.sh; one process per library (the library is chosen at import)
OUT=$1; shift
mkdir -p "$(dirname "$OUT")"
echo "== default (update chain on 16-row tiles where the library picks them)" > $OUT
python tools/chain_bench.py 10000 128 3 200 2>&1 | grep -v amdgpu.ids >> $OUT
echo "== HN_OPTIONS=update_tile16=0 (32-row update tiles: the baseline of the variants below)" >> $OUT
HN_OPTIONS=update_tile16=0 python tools/chain_bench.py 10000 128 3 200 2>&1 | grep -v amdgpu.ids >> $OUT
for v in "$@"; do
  echo "== $v" >> $OUT
  case $v in u16*) T16=2;; *) T16=0;; esac
  HN_OPTIONS=update_tile16=$T16 HERMNET_LIB_PATH=hermnet_amd/csrc/variants/libhermnet_$v.so python tools/chain_bench.py 10000 128 3 200 2>&1 | grep -v amdgpu.ids >> $OUT
done
grep -v "^rows" $OUT
