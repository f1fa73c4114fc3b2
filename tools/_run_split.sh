for rep in 1 2; do
for v in default split1 split2; do
  echo "== $v"
  if [ $v = default ]; then python tools/chain_bench.py 10000 128 3 200 2>&1 | grep -v amdgpu.ids
  else HERMNET_LIB_PATH=hermnet_amd/csrc/variants/libhermnet_$v.so python tools/chain_bench.py 10000 128 3 200 2>&1 | grep -v amdgpu.ids; fi
done
done
