#!/bin/bash
# Diagnostic / experiment builds of the node chain kernels:
#   bash tools/build_chain_variant.sh rs8 -DHN_RS_SMALL=8   ->  hermnet_amd/csrc/variants/libhermnet_rs8.so
# then:  HERMNET_LIB_PATH=hermnet_amd/csrc/variants/libhermnet_rs8.so python tools/chain_bench.py
set -e
cd "$(dirname "$0")/../hermnet_amd/csrc"
NAME=$1; shift
mkdir -p variants
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function"
/opt/rocm/bin/hipcc $F "$@" -c node_chain.hip -o variants/node_chain_$NAME.o &
/opt/rocm/bin/hipcc $F "$@" -c node_chain_wide.hip -o variants/node_chain_wide_$NAME.o &
/opt/rocm/bin/hipcc $F "$@" -c node_chain16.hip -o variants/node_chain16_$NAME.o &
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 message_kernels.o message_bwd_cl.o geometry_kernels.o node_kernels.o \
  variants/node_chain_$NAME.o variants/node_chain_wide_$NAME.o variants/node_chain16_$NAME.o relation_kernels.o neighbor_kernels.o train_kernels.o train_node_kernels.o stream_kernels.o host_api.o \
  -o variants/libhermnet_$NAME.so
rm -f variants/node_chain_$NAME.o variants/node_chain_wide_$NAME.o variants/node_chain16_$NAME.o
echo built variants/libhermnet_$NAME.so
