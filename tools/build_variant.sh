#!/bin/bash
# Build a library variant with extra -D flags for the message kernels (A/B timing with tools/kbench.py):
#   bash tools/build_variant.sh b6 -DHN_BURST_PART=6   ->  hermnet_amd/csrc/variants/libhermnet_b6.so
# then:  HERMNET_LIB_PATH=hermnet_amd/csrc/variants/libhermnet_b6.so python tools/kbench.py
set -e
NAME=$1; shift
cd "$(dirname "$0")/../hermnet_amd/csrc"
make -s -j4 >/dev/null
mkdir -p variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -O2 "$@" -c message_kernels.hip -o variants/message_kernels_$NAME.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-slp-vectorize "$@" -c message_bwd_cl.hip -o variants/message_bwd_cl_$NAME.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 variants/message_kernels_$NAME.o variants/message_bwd_cl_$NAME.o geometry_kernels.o node_kernels.o node_chain.o node_chain_wide.o node_chain16.o relation_kernels.o neighbor_kernels.o train_kernels.o train_node_kernels.o stream_kernels.o host_api.o -o variants/libhermnet_$NAME.so
echo built variants/libhermnet_$NAME.so
