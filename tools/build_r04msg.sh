#!/bin/bash
# A/B library for VERDICT r5 item 2 (the message kernels got 3-5 % slower between round 4 and round 5): the CURRENT library with
# the message kernels' three source files as they stood at the end of round 4 (commit 974e587; the C interface of those
# files has not changed since) -> hermnet_amd/csrc/variants/libhermnet_r04msg.so
#   HERMNET_LIB_PATH=hermnet_amd/csrc/variants/libhermnet_r04msg.so HERMNET_ALLOW_STALE_LIB=1 python bench.py ...
set -e
REV=${1:-974e587}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
S=/tmp/hn_r04msg/a/b
rm -rf /tmp/hn_r04msg; mkdir -p $S /tmp/hn_r04msg/include
for f in message_kernels.hip message_bwd_cl.hip message_bwd_cl.h hermnet_math.h scan_i32.h; do
  git -C "$ROOT" show $REV:hermnet_amd/csrc/$f > $S/$f
done
cp "$ROOT/include/hermnet_hip.h" /tmp/hn_r04msg/include/
cd "$ROOT/hermnet_amd/csrc"
make -s -j4 >/dev/null
mkdir -p variants
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function"
/opt/rocm/bin/hipcc $F -O2 -c $S/message_kernels.hip -o variants/message_kernels_r04msg.o
/opt/rocm/bin/hipcc $F -fno-slp-vectorize -c $S/message_bwd_cl.hip -o variants/message_bwd_cl_r04msg.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 variants/message_kernels_r04msg.o variants/message_bwd_cl_r04msg.o geometry_kernels.o node_kernels.o node_chain.o node_chain_wide.o node_chain16.o relation_kernels.o neighbor_kernels.o train_kernels.o train_node_kernels.o stream_kernels.o host_api.o -o variants/libhermnet_r04msg.so
echo built variants/libhermnet_r04msg.so
