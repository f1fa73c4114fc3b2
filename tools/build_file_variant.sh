#!/bin/bash
# One source file of the library rebuilt with extra -D flags, linked with the tree's other objects:
#   bash tools/build_file_variant.sh p_plain band_product.hip -DHN_P_NT=0  ->  hermnet_amd/csrc/variants/libhermnet_p_plain.so
# then:  HERMNET_LIB_PATH=hermnet_amd/csrc/variants/libhermnet_p_plain.so python tools/band_bench.py
set -e
NAME=$1; FILE=$2; shift 2
cd "$(dirname "$0")/../hermnet_amd/csrc"
make -s -j4 >/dev/null
mkdir -p variants
OBJ=variants/${FILE%.hip}_$NAME.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function "$@" -c $FILE -o $OBJ
OTHERS=$(ls *.o | grep -v "^${FILE%.hip}.o$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $OBJ $OTHERS -o variants/libhermnet_$NAME.so
echo built variants/libhermnet_$NAME.so
