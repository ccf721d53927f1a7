#!/bin/bash
# Lab: a variant of librime_hip.so with extra -D switches on ONE source (default fringe_mfma.hip), for A/B runs through
# RIME_LIB_PATH=tools/bin/librime_<name>.so.   usage: tools/build_variant.sh <name> "<-D flags>" [source.hip]
# fringe_mfma.hip is built from the LAB source: the product file + tools/lab/fringe_mfma_lab.patch (tools/lab/README.md), which
# carries every RIME_LAB_* / RIME_ABL_* / RIME_PHASE_MAGIC / RIME_BUILD_FWD_V2 block and the RIME_BWD_PIPE kernel; the library
# build contains none of them.
set -e
name=$1; flags=$2; src=${3:-fringe_mfma.hip}
root=$(cd "$(dirname "$0")/.." && pwd)
make -s -C $root/bayeslim_amd/csrc -j4 > /dev/null
mkdir -p $root/tools/bin/obj_$name
obj=$root/tools/bin/obj_$name/${src%.hip}.o
srcpath=$root/bayeslim_amd/csrc/$src
if [ "$src" = fringe_mfma.hip ]; then srcpath=$($root/tools/lab/make_lab_source.sh); fi
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wall -Wno-unused-function $flags \
    -I$root/bayeslim_amd/csrc -c $srcpath -o $obj
others=$(ls $root/bayeslim_amd/lib/obj/*.o | grep -v -- "-hip-amdgcn" | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 $obj $others -ldl -o $root/tools/bin/librime_$name.so
echo built tools/bin/librime_$name.so
