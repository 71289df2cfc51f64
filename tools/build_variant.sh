#!/bin/bash
# Build container: tools/probe/libn3d_<name>.so = the in-tree objects with ONE source recompiled under extra flags, for A/B runs
# on one GPU box (N3D_LIB=tools/probe/libn3d_<name>.so python ...).  Usage: tools/build_variant.sh <name> <source.hip> <flags...>
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
python -m next3d_amd.build > /dev/null
mkdir -p tools/probe
obj=tools/probe/${src%.hip}_$name.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 "$@" -c next3d_amd/csrc/$src -o $obj
others=$(ls next3d_amd/build/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/probe/libn3d_$name.so $others $obj
rm -f $obj
echo tools/probe/libn3d_$name.so
