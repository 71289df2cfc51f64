#!/bin/bash
# Build container: tools/probe/libn3d_tuning.so = EVERY source compiled with -DN3D_TUNING, i.e. with the ablation / variant switches
# the shipped library does not read (csrc/common.h n3d_tune: N3D_CONV_DBG, N3D_PS_NBUF, N3D_F16_NBUF, N3D_CONV_PERSIST, N3D_CONV_SK).
# Use on a GPU box as  N3D_LIB=tools/probe/libn3d_tuning.so python tools/conv_ps_abl.py ...
set -e
cd "$(dirname "$0")/.."
mkdir -p tools/probe/tuning_obj
objs=""
for src in next3d_amd/csrc/*.hip; do
  b=$(basename ${src%.hip})
  extra=""; [ "$b" = raster ] && extra="-ffp-contract=off"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DN3D_TUNING $extra "$@" -c $src -o tools/probe/tuning_obj/$b.o &
  objs="$objs tools/probe/tuning_obj/$b.o"
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/probe/libn3d_tuning.so $objs
rm -rf tools/probe/tuning_obj
echo tools/probe/libn3d_tuning.so
