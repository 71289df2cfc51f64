#!/bin/bash
# Build container: one small library per RASTER_VARIANT (csrc/raster.hip) for tools/dbg_race3.sh.
cd "$(dirname "$0")/.."
for v in 0 1 2 4 6 8 15 16 31 32 64 65 68 72 79 320; do      # 320 = 64 + 256: dumps + every table load a plain single-dword load      # 64 = variant 0 + the raw-word dumps (probe); 65 / 68 / 72 = + agent-scope clear / table loads / z-buffer loads; 79 = 15 + dump
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -ffp-contract=off -DRASTER_VARIANT=$v \
      -o tools/probe/libraster_v$v.so next3d_amd/csrc/raster.hip next3d_amd/csrc/runtime.hip &
done
wait
ls -la tools/probe/*.so
