#!/bin/bash
mkdir -p gpurun_out
{ timeout 300 python tools/soak.py --steps 600 2>&1 | tail -1
  timeout 300 python tools/soak.py --steps 600 --fp16 2>&1 | tail -1
  N3D_PAIR_BACKBONES=1 timeout 300 python tools/soak.py --steps 300 2>&1 | tail -1
  echo "(third run: N3D_PAIR_BACKBONES=1; final round-3 build: toRGB split8 side outputs, fromrgb split8 results, separable NCHW FIR)"; } | tee gpurun_out/soak.txt
