#!/bin/bash
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "conv1x1" 2>&1 | tail -3
timeout 120 python tools/c1_bench.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/c1_bench.txt
N3D_C1_RES3=0 timeout 120 python tools/c1_bench.py 2>&1 | grep -v amdgpu.ids | head -4 | tee -a gpurun_out/c1_bench.txt
