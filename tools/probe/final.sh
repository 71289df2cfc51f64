#!/bin/bash
mkdir -p gpurun_out
for i in 1 2 3; do timeout 600 python -m pytest tests -q -m gpu 2>&1 | tail -1; done | tee gpurun_out/final_gpu_tests_x3.txt
timeout 300 python tools/soak.py --steps 3000 2>&1 | tail -1 | tee gpurun_out/soak_long.txt
timeout 300 python tools/soak.py --steps 3000 --fp16 2>&1 | tail -1 | tee -a gpurun_out/soak_long.txt
