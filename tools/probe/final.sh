#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -6 > gpurun_out/final_gpu_tests.txt
cat gpurun_out/final_gpu_tests.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -4
timeout 900 bash tools/refresh_profiles.sh r03 2>&1 | tail -30
