#!/bin/bash
# A/B of N3D_DIRECT_SPLIT8 (1x1 layers writing split8 for their sole 3x3 consumer): tests + two bench runs each
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "conv1x1" 2>&1 | tail -5
timeout 500 python -m pytest tests/test_generator_gpu.py -q -m gpu -x 2>&1 | tail -5
for v in 1 0 1 0; do
  N3D_DIRECT_SPLIT8=$v timeout 200 python bench.py --steps 30 --warmup 5 2>/dev/null | tail -1 > gpurun_out/ab_direct_$v.json
  python -c "import json; d=json.load(open('gpurun_out/ab_direct_$v.json')); print('direct=$v', d['value'], d['ms_per_step'], d.get('sr_fp16_mode',{}).get('value'))"
done
