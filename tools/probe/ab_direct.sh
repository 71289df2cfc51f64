#!/bin/bash
# A/B of the layout-handover fusions (N3D_TORGB_SIDE: toRGB writes split8 for the next block; N3D_DIRECT_SPLIT8: fromrgb writes
# split8 for conv1): their tests + bench runs with the switch on / off
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "conv1x1" 2>&1 | tail -5
timeout 500 python -m pytest tests/test_generator_gpu.py -q -m gpu -x 2>&1 | tail -5
for v in 1 0 1 0; do
  N3D_TORGB_SIDE=$v timeout 200 python bench.py --steps 30 --warmup 5 2>/dev/null | tail -1 > gpurun_out/ab_side_$v.json
  python -c "import json; d=json.load(open('gpurun_out/ab_side_$v.json')); print('torgb_side=$v', d['value'], d['ms_per_step'], d.get('sr_fp16_mode',{}).get('value'))"
done
