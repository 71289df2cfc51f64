#!/bin/bash
# env-switch A/B on the whole bench (interleaved repetitions): transposed kernel shapes
mkdir -p gpurun_out
for rep in 1 2 3 4 5; do
  for cfg in "N3D_UP_PS_MT=1" "N3D_UP_PS_MT=3"; do
    env $cfg timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > gpurun_out/ab_tmp.json
    python -c "import json; d=json.load(open('gpurun_out/ab_tmp.json')); print('$cfg', round(d['value'],1), round(d['ms_per_step'],3))"
  done
done
for cfg in "N3D_UP_PS_MT=1" "N3D_UP_PS_MT=3"; do
  env $cfg timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras --sr-fp16 2>/dev/null | tail -1 > gpurun_out/ab_tmp.json
  python -c "import json; d=json.load(open('gpurun_out/ab_tmp.json')); print('sr-fp16 $cfg', round(d['value'],1), round(d['ms_per_step'],3))"
done
