#!/bin/bash
mkdir -p gpurun_out
for rep in 1 2; do
for k in 64 8 4 2 1; do
  N3D_KSPLIT_MAX=$k timeout 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 > gpurun_out/ab_tmp.json
  python -c "import json; d=json.load(open('gpurun_out/ab_tmp.json')); print('ksplit_max=$k', round(d['value'],1), round(d['ms_per_step'],3))"
done
done
