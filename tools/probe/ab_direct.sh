#!/bin/bash
# FIR kernel A/B: isolated launches (tools/fir_bench.py, both row counts of the c8 kernel), the FIR tests, bench with the NCHW separable kernel on / off
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_ops_gpu.py -q -m gpu -k "fir4 or stride2" 2>&1 | tail -4
timeout 120 python tools/fir_bench.py 2>&1 | tee gpurun_out/fir_bench.txt
N3D_FIR_SEP_RPT=8 timeout 120 python tools/fir_bench.py 2>&1 | head -8 | tee -a gpurun_out/fir_bench.txt
for v in 1 0 1 0; do
  N3D_FIR_SEP_NCHW=$v timeout 200 python bench.py --steps 30 --warmup 5 2>/dev/null | tail -1 > gpurun_out/ab_fir_$v.json
  python -c "import json; d=json.load(open('gpurun_out/ab_fir_$v.json')); print('fir_sep_nchw=$v', d['value'], d['ms_per_step'], d.get('sr_fp16_mode',{}).get('value'))"
done
