cd /root/repo
mkdir -p gpurun_out
python -m pytest tests/test_f16_gpu.py -q -m gpu -s -k "fused_torgb" 2>&1 | grep -E "fused float16|passed|failed|^E" | head -12
N3D_LIB=tools/probe/libn3d_tuning.so timeout 300 python tools/f16_bench.py > gpurun_out/r4_f16_bench2.txt 2>&1; grep -E "^---|stride-1|torgb" gpurun_out/r4_f16_bench2.txt
for rep in 1 2; do
for on in True False; do
python - > gpurun_out/r4_ab_fused_$on.json 2>gpurun_out/r4_ab.err <<PY
import sys
from next3d_amd import layers
layers.FUSED_TORGB = $on
sys.argv = ['bench.py', '--no-extras', '--no-cpu-baseline', '--steps', '30']
import bench
bench.main()
PY
python - <<PY
import json
d=json.loads(open('gpurun_out/r4_ab_fused_$on.json').read().strip().splitlines()[-1])
print('FUSED_TORGB=$on value',round(d['value'],1),'single',round(d['single_stream']['value'],1),'frac',round(d['roofline']['frac'],4), d['roofline']['family_ms_per_step']['conv2d_bf16x3'], d['roofline']['family_ms_per_step']['conv1x1_bf16x3'])
PY
done; done
python tools/layer_trace.py --batch 1 > gpurun_out/r4_layer_trace_b1_new.txt 2>&1; tail -1 gpurun_out/r4_layer_trace_b1_new.txt
python tools/layer_trace.py --batch 4 > gpurun_out/r4_layer_trace_b4_new.txt 2>&1; tail -1 gpurun_out/r4_layer_trace_b4_new.txt
