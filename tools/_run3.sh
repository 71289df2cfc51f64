cd /root/repo
mkdir -p gpurun_out
python -m pytest tests/test_f16_gpu.py tests/test_ops_gpu.py -x -q -m gpu -s -k "fused_torgb" > gpurun_out/r4_rgb16_tests.txt 2>&1; grep -E "fused|passed|failed|Error" gpurun_out/r4_rgb16_tests.txt | tail -12
python -m pytest tests/test_generator_gpu.py tests/test_networks_gpu.py -x -q -m gpu -s -k "fused_last or fp16 or default_route or superres" > gpurun_out/r4_rgb16_tests2.txt 2>&1; grep -E "fused vs|float16 route|passed|failed|Error" gpurun_out/r4_rgb16_tests2.txt | tail -8
timeout 600 python bench.py > gpurun_out/r4_bench_rgb16.json 2> gpurun_out/r4_bench_rgb16.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_bench_rgb16.json').read().strip().splitlines()[-1])
print('value',d['value'],'single',d['single_stream']['value'],'frac',d['roofline']['frac'], d['roofline']['family_ms_per_step'])
print('sr_fp16',d['sr_fp16_mode']['value'],d['sr_fp16_mode']['roofline_f16']['frac'],d['sr_fp16_mode']['roofline_f16']['avg_launch_ms'], d['sr_fp16_mode']['family_ms_per_step'])
print('fp16bb',d['fp16_backbones_mode']['value'], d['fp16_backbones_mode']['roofline_f16']['frac'])
PY
