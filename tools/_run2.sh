cd /root/repo
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "fused_torgb or presplit_conv" > gpurun_out/r4_rgb_tests.txt 2>&1; tail -5 gpurun_out/r4_rgb_tests.txt
python -m pytest tests/test_f16_gpu.py tests/test_generator_gpu.py -x -q -m gpu -s -k "f16 or fused_last or benched_case or forward_matches or presplit_pipeline" > gpurun_out/r4_rgb_tests2.txt 2>&1; grep -E "fused vs|passed|failed|Error" gpurun_out/r4_rgb_tests2.txt | tail -8
./tools/probe/mfma_peak_f16.bin > gpurun_out/r4_mfma_peak_f16.txt 2>&1; grep random gpurun_out/r4_mfma_peak_f16.txt | head -30
timeout 600 python bench.py > gpurun_out/r4_bench_rgb.json 2> gpurun_out/r4_bench_rgb.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_bench_rgb.json').read().strip().splitlines()[-1])
print('value',d['value'],'single',d['single_stream']['value'],'frac',d['roofline']['frac'], d['roofline']['family_ms_per_step'])
print('sr_fp16',d['sr_fp16_mode']['value'],d['sr_fp16_mode']['roofline_f16']['frac'],d['sr_fp16_mode']['roofline_f16']['avg_launch_ms'])
print('fp16bb',d['fp16_backbones_mode']['value'], d['fp16_backbones_mode']['roofline_f16']['frac'])
PY
