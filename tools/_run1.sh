cd /root/repo
mkdir -p gpurun_out
python -m pytest tests/test_f16_gpu.py -x -q -m gpu > gpurun_out/r4_f16_tests.txt 2>&1; tail -3 gpurun_out/r4_f16_tests.txt
N3D_LIB=tools/probe/libn3d_tuning.so timeout 300 python tools/f16_bench.py > gpurun_out/r4_f16_bench.txt 2>&1; tail -40 gpurun_out/r4_f16_bench.txt
timeout 600 python bench.py > gpurun_out/r4_bench_f16wide.json 2> gpurun_out/r4_bench_f16wide.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_bench_f16wide.json').read().strip().splitlines()[-1])
print('value',d['value'],'single',d['single_stream']['value'],'frac',d['roofline']['frac'])
print('sr_fp16',d['sr_fp16_mode']['value'],d['sr_fp16_mode']['roofline_f16']['frac'],d['sr_fp16_mode']['roofline_f16']['avg_launch_ms'])
print('fp16bb',json.dumps(d.get('fp16_backbones_mode'))[:1500])
PY
