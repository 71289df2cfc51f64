cd /root/repo
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/r4_tests6.txt 2>&1; tail -4 gpurun_out/r4_tests6.txt
N3D_LIB=tools/probe/libn3d_tuning.so timeout 300 python tools/f16_bench.py > gpurun_out/r4_f16_bench2.txt 2>&1; grep -E "^---|stride-1|torgb|transposed" gpurun_out/r4_f16_bench2.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 600 python bench.py > gpurun_out/r4_bench_rgbs.json 2> gpurun_out/r4_bench_rgbs.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4_bench_rgbs.json').read().strip().splitlines()[-1])
print('value',d['value'],'single',d['single_stream']['value'],'frac',d['roofline']['frac'], d['roofline']['family_ms_per_step'])
print('sr_fp16',d['sr_fp16_mode']['value'],d['sr_fp16_mode']['roofline_f16']['frac'],d['sr_fp16_mode']['roofline_f16'].get('frac_vs_measured_ceiling'))
print('fp16bb',d['fp16_backbones_mode']['value'], d['fp16_backbones_mode']['roofline_f16']['frac'])
print('config1b',d['config1b']); print('b1',d['b1_route']['force_fp32']['value'], d['b1_route']['default_fp16_sr']['value'])
PY
