# GPU box (round 4): generator.RASTER_ON_SIDE_STREAM forced on / off for every batch size, on the full bench line (profiles/r04_raster_side_stream_ab.txt).
cd /root/repo
mkdir -p gpurun_out
python -m pytest tests/test_generator_gpu.py tests/test_path_kernels_gpu.py -q -m gpu 2>&1 | tail -3
for rep in 1 2; do
for on in True False; do
python - > gpurun_out/r4_ab3_$on.json 2>gpurun_out/r4_ab3.err <<PY
import sys
from next3d_amd import generator
generator.RASTER_ON_SIDE_STREAM = 1 << 30 if $on else 0
sys.argv = ['bench.py', '--no-cpu-baseline', '--no-roofline', '--steps', '30']
import bench
bench.main()
PY
python - <<PY
import json
d=json.loads(open('gpurun_out/r4_ab3_$on.json').read().strip().splitlines()[-1])
print('raster on side stream: $on  value',round(d['value'],1),'single',round((d.get('single_stream') or {}).get('value',0),1), 'config1b', {k:round(v,2) for k,v in d.get('config1b',{}).items() if isinstance(v,float)})
PY
done; done
