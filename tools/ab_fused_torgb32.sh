# GPU box (round 5): same-box A/B of layers.FUSED_TORGB_MAX — 32 (the backbones' 32-colour toRGB layers fused into conv1's epilogue on the matrix cores)
# against 4 (round 4: only the super-resolution's 3-colour layers fuse) — on `bench.py --no-extras` (profiles/r05_fused_torgb32_ab.txt).
cd /root/repo
mkdir -p gpurun_out
for rep in 1 2; do
for mode in 32 4; do
python - > gpurun_out/r5_ab_rgb$mode.json 2>gpurun_out/r5_ab_rgb.err <<PY
import sys
from next3d_amd import layers
layers.FUSED_TORGB_MAX = $mode
sys.argv = ['bench.py', '--no-extras', '--no-cpu-baseline', '--steps', '30']
import bench
bench.main()
PY
python - <<PY
import json
d=json.loads(open('gpurun_out/r5_ab_rgb$mode.json').read().strip().splitlines()[-1])
f=d['roofline']['family_ms_per_step']
print('FUSED_TORGB_MAX $mode  frames/s',round(d['value'],1),'ms/step',round(d['ms_per_step'],3),'single_stream',round((d.get('single_stream') or {}).get('value',0),1),'frac',round(d['roofline']['frac'],4), 'conv3x3',f['conv2d_bf16x3'], 'conv1x1',f['conv1x1_bf16x3'], 'fir', f.get('upfirdn2d'), 'misc', f['misc'])
PY
done; done
