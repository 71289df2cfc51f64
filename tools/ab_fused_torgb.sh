# GPU box (round 4): same-box A/B of layers.FUSED_TORGB — both super-resolution blocks / the last block only / off — on `bench.py --no-extras` (profiles/r04_fused_torgb_ab.txt).
cd /root/repo
mkdir -p gpurun_out
python -m pytest tests/test_ops_gpu.py -q -m gpu -k "fused_torgb" 2>&1 | tail -3
for rep in 1 2; do
for mode in both last off; do
python - > gpurun_out/r4_ab2_$mode.json 2>gpurun_out/r4_ab2.err <<PY
import sys
from next3d_amd import layers, networks
mode = '$mode'
if mode == 'off':
    layers.FUSED_TORGB = False
if mode == 'last':                      # fused toRGB only for the last block (no split8 side output)
    orig = networks._Block.__call__
    def call(self, x, img, bank, n, fir, noise_mode, x_out=None, x_split8=None, next_block=None, last=False):
        if not last:
            old = layers.FUSED_TORGB; layers.FUSED_TORGB = False
            try: return orig(self, x, img, bank, n, fir, noise_mode, x_out=x_out, x_split8=x_split8, next_block=next_block, last=last)
            finally: layers.FUSED_TORGB = old
        return orig(self, x, img, bank, n, fir, noise_mode, x_out=x_out, x_split8=x_split8, next_block=next_block, last=last)
    networks._Block.__call__ = call
sys.argv = ['bench.py', '--no-extras', '--no-cpu-baseline', '--steps', '30']
import bench
bench.main()
PY
python - <<PY
import json
d=json.loads(open('gpurun_out/r4_ab2_$mode.json').read().strip().splitlines()[-1])
f=d['roofline']['family_ms_per_step']
print('fused toRGB: $mode  value',round(d['value'],1),'ms',round(d['ms_per_step'],3),'frac',round(d['roofline']['frac'],4), 'conv3x3',f['conv2d_bf16x3'], 'conv1x1',f['conv1x1_bf16x3'], 'misc', f['misc'])
PY
done; done
