cd /root/repo
mkdir -p gpurun_out
N3D_LIB=tools/probe/libn3d_tuning.so python tools/ps_staged_bench.py 2>/dev/null | tee gpurun_out/r4_ps_staged.txt
for rep in 1 2; do for st in 0 1; do
N3D_LIB=tools/probe/libn3d_tuning.so N3D_PS_STAGED=$st python bench.py --no-extras --no-cpu-baseline --steps 30 > gpurun_out/r4_ab5.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/r4_ab5.json').read().strip().splitlines()[-1])
print('staged stores=$st  value',round(d['value'],1),'frac',round(d['roofline']['frac'],4),'conv3x3',d['roofline']['family_ms_per_step']['conv2d_bf16x3'])
PY
done; done | tee -a gpurun_out/r4_ps_staged.txt
