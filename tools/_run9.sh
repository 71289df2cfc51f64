cd /root/repo
mkdir -p gpurun_out
for rep in 1 2; do
for cfg in "1 3" "0 3" "0 4" "0 5" "1 2"; do
set -- $cfg
N3D_OVERLAP_STATIC=$1 python bench.py --no-extras --no-cpu-baseline --no-roofline --steps 30 --lanes $2 > gpurun_out/r4_ab4.json 2>gpurun_out/r4_ab4.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r4_ab4.json').read().strip().splitlines()[-1])
print('overlap_static=$1 lanes=$2  value',round(d['value'],1))
PY
done; done
