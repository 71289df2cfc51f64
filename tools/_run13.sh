cd /root/repo
mkdir -p gpurun_out
for lib in tools/probe/libn3d_oldstore.so next3d_amd/libn3d.so tools/probe/libn3d_oldstore.so next3d_amd/libn3d.so; do
N3D_LIB=$lib python tools/layer_trace.py --batch 4 2>/dev/null | grep "mode2" | grep "split8" > gpurun_out/r4_trace_up.txt
python - <<PY
import collections
rows=[l.split() for l in open('gpurun_out/r4_trace_up.txt')]
agg=collections.OrderedDict()
for r in rows:
    key=' '.join(r[6:13])
    agg.setdefault(key,[]).append(float(r[0]))
print('$lib'.split('/')[-1], ' | '.join(f"{k}: n{len(v)} {sum(v)/len(v):.1f}" for k,v in agg.items()), ' total', round(sum(sum(v) for v in agg.values()),1))
PY
done
for rep in 1 2; do for lib in tools/probe/libn3d_oldstore.so next3d_amd/libn3d.so; do
N3D_LIB=$lib python bench.py --no-extras --no-cpu-baseline --steps 30 > gpurun_out/r4_ab6.json 2>/dev/null
python - <<PY
import json
d=json.loads(open('gpurun_out/r4_ab6.json').read().strip().splitlines()[-1])
print('$lib'.split('/')[-1],' value',round(d['value'],1),'frac',round(d['roofline']['frac'],4),'conv3x3',d['roofline']['family_ms_per_step']['conv2d_bf16x3'])
PY
done; done
