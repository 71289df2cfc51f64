cd /root/repo
mkdir -p gpurun_out
for st in 0 1 0 1; do
N3D_LIB=tools/probe/libn3d_tuning.so N3D_PS_STAGED=$st python tools/layer_trace.py --batch 4 2>/dev/null | grep "split8->nchw" | grep "mode0" > gpurun_out/r4_trace_st$st.txt
python - <<PY
import re,collections
rows=[l.split() for l in open('gpurun_out/r4_trace_st$st.txt')]
agg=collections.OrderedDict()
for r in rows:
    key=' '.join(r[7:13])
    agg.setdefault(key,[]).append(float(r[0]))
print('staged=$st', ' | '.join(f"{k}: n{len(v)} avg {sum(v)/len(v):.1f}" for k,v in agg.items()), ' total', round(sum(sum(v) for v in agg.values()),1))
PY
done
