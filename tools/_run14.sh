cd /root/repo
mkdir -p gpurun_out
python -m pytest tests -q -m gpu > gpurun_out/r4_tests7.txt 2>&1; tail -4 gpurun_out/r4_tests7.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
python tools/layer_trace.py --batch 4 > gpurun_out/r04_layer_trace.txt 2>/dev/null; tail -1 gpurun_out/r04_layer_trace.txt
python tools/layer_trace.py --batch 1 > gpurun_out/r04_layer_trace_batch1.txt 2>/dev/null; tail -1 gpurun_out/r04_layer_trace_batch1.txt
grep "mode2" gpurun_out/r04_layer_trace.txt | grep "split8->nchw" | head -5
python bench.py --no-extras --no-cpu-baseline --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('value',d['value'],'frac',d['roofline']['frac'],'single',d['single_stream'])"
