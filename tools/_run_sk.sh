cd /root/repo
python -m pytest tests/test_ops_gpu.py -q -m gpu -k "conv2d_up_bf16x3" 2>&1 | tail -3
python -m pytest tests/test_generator_gpu.py tests/test_networks_gpu.py -q -m gpu 2>&1 | tail -2
python tools/layer_trace.py --batch 4 2>/dev/null | grep "mode2" | grep "nchw  ->nchw\|split8->nchw"
python tools/layer_trace.py --batch 1 2>/dev/null | grep "mode2" | grep "nchw  ->nchw\|split8->nchw"
python bench.py --no-cpu-baseline --no-roofline --steps 30 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d['config1b']
print('value',round(d['value'],1),'single',round(d['single_stream']['value'],1),'b1 eager/graph',round(c['eager_ms_per_frame'],2),round(c['hip_graph_ms_per_frame'],2),'pipelined',round(c['pipelined_frames_per_s'],1),round(c['pipelined_hip_graph_frames_per_s'],1), 'config1', round(d['config1']['eager_ms_per_frame'],2), round(d['config1']['hip_graph_ms_per_frame'],2))"
