#!/usr/bin/env python3
"""Host enqueue time vs GPU time of one step (is the Python launch path the bottleneck?)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from next3d_amd import _lib, demo
dev = torch.device('cuda')
G, _ = demo.build_generator(dev)
z, c, c_cond, v = demo.demo_batch([0, 1, 2, 3], device=dev)
jit = torch.rand((4, 4096, 48, 1), device=dev); u = torch.rand((4 * 4096, 48), device=dev)
def step():
    ws = G.mapping(z, c_cond, truncation_psi=0.7, truncation_cutoff=14)
    return G.synthesis(ws, c, v, neural_rendering_resolution=64, noise_mode='const', depth_jitter=jit, importance_u=u, force_fp32=True)['image']
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print(f'enqueue {t_enq / 10 * 1e3:.2f} ms/step, total {t_all / 10 * 1e3:.2f} ms/step')
# graph capture
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
try:
    with torch.cuda.graph(g):
        out = step()
    torch.cuda.synchronize()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
    print(f'graph replay {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms/step')
    ref = step()
    torch.cuda.synchronize()
    g.replay(); torch.cuda.synchronize()
    print('graph vs eager max diff', float((out - ref).abs().max()))
except Exception as e:
    print('graph capture failed:', repr(e)[:500])
