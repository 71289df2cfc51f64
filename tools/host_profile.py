#!/usr/bin/env python3
"""cProfile of the HOST side of the scripts' batch-1 call (one G.mapping + G.synthesis per frame, eager): where the ~4 ms of enqueue time per frame go.
    python tools/host_profile.py [--fp32]"""
import cProfile, os, pstats, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from next3d_amd import demo
dev = torch.device('cuda', 0)
G, _ = demo.build_generator(dev)
z, c, c_cond, v = demo.demo_batch([0], device=dev)
R, Sc, Sf = 64, 48, 48
g = torch.Generator(device=dev).manual_seed(1)
jit, u = torch.rand((1, R * R, Sc, 1), device=dev, generator=g), torch.rand((R * R, Sf), device=dev, generator=g)
kw = dict(force_fp32=True) if '--fp32' in sys.argv else {}


def frame():
    ws = G.mapping(z, c_cond, truncation_psi=0.7, truncation_cutoff=14)
    return G.synthesis(ws, c, v, neural_rendering_resolution=R, noise_mode='const', depth_jitter=jit, importance_u=u, **kw)['image']


for _ in range(10):
    frame()
torch.cuda.synchronize()
host = []
for _ in range(20):
    torch.cuda.synchronize(); t0 = time.perf_counter(); frame(); host.append(time.perf_counter() - t0)
print(f'host enqueue per frame: min {min(host) * 1e3:.3f} ms, median {sorted(host)[10] * 1e3:.3f} ms')
pr = cProfile.Profile()
for _ in range(30):
    torch.cuda.synchronize(); pr.enable(); frame(); pr.disable()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(45)
