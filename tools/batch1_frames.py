#!/usr/bin/env python3
"""K single-frame forwards (batch 1 at the metric's 512² / 64² / 48+48, one G.synthesis call per frame, eager, one stream) — the unchanged scripts' call pattern —
for `rocprofv3 --kernel-trace --stats -- python tools/batch1_frames.py` (then tools/rocpd_summary.py): per-kernel time of a frame.  Prints the wall time per frame."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from next3d_amd import demo          # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith('--')]
K = int(args[0]) if args else 40
DEFAULT_ROUTE = '--default-route' in sys.argv        # no force_fp32: the scripts' own call (float16 super-resolution blocks)
dev = torch.device('cuda', 0)
G, _ = demo.build_generator(dev)
z, c, c_cond, v = demo.demo_batch([0], device=dev)
R, Sc, Sf = 64, 48, 48
g = torch.Generator(device=dev).manual_seed(1)
jit, u = torch.rand((1, R * R, Sc, 1), device=dev, generator=g), torch.rand((R * R, Sf), device=dev, generator=g)


def frame():
    ws = G.mapping(z, c_cond, truncation_psi=0.7, truncation_cutoff=14)
    return G.synthesis(ws, c, v, neural_rendering_resolution=R, noise_mode='const', depth_jitter=jit, importance_u=u, **({} if DEFAULT_ROUTE else dict(force_fp32=True)))['image']


for _ in range(5):
    frame()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(K):
    frame()
torch.cuda.synchronize()
print(f'{K} frames, {(time.perf_counter() - t0) / K * 1e3:.3f} ms per frame (wall, eager, one stream, {"default route" if DEFAULT_ROUTE else "force_fp32"})')
