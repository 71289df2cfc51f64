#!/usr/bin/env python3
"""Renderer time per call on four shapes with fixed inputs and a checksum; run once per library (N3D_LIB=tools/probe/libn3d_<tag>.so, tools/build_variant.sh) alternating in ONE gpurun
call — the A/B form used for every renderer change of round 6 (boxes of the pool differ by more than the effects measured)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from next3d_amd import demo, generator
dev = torch.device('cuda', 0)
G, _ = demo.build_generator(dev)
for (B, R, S) in [(4, 64, 48), (1, 64, 48), (4, 128, 48), (4, 64, 24)]:
    G.rendering_kwargs['depth_resolution'] = G.rendering_kwargs['depth_resolution_importance'] = S
    z, c, c_cond, v = demo.demo_batch(list(range(B)), device=dev)
    ws = G.mapping(z, c_cond, truncation_psi=0.7, truncation_cutoff=14)
    planes, _ = G._planes(ws.to(dev), v, 'const', False, False)
    g = torch.Generator(device=dev).manual_seed(3)
    jit = torch.rand(B, R * R, S, 1, device=dev, generator=g); u = torch.rand(B * R * R, S, device=dev, generator=g)
    ts = []
    for rep in range(3):
        for _ in range(3):
            out = G.render(planes, c, R, depth_jitter=jit, importance_u=u)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            out = G.render(planes, c, R, depth_jitter=jit, importance_u=u)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f'{os.environ.get("N3D_LIB", "new"):30s} batch {B}, {R}x{R}, {S}+{S}: {min(ts):7.1f} us  checksum {float(out[0].double().sum()):.6f}', flush=True)
