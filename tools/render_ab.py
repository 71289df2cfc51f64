#!/usr/bin/env python3
"""A/B of the renderer's gather (tuning build: N3D_RENDER_GATHER=0 round 5's 64-bytes-per-lane gather, 1 = eight lanes per texel), same box, alternating; the outputs must be
bit-identical.  tools/build_variant.sh tune render.hip -DN3D_TUNING && N3D_LIB=tools/probe/libn3d_tune.so python tools/render_ab.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from next3d_amd import demo
dev = torch.device('cuda', 0)
G, _ = demo.build_generator(dev)
for (B, R, S) in [(4, 64, 48), (1, 64, 48), (4, 64, 96), (4, 128, 48)]:
    z, c, c_cond, v = demo.demo_batch(list(range(B)), device=dev)
    ws = G.mapping(z, c_cond, truncation_psi=0.7, truncation_cutoff=14)
    planes, _ = G._planes(ws.to(dev), v, 'const', False, False)
    jit = torch.rand(B, R * R, S, 1, device=dev); u = torch.rand(B * R * R, S, device=dev)
    G.rendering_kwargs = dict(G.rendering_kwargs, depth_resolution=S, depth_resolution_importance=S)
    res = {}
    for rep in range(2):
        for mode in ('0', '1'):
            os.environ['N3D_RENDER_GATHER'] = mode
            for _ in range(3):
                out = G.render(planes, c, R, depth_jitter=jit, importance_u=u)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                out = G.render(planes, c, R, depth_jitter=jit, importance_u=u)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(mode, []).append((e0.elapsed_time(e1) / 20 * 1e3, [o.clone() for o in out]))
    a, b = min(t for t, _ in res['0']), min(t for t, _ in res['1'])
    same = all(torch.equal(x, y) for x, y in zip(res['0'][0][1], res['1'][0][1]))
    print(f'batch {B}, {R}x{R} rays, {S}+{S} samples: 64 bytes per lane {a:7.1f} us   eight lanes per texel {b:7.1f} us   ({a / b:.2f}x)   outputs bit-identical: {same}', flush=True)
