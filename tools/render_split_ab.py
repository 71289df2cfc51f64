#!/usr/bin/env python3
"""A/B of the renderer's decoder arithmetic: float32-input MFMAs (v_mfma_f32_32x32x2_f32, 67 per 32 samples) vs the convolutions' split-bf16
(v_mfma_f32_32x32x16_bf16, 24 per 32 samples; n3d_render_opts.decoder_split_bf16).  Time per launch and the difference of the outputs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from next3d_amd import demo, generator
dev = torch.device('cuda', 0)
G, _ = demo.build_generator(dev)
for (B, R, S) in [(4, 64, 48), (1, 64, 48), (4, 128, 48), (4, 64, 96)]:
    G.rendering_kwargs['depth_resolution'] = G.rendering_kwargs['depth_resolution_importance'] = S
    z, c, c_cond, v = demo.demo_batch(list(range(B)), device=dev)
    ws = G.mapping(z, c_cond, truncation_psi=0.7, truncation_cutoff=14)
    planes, _ = G._planes(ws.to(dev), v, 'const', False, False)
    jit = torch.rand(B, R * R, S, 1, device=dev); u = torch.rand(B * R * R, S, device=dev)
    res = {}
    for rep in range(3):
        for mode in (False, True):
            generator.RENDER_DECODER_SPLIT = mode
            for _ in range(3):
                out = G.render(planes, c, R, depth_jitter=jit, importance_u=u)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                out = G.render(planes, c, R, depth_jitter=jit, importance_u=u)
            e1.record(); torch.cuda.synchronize()
            res.setdefault(mode, []).append((e0.elapsed_time(e1) / 20 * 1e3, [o.clone() for o in out]))
    a, b = min(t for t, _ in res[False]), min(t for t, _ in res[True])
    fa, fb = res[False][0][1], res[True][0][1]
    d = [(x - y).abs() for x, y in zip(fa, fb)]
    big = int((d[0].amax(1) > 1e-3).sum())
    print(f'batch {B}, {R}x{R} rays, {S}+{S}: float32 MFMA {a:7.1f} us   split-bf16 {b:7.1f} us  ({a / b:.3f}x)   features max-abs diff {float(d[0].max()):.2e} '
          f'(mean {float(d[0].mean()):.2e}; pixels beyond 1e-3: {big} of {d[0][:, 0].numel()}), depth {float(d[1].max()):.2e}', flush=True)
