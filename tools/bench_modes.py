#!/usr/bin/env python3
"""Frames/s of the three call patterns of the inference scripts (batch 4, 512²/64²/48+48, one GPU):
  full     gen_samples_next3d.py        mapping + synthesis per frame (bench.py's headline)
  orbit    gen_videos_next3d.py         new camera per frame, planes cached (cache_backbone / use_cached_backbone)
  reenact  reenact_avatar_next3d.py     new mesh per frame, latent-only networks cached (cache_identity / use_cached_identity)
Prints one JSON object (not the driver's bench line).
  --mult 2      gen_videos_next3d.py's default sampling multiplier: 96 coarse + 96 importance samples
  --config1     BASELINE.json configs[0]: batch 1, 32² neural render, 24 + 24 samples (the reference's CPU-runnable case)"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from next3d_amd import _lib, demo

ap = argparse.ArgumentParser()
ap.add_argument('--mult', type=int, default=1)
ap.add_argument('--config1', action='store_true')
args = ap.parse_args()
dev = torch.device('cuda', 0)
G, _ = demo.build_generator(dev)
B, R, Sc, Sf = (1, 32, 24, 24) if args.config1 else (4, 64, 48 * args.mult, 48 * args.mult)
G.rendering_kwargs['depth_resolution'], G.rendering_kwargs['depth_resolution_importance'] = Sc, Sf
z, c, c_cond, v = demo.demo_batch(list(range(B)), device=dev)
g = torch.Generator(device=dev).manual_seed(1)
jitter = torch.rand((B, R * R, Sc, 1), device=dev, generator=g)
u = torch.rand((B * R * R, Sf), device=dev, generator=g)
kw = dict(neural_rendering_resolution=R, noise_mode='const', depth_jitter=jitter, importance_u=u, force_fp32=True)
ws = G.mapping(z, c_cond, truncation_psi=0.7, truncation_cutoff=14)
G.synthesis(ws, c, v, cache_backbone=True, cache_identity=True, **kw)


def run(fn, steps=20, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return steps * B / (time.perf_counter() - t0)


out = {
    'full': run(lambda: G.synthesis(G.mapping(z, c_cond, truncation_psi=0.7, truncation_cutoff=14), c, v, **kw)),
    'orbit_cached_planes': run(lambda: G.synthesis(ws, c, v, use_cached_backbone=True, **kw)),
    'reenact_cached_identity': run(lambda: G.synthesis(ws, c, v, use_cached_identity=True, **kw)),
}
print(json.dumps({'unit': 'frames/s', 'batch': B, 'render': R, 'samples': [Sc, Sf], **out}))
