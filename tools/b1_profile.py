#!/usr/bin/env python3
"""The operator-boundary route (oracle/b1_route.py: reference code pattern on next3d_amd.torch_utils.ops + shims) alone, batch 4 at the
benchmark's shape: wall time per step, the n3d kernel families' event time, and — under `rocprofv3 --kernel-trace --stats` — every
kernel of the step, torch's included.  Usage (GPU box): python tools/b1_profile.py [--steps 6] [--fp16]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--fp16', action='store_true')
    a = ap.parse_args()
    from next3d_amd import _lib, demo, mesh, spec
    from oracle import b1_route
    dev = torch.device('cuda', 0)
    d = demo.demo_arrays()
    P = spec.synthetic_state_dict(0)
    P.update(mesh.mesh_buffers(d['faces'], d['uvs'], d['uvfaces']))
    route = b1_route.Route(dev)
    P = route.to_device(P)
    mask = torch.nn.functional.interpolate(mesh.synthetic_uv_face_mask().float(), [256, 256]).to(dev)
    rk = dict(demo.RENDERING_KWARGS)
    B, R, Sc, Sf = a.batch, 64, 48, 48
    z, c, c_cond, v = demo.demo_batch(list(range(B)), device=dev)
    z = z.float()
    g = torch.Generator(device=dev).manual_seed(1)
    jitter = torch.rand((B, R * R, Sc, 1), device=dev, generator=g)
    u = torch.rand((B * R * R, Sf), device=dev, generator=g)

    def step(stages=None):
        with torch.no_grad():
            ws = route.mapping(P, z, c_cond, rk, truncation_psi=0.7, truncation_cutoff=14)
            return route.synthesis(P, ws, c, v, mask, rk, jitter, u, neural_rendering_resolution=R, force_fp32=not a.fp16)['image']
    step(); step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(f'B1 route: {1e3 * dt:.2f} ms per batch-{B} step = {B / dt:.1f} frames/s')
    _lib.prof_reset(); _lib.prof_enable(True)
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    _lib.prof_enable(False)
    prof = _lib.prof_read()
    tot = 0.0
    for k, p in prof.items():
        if p['launches']:
            print(f'  n3d {k:16s} {p["ms"] / a.steps:8.3f} ms/step  {p["launches"] / a.steps:6.1f} launches/step')
            tot += p['ms'] / a.steps
    print(f'  n3d kernels total {tot:.3f} ms/step (the rest: torch kernels of the reference code, host round trips, launch gaps)')
    # coarse stage timing (synchronising between stages)
    net, gen = route.networks, route.generator
    ws = route.mapping(P, z, c_cond, rk, truncation_psi=0.7, truncation_cutoff=14)

    def timeit(name, fn, n=3):
        fn(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            out = fn()
        torch.cuda.synchronize()
        print(f'  stage {name:28s} {1e3 * (time.perf_counter() - t) / n:8.3f} ms')
        return out
    with torch.no_grad(), torch.device(dev):
        tex = timeit('texture backbone', lambda: net.synthesis_network(P, 'texture_backbone.synthesis', ws[:, 14:]))
        timeit('static backbone', lambda: net.synthesis_network(P, 'backbone.synthesis', ws[:, :14]))
        vv, lms = v[:, :5023], v[:, 5023:]
        timeit('rasterize (4 views + fill)', lambda: gen.rasterize(P, vv, lms, tex, mask))
        crops = torch.randn(B, 32, 64, 64, device=dev)
        timeit('mouth StyleUNet', lambda: net.styleunet_synthesis(P, 'mouth_backbone.synthesis', crops, ws[:, :14], in_size=64, final_size=4, num_cond_res=64))
        st = torch.randn(B, 32, 256, 256, device=dev)
        timeit('blending StyleUNet', lambda: net.styleunet_synthesis(P, 'neural_blending.synthesis', st, ws[:, :14], in_size=256, final_size=32, num_cond_res=256))
        planes = torch.randn(B, 3, 32, 256, 256, device=dev)
        ro, rd = route.renderer.ray_sampler(c[:, :16].view(-1, 4, 4), c[:, 16:25].view(-1, 3, 3), R)
        feat = timeit('renderer (torch)', lambda: route.renderer.importance_renderer(P, 'decoder', planes, ro, rd, rk, jitter, u))[0]
        fi = feat.permute(0, 2, 1).reshape(B, 32, R, R).contiguous()
        timeit('superresolution', lambda: net.superresolution(P, 'superresolution', fi[:, :3], fi, ws[:, :14], force_fp32=not a.fp16))


if __name__ == '__main__':
    main()
