#!/usr/bin/env python3
"""Batch-1 A/B of one module switch (GPU box): the scripts' call pattern — one G.synthesis per frame at 512² / 64² / 48+48 — timed three ways per value:
eager wall time per frame (60 frames, one stream), HOST time per frame (a frame's launches enqueued on an idle device: time until Python returns, no synchronize)
and HIP-graph replay time per frame (no host work between launches: the device-side chain alone).
    python tools/batch1_ab.py cg.SK_SEAM False True [--fp32] [--reps 2]"""
import argparse, ast, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('name'); ap.add_argument('values', nargs='+')
    ap.add_argument('--reps', type=int, default=2); ap.add_argument('--fp32', action='store_true')
    a = ap.parse_args()
    from next3d_amd import _lib, demo, layers, networks
    from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
    mod = {'layers': layers, 'cg': cg, 'networks': networks}[a.name.split('.')[0]]
    name = a.name.split('.')[1]
    vals = [ast.literal_eval(v) for v in a.values]
    dev = torch.device('cuda', 0)
    G, _ = demo.build_generator(dev)
    z, c, c_cond, v = demo.demo_batch([0], device=dev)
    R, Sc, Sf = 64, 48, 48
    g = torch.Generator(device=dev).manual_seed(1)
    jit, u = torch.rand((1, R * R, Sc, 1), device=dev, generator=g), torch.rand((R * R, Sf), device=dev, generator=g)
    kw = dict(force_fp32=True) if a.fp32 else {}
    ws = G.mapping(z, c_cond, truncation_psi=0.7, truncation_cutoff=14)

    def frame():
        w = G.mapping(z, c_cond, truncation_psi=0.7, truncation_cutoff=14)
        return G.synthesis(w, c, v, neural_rendering_resolution=R, noise_mode='const', depth_jitter=jit, importance_u=u, **kw)['image']
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 2.0:
        frame(); torch.cuda.synchronize()
    for rep in range(a.reps):
        for val in vals:
            setattr(mod, name, val)
            G._graphs = None
            for _ in range(5):
                frame()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(60):
                frame()
            torch.cuda.synchronize()
            eager = (time.perf_counter() - t0) / 60 * 1e3
            host = []
            for _ in range(10):
                torch.cuda.synchronize()
                t0 = time.perf_counter(); frame(); host.append(time.perf_counter() - t0)
            torch.cuda.synchronize()
            gkw = dict(neural_rendering_resolution=R, noise_mode='const', depth_jitter=jit, importance_u=u, **kw)
            G.synthesis_graph(ws, c, v, **gkw); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(60):
                G.synthesis_graph(ws, c, v, **gkw)
            torch.cuda.synchronize()
            graph = (time.perf_counter() - t0) / 60 * 1e3
            print(f'{a.name} = {val!r:6}: eager {eager:.3f} ms/frame, host enqueue {min(host) * 1e3:.3f} ms/frame (median {sorted(host)[5] * 1e3:.3f}), graph replay {graph:.3f} ms/frame'
                  f' ({"force_fp32" if a.fp32 else "default route"})', flush=True)


if __name__ == '__main__':
    main()
