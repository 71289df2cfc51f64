#!/usr/bin/env python3
"""Same-box A/B of one module switch of next3d_amd.layers (GPU box):

    python tools/ab_switch.py SK_S2 True False [--reps 2]        (a switch of another module: cg.SK_SEAM, networks.X)

Per value and repetition (alternating, one process): the headline loop of bench.py's shape — batch 4, 512² / 64² / 48 + 48, force_fp32, steps round-robin on three HIP
streams, 30 timed steps — the same steps on ONE stream, and the scripts' call pattern (batch 1, default route, one eager G.synthesis per frame, 60 frames).  The switch is
flipped in-process (module constants, no environment variables); prepared weights do not depend on it."""
import argparse
import ast
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('name')
    ap.add_argument('values', nargs='+')
    ap.add_argument('--reps', type=int, default=2)
    ap.add_argument('--steps', type=int, default=30)
    a = ap.parse_args()
    from next3d_amd import _lib, demo, layers, networks
    from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
    mod = layers
    if '.' in a.name:
        mod = {'layers': layers, 'cg': cg, 'networks': networks}[a.name.split('.')[0]]
        a.name = a.name.split('.')[1]
    vals = [ast.literal_eval(v) for v in a.values]
    dev = torch.device('cuda', 0)
    G, _ = demo.build_generator(dev)
    g = torch.Generator(device=dev).manual_seed(1234)
    R, Sc, Sf = 64, 48, 48

    def inputs(B):
        z, c, c_cond, v = demo.demo_batch(list(range(B)), device=dev)
        return z, c, c_cond, v, torch.rand((B, R * R, Sc, 1), device=dev, generator=g), torch.rand((B * R * R, Sf), device=dev, generator=g)
    in4, in1 = inputs(4), inputs(1)
    lanes = [torch.cuda.Stream(device=dev) for _ in range(3)]

    def step(inp, fp32, stream=None):
        z, c, c_cond, v, jit, u = inp
        with torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream()):
            ws = G.mapping(z, c_cond, truncation_psi=0.7, truncation_cutoff=14)
            kw = dict(force_fp32=True) if fp32 else {}
            img = G.synthesis(ws, c, v, neural_rendering_resolution=R, noise_mode='const', depth_jitter=jit, importance_u=u, **kw)['image']
            out = torch.empty(img.shape, dtype=torch.uint8, device=dev)
            _lib.check(_lib.lib().n3d_to_uint8(_lib.ptr(img), _lib.ptr(out), img.numel(), _lib.stream()))
        return out

    def timed(fn, n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n):
            fn(k)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    # clock ramp
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 2.0:
        step(in4, True); torch.cuda.synchronize()
    for rep in range(a.reps):
        for val in vals:
            setattr(mod, a.name, val)
            for k in range(4):
                step(in4, True, lanes[k % 3]); step(in1, False)
            torch.cuda.synchronize()
            t3 = timed(lambda k: step(in4, True, lanes[k % 3]), a.steps)
            t1 = timed(lambda k: step(in4, True), a.steps)
            tb = timed(lambda k: step(in1, False), 60)
            print(f'{mod.__name__.split(".")[-1]}.{a.name} = {val!r:6}: batch 4 three lanes {a.steps * 4 / t3:7.1f} frames/s ({1e3 * t3 / a.steps:.3f} ms/step), one stream {a.steps * 4 / t1:7.1f} frames/s '
                  f'({1e3 * t1 / a.steps:.3f} ms/step); batch 1 default route {1e3 * tb / 60:.3f} ms/frame', flush=True)


if __name__ == '__main__':
    main()
