#!/usr/bin/env python3
"""Soak test of the benchmark's issue pattern: K forwards of the same inputs issued round-robin on three HIP streams without host
synchronisation; every image must equal the first bit for bit.  Usage (GPU box): python tools/soak.py [--steps 300] [--fp16]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--lanes', type=int, default=3)
    ap.add_argument('--fp16', action='store_true', help="the reference's default float16 super-resolution mode")
    a = ap.parse_args()
    from next3d_amd import demo
    dev = torch.device('cuda', 0)
    G, _ = demo.build_generator(dev)
    z, c, c_cond, v = demo.demo_batch([0, 1, 2, 3], device=dev)
    jit = torch.rand(4, 64 * 64, 48, 1, device=dev)
    u = torch.rand(4 * 64 * 64, 48, device=dev)
    kw = dict(neural_rendering_resolution=64, noise_mode='const', depth_jitter=jit, importance_u=u, force_fp32=not a.fp16)

    def forward():
        ws = G.mapping(z, c_cond, truncation_psi=0.7, truncation_cutoff=14)
        return G.synthesis(ws, c, v, **kw)

    ref = forward()
    torch.cuda.synchronize()
    lanes = [torch.cuda.Stream() for _ in range(a.lanes)]
    # one counter tensor PER LANE: increments issued from different streams on one tensor would race (a lost update could hide a mismatch)
    bad = [torch.zeros(3, dtype=torch.int64, device=dev) for _ in lanes]
    for s in lanes:
        s.wait_stream(torch.cuda.current_stream())
    for k in range(a.steps):
        lane = k % a.lanes
        with torch.cuda.stream(lanes[lane]):
            out = forward()
            for i, name in enumerate(('image', 'image_raw', 'image_depth')):
                bad[lane][i] += (out[name] != ref[name]).any().long()      # (accumulated on the device, in stream order of this lane: no host sync in the loop)
    torch.cuda.synchronize()
    total = torch.stack(bad).sum(0)
    print(f'{a.steps} pipelined forwards on {a.lanes} streams ({"fp16 SR" if a.fp16 else "fp32"} mode): mismatching image / image_raw / image_depth:', total.tolist())
    sys.exit(1 if int(total.sum()) else 0)


if __name__ == '__main__':
    main()
