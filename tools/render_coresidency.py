#!/usr/bin/env python3
"""render_rays_kernel as the VICTIM of the rasteriser's co-residency finding.

render_rays_kernel gathers its texels through the vector L1 with plain loads — the access kind that returned wrong words in the rasteriser
while 8-wave split-bf16 MFMA workgroups of another stream shared the CUs.  Here the renderer is launched K times on one stream while the
REAL pre-split 3x3 convolution kernel (64 x 64 x 512 -> 512 channels at batch 4: the co-runner that corrupted the plain-load rasteriser in
36 of 36 runs) is kept queued on a second stream, and every output word is compared with a solo render on the device (no host sync in the
loop).  Also counted: how many renders really overlapped a convolution (event timestamps), so that "0 mismatches" cannot mean "never co-resident".

    python tools/render_coresidency.py [--launches 20000] [--batch 4]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--launches', type=int, default=20000)
    ap.add_argument('--batch', type=int, default=4)
    ap.add_argument('--convs-per-render', type=int, default=3, help='co-runner launches queued per render (each ~180 us; a render ~530 us at batch 4)')
    a = ap.parse_args()
    from next3d_amd import _lib, demo
    from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
    dev = torch.device('cuda', 0)
    G, _ = demo.build_generator(dev)
    B = a.batch
    z, c, c_cond, v = demo.demo_batch(list(range(B)), device=dev)
    ws = G.mapping(z, c_cond, truncation_psi=0.7, truncation_cutoff=14)
    planes, _ = G._planes(ws, v, 'const', False, False)
    jit = torch.rand(B, 64 * 64, 48, 1, device=dev)
    u = torch.rand(B * 64 * 64, 48, device=dev)
    render = lambda: G.render(planes, c, 64, depth_jitter=jit, importance_u=u)
    ref_f, ref_d = render()
    ref_f, ref_d = ref_f.clone(), ref_d.clone()
    # the co-runner: the pre-split stride-1 kernel on a 4 x 512 x 64 x 64 split8 input (256 eight-wave workgroups, 150 KB of LDS each)
    x = cg.split8_from_nchw(torch.randn(4, 512, 64, 64, device=dev))
    wt = cg.prep_weight_bf16x3(torch.randn(512, 512, 3, 3, device=dev) * 0.02)
    y = torch.empty(4, 512, 64, 64, device=dev)
    conv = lambda: cg.conv_launch(x, wt, 3, 0, 512, out=y, bf16x3=True)
    conv(); torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    bad = torch.zeros(2, dtype=torch.int64, device=dev)
    words = torch.zeros(2, dtype=torch.int64, device=dev)
    # phase 0: solo (nothing on the second stream): the renderer is reproducible by itself
    for phase, corun in (('solo', False), ('beside the convolution kernel', True)):
        bad.zero_(); words.zero_()
        n = a.launches if corun else max(200, a.launches // 20)
        ev = []
        for k in range(n):
            if corun:
                with torch.cuda.stream(sb):
                    for _ in range(a.convs_per_render):
                        conv()
            with torch.cuda.stream(sa):
                probe = k % 500 == 0
                if probe:
                    e0 = torch.cuda.Event(enable_timing=True); e0.record()
                f, d = render()
                if probe:
                    e1 = torch.cuda.Event(enable_timing=True); e1.record(); ev.append((e0, e1))
                nf, nd = (f != ref_f).sum(), (d != ref_d).sum()
                bad[0] += (nf > 0).long(); bad[1] += (nd > 0).long()
                words[0] += nf; words[1] += nd
            if k % 64 == 63:                   # keep the queues bounded (no device-side effect on co-residency: both streams stay busy)
                sa.synchronize() if not corun else None
                if corun and k % 1024 == 1023:
                    torch.cuda.synchronize()
        torch.cuda.synchronize()
        t = [e0.elapsed_time(e1) * 1e3 for e0, e1 in ev]
        print(f'{phase}: {n} render launches (batch {B}, 64 x 64 rays, 48 + 48 samples), launches with a mismatching feature / depth word: {bad.tolist()}, '
              f'mismatching words: {words.tolist()}; render time (sampled): {min(t):.0f} - {max(t):.0f} us, median {sorted(t)[len(t) // 2]:.0f} us')
        if int(bad.sum()):
            sys.exit(1)


if __name__ == '__main__':
    main()
