#!/usr/bin/env python3
"""A/B of the transposed pre-split kernel's wave tile (tuning build, N3D_UP_WIDE flipped in-process): 0 = eight waves x 32 positions x 32 channels (shipped, two workgroups = 16 waves
per CU, 26 LDS fragment reads per 27 MFMAs), 1 = four waves x 64 positions x 32 channels (two workgroups = 8 waves per CU, 34 reads per 54 MFMAs, 128 accumulators) — VERDICT r5 item 1b.
    tools/build_variant.sh tune conv2d_ps_bf16x3.hip -DN3D_TUNING && N3D_LIB=tools/probe/libn3d_tune.so python tools/up_wide_ab.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
dev = torch.device('cuda', 0)


def t_us(fn, iters=40):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for (N, I, O, H) in [(4, 512, 512, 32), (4, 512, 256, 64), (4, 256, 128, 128), (4, 256, 128, 256), (4, 32, 256, 128)]:
    wt = cg.prep_weight_bf16x3(torch.randn(O, I, 3, 3, device=dev) / (3 * I ** 0.5))
    xs = cg.split8_from_nchw(torch.randn(N, I, H, H, device=dev))
    fn = lambda: cg.conv_launch(xs, wt, 3, 2, O, bf16x3=True, out_c8=True)
    res = {}
    for v in ('0', '1', '0', '1'):
        os.environ['N3D_UP_WIDE'] = v
        y = fn().data.clone()
        res.setdefault(v, []).append((t_us(fn), y))
    a, b = min(t for t, _ in res['0']), min(t for t, _ in res['1'])
    gf = 2.0 * N * O * I * 9 * H * H / 1e9
    print(f'transposed N{N} I{I} O{O} {H}x{H} ({gf:.0f} GF): 8 waves x 32 positions {a:7.1f} us ({gf / a * 1e3:4.0f} TF)   4 waves x 64 positions {b:7.1f} us ({gf / b * 1e3:4.0f} TF)  {100 * (a / b - 1):+5.1f} %'
          f'  bitwise equal: {torch.equal(res["0"][0][1], res["1"][0][1])}', flush=True)
os.environ['N3D_UP_WIDE'] = '0'
