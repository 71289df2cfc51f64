#!/usr/bin/env python3
"""A/B of the stride-2 pre-split kernel's workgroup (tuning build, N3D_S2_WIDE flipped in-process): 0 = 64 channels x 16 x 32 pixels, three 52 KB LDS buffers (shipped), 1 = 128 channels,
two 68 KB buffers, 128 accumulators per lane — VERDICT r5 item 1c.
    tools/build_variant.sh tune conv2d_ps_bf16x3.hip -DN3D_TUNING && N3D_LIB=tools/probe/libn3d_tune.so python tools/s2_wide_ab.py"""
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


# the generator's stride-2 layers on the pre-split kernel (input H x H after the FIR pad: 2 OH + 1)
for (N, I, O, H) in [(4, 128, 256, 257), (4, 256, 512, 129), (4, 512, 512, 65), (4, 512, 512, 33)]:
    wt = cg.prep_weight_bf16x3(torch.randn(O, I, 3, 3, device=dev) / (3 * I ** 0.5))
    xs = cg.split8_from_nchw(torch.randn(N, I, H, H, device=dev))
    fn = lambda: cg.conv_launch(xs, wt, 3, 1, O, bf16x3=True)
    res = {}
    for v in ('0', '1', '0', '1'):
        os.environ['N3D_S2_WIDE'] = v
        y = fn().clone()
        res.setdefault(v, []).append((t_us(fn), y))
    a, b = min(t for t, _ in res['0']), min(t for t, _ in res['1'])
    gf = 2.0 * N * O * I * 9 * ((H - 3) // 2 + 1) ** 2 / 1e9
    print(f'stride 2 N{N} I{I} O{O} {H}x{H} ({gf:.1f} GF): 64-channel workgroups {a:7.1f} us ({gf / a * 1e3:4.0f} TF)   128-channel workgroups {b:7.1f} us ({gf / b * 1e3:4.0f} TF)  {100 * (a / b - 1):+5.1f} %'
          f'  bitwise equal: {torch.equal(res["0"][0][1], res["1"][0][1])}', flush=True)
os.environ['N3D_S2_WIDE'] = '0'
