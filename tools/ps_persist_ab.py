#!/usr/bin/env python3
"""A/B of the PERSISTENT form of the pre-split stride-1 / transposed kernels (tuning build: N3D_LIB=tools/probe/libn3d_tune.so, N3D_PS_PERSIST flipped
in-process): same results bit for bit, time per launch alone and with a second stream running the same layer (co-residency as in the three-lane bench).
    tools/build_variant.sh tune conv2d_ps_bf16x3.hip -DN3D_TUNING && N3D_LIB=tools/probe/libn3d_tune.so python tools/ps_persist_ab.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from next3d_amd import _lib
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


def ab(tag, fn, out, gf):
    """N3D_PS_PERSIST: 0 = one launch-time workgroup per tile (shipped), 1 = static persistent (round 5), 2 = dynamic queue (round 6)"""
    res = {}
    for v in ('0', '1', '2', '0', '1', '2'):
        os.environ['N3D_PS_PERSIST'] = v
        y = out(fn())
        res.setdefault(v, []).append((t_us(fn), y.clone()))
    same = torch.equal(res['0'][0][1], res['1'][0][1]) and torch.equal(res['0'][0][1], res['2'][0][1])
    a, b, c = (min(t for t, _ in res[k]) for k in ('0', '1', '2'))
    print(f'{tag} ({gf:.0f} GF): launch per tile {a:7.1f} us ({gf / a * 1e3:4.0f} TF)  static persistent {b:7.1f} us ({100 * (a / b - 1):+5.1f} %)  dynamic queue {c:7.1f} us ({gf / c * 1e3:4.0f} TF, {100 * (a / c - 1):+5.1f} %)'
          f'  bitwise equal: {same}', flush=True)
    os.environ['N3D_PS_PERSIST'] = '0'
    assert int(_lib.tickets().abs().sum()) == 0, 'queue words not re-armed'


cg.PS_TICKETS = True


for (N, I, O, H) in [(4, 256, 256, 128), (4, 128, 128, 256), (4, 256, 256, 256), (4, 128, 128, 512)]:
    wt = cg.prep_weight_bf16x3(torch.randn(O, I, 3, 3, device=dev) / (3 * I ** 0.5))
    xs = cg.split8_from_nchw(torch.randn(N, I, H, H, device=dev))
    y = torch.empty(N, O, H, H, device=dev)
    ab(f'stride-1 N{N} I{I} O{O} {H}x{H}', lambda: cg.conv_launch(xs, wt, 3, 0, O, out=y, bf16x3=True), lambda r: r, 2.0 * N * O * I * 9 * H * H / 1e9)
for (N, I, O, H) in [(4, 512, 512, 32), (4, 512, 256, 64), (4, 256, 128, 128), (4, 256, 128, 256)]:
    wt = cg.prep_weight_bf16x3(torch.randn(O, I, 3, 3, device=dev) / (3 * I ** 0.5))
    xs = cg.split8_from_nchw(torch.randn(N, I, H, H, device=dev))
    ab(f'transposed N{N} I{I} O{O} {H}x{H}', lambda: cg.conv_launch(xs, wt, 3, 2, O, bf16x3=True, out_c8=True), lambda r: r.data, 2.0 * N * O * I * 9 * H * H / 1e9)
