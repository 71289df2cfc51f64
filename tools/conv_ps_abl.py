#!/usr/bin/env python3
"""Ablation timing of conv2d_ps1 / ps2_bf16x3_kernel (N3D_CONV_DBG bits: 1 skip stores, 2 skip MFMA, 4 skip the DMA of chunks > 0,
8 no per-chunk wait / barrier) on the layer shapes of the benchmark, next to the register-staged kernel on the same shape.
Timing only — the ablated variants compute garbage.  Usage (GPU box): python tools/conv_ps_abl.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from next3d_amd import _lib
from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
dev = torch.device('cuda', 0)


def t_us(fn, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for (N, I, O, H, W) in [(4, 512, 512, 64, 64), (4, 256, 256, 128, 128), (4, 128, 128, 256, 256), (4, 128, 128, 512, 512)]:
    x = torch.randn(N, I, H, W, device=dev)
    wt = cg.prep_weight_bf16x3(torch.randn(O, I, 3, 3, device=dev) / (3 * I ** 0.5))
    s = _lib.Split8(N, I, H, W, dev)
    s.data.copy_(torch.randn(s.data.numel(), device=dev).bfloat16())
    y = torch.empty(N, O, H, W, device=dev)
    gf = 2.0 * N * O * I * 9 * H * W / 1e9
    os.environ['N3D_CONV_DBG'] = '0'
    base = t_us(lambda: cg.conv_launch(x, wt, 3, 0, O, out=y, bf16x3=True))
    row = [f'N{N} I{I} O{O} {H}x{W} ({gf:.0f} GF): register-staged {base:7.1f} us {gf / base * 1e3:6.1f} TF |']
    for nbuf in (2, 1):                                   # N3D_PS_NBUF: 2 = one workgroup per CU, double-buffered; 1 = two per CU
        os.environ['N3D_PS_NBUF'] = str(nbuf)
        row.append(f'| nbuf{nbuf}:')
        for dbg in ((0, 1, 2, 4, 5) if nbuf == 1 else (0, 1, 5)):
            os.environ['N3D_CONV_DBG'] = str(dbg)
            t = t_us(lambda: cg.conv_launch(s, wt, 3, 0, O, out=y, bf16x3=True))
            row.append(f'dbg{dbg}: {t:6.1f} us ({gf / t * 1e3:5.0f})')
    os.environ.pop('N3D_PS_NBUF')
    os.environ['N3D_CONV_DBG'] = '0'
    print(' '.join(row), flush=True)

# ---- the FIR in front of those layers: float32 path (fir4_vec_kernel) vs c8 -> split8 (fir4_c8_split8_kernel), same epilogue
from next3d_amd.torch_utils.ops import upfirdn2d as uf
f = uf.setup_filter([1, 3, 3, 1]).to(dev)
for (N, C, H) in [(4, 512, 64), (4, 256, 128), (4, 128, 256), (4, 128, 512)]:
    zh = H + 1
    z = torch.empty(N, C, zh, (zh + 3) // 4 * 4, device=dev)[..., :zh].normal_()
    zc = _lib.C8(N, C, zh, zh, dev); zc.data.normal_()
    noise, bias, ns, st = torch.randn(H, H, device=dev), torch.randn(C, device=dev), torch.tensor(0.1, device=dev), torch.rand(N, C, device=dev) + 0.5
    act = dict(noise=noise, noise_strength=ns, bias=bias, act='lrelu', gain=1.414)
    a = t_us(lambda: uf.upfirdn2d(z, f, padding=[1, 1, 1, 1], gain=4, _epilogue=_lib.make_epilogue(**act)))
    b = t_us(lambda: uf._fir4_split8(zc, f, 4, _lib.make_epilogue(**act), st))
    gb = 8.0 * N * C * H * H / 1e9
    print(f'FIR N{N} C{C} -> {H}x{H}: float32 {a:7.1f} us ({gb / a * 1e3:5.2f} TB/s)   c8->split8 {b:7.1f} us ({gb / b * 1e3:5.2f} TB/s)', flush=True)

# ---- transposed layers: register-staged (NCHW pitched out) vs pre-split (split8 in, c8 out) + the one-off conversion
for (N, I, O, H) in [(4, 512, 512, 32), (4, 512, 256, 64), (4, 256, 128, 128), (4, 256, 128, 256)]:
    x = torch.randn(N, I, H, H, device=dev); st = torch.rand(N, I, device=dev) + 0.5
    wt = cg.prep_weight_bf16x3(torch.randn(O, I, 3, 3, device=dev) / (3 * I ** 0.5))
    gf = 2.0 * N * O * I * 9 * H * H / 1e9
    a = t_us(lambda: cg.conv_launch(x, wt, 3, 2, O, style=st, bf16x3=True, row_pitch=True))
    a8 = t_us(lambda: cg.conv_launch(x, wt, 3, 2, O, style=st, bf16x3=True, out_c8=True, ksplit=1))
    cv = t_us(lambda: cg.split8_from_nchw(x, st))
    xs = cg.split8_from_nchw(x, st)
    b = t_us(lambda: cg.conv_launch(xs, wt, 3, 2, O, bf16x3=True, out_c8=True))
    abl = []
    for dbg in (1, 4, 5):                                 # no stores / no DMA after the first chunk / neither
        os.environ['N3D_CONV_DBG'] = str(dbg)
        abl.append(f'dbg{dbg} {t_us(lambda: cg.conv_launch(xs, wt, 3, 2, O, bf16x3=True, out_c8=True)):6.1f} us')
    os.environ['N3D_CONV_DBG'] = '0'
    print(f'UP N{N} I{I} O{O} {H}x{H} ({gf:.0f} GF): register-staged {a:7.1f} us ({gf / a * 1e3:4.0f} TF)  same, c8 out {a8:7.1f} us  | pre-split {b:7.1f} us ({gf / b * 1e3:4.0f} TF) + conversion {cv:6.1f} us | ' + ' '.join(abl), flush=True)

# ---- stride-2 encoder layers: register-staged vs pre-split (split8 in) + the conversion pass in front
for (N, I, O, H, ks) in [(4, 128, 256, 257, 1), (4, 256, 512, 129, 1), (4, 512, 512, 65, 4), (4, 512, 512, 65, 2), (4, 512, 512, 65, 1)]:
    x = torch.randn(N, I, H, H, device=dev)
    wt = cg.prep_weight_bf16x3(torch.randn(O, I, 3, 3, device=dev) / (3 * I ** 0.5))
    oh = (H - 3) // 2 + 1
    gf = 2.0 * N * O * I * 9 * oh * oh / 1e9
    a = t_us(lambda: cg.conv_launch(x, wt, 3, 1, O, bf16x3=True, ksplit=ks))
    xs = cg.split8_from_nchw(x)
    cv = t_us(lambda: cg.split8_from_nchw(x))
    b = t_us(lambda: cg.conv_launch(xs, wt, 3, 1, O, bf16x3=True, ksplit=ks))
    print(f'S2 N{N} I{I} O{O} {H}x{H} ksplit{ks} ({gf:.0f} GF): register-staged {a:7.1f} us ({gf / a * 1e3:4.0f} TF) | pre-split {b:7.1f} us ({gf / b * 1e3:4.0f} TF) + conversion {cv:6.1f} us', flush=True)
