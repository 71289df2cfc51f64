#!/usr/bin/env python3
"""List every convolution / FIR launch of one generator forward with its shape and the kernel path it takes, timed one
by one (HIP events, launch stream).  Usage (GPU box): python tools/layer_trace.py [--precision fp32]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('N3D_OVERLAP_STATIC', '0')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--precision', default=None)
    ap.add_argument('--batch', type=int, default=4)
    a = ap.parse_args()
    from next3d_amd import demo, layers, _lib
    from next3d_amd.torch_utils.ops import conv2d_gradfix as cg, upfirdn2d as uf
    if a.precision:
        layers.set_precision(a.precision)
    dev = torch.device('cuda', 0)
    G, _ = demo.build_generator(dev)
    z, c, c_cond, v = demo.demo_batch(list(range(a.batch)), device=dev)
    rows, on = [], [False]
    orig_conv, orig_fir = cg.conv_launch, uf._launch

    def timed(fn, label):
        if not on[0]:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); y = fn(); e1.record(); torch.cuda.synchronize()
        rows.append((label, e0.elapsed_time(e1)))
        return y

    def conv(x, wt, ksize, mode, oc, *args, **kw):
        n, i, h, w = x.shape
        gh, gw = (h + 1, w + 1) if mode == 2 else cg.out_shape(h, w, mode)
        b16 = kw.get('bf16x3', False)
        pre = isinstance(x, _lib.Split8)
        ks = 1 if pre else (kw.get('ksplit') or (cg.pick_ksplit_bf16x3(n, i, oc, h, w, mode) if b16 else cg.pick_ksplit(n, i, oc, gh, gw, ksize, mode)))
        flops = 2.0 * n * i * oc * ksize * ksize * (h * w if mode != 1 else gh * gw)
        lab = (f'conv k{ksize} mode{mode} {"bf16x3" if b16 else "fp32  "} {"split8" if pre else "nchw  "}->{"c8  " if kw.get("out_c8") else "nchw"} '
               f'N{n} I{i:4d} O{oc:4d} {h:3d}x{w:<3d} ksplit{ks} gflop={flops / 1e9:7.2f}')
        return timed(lambda: orig_conv(x, wt, ksize, mode, oc, *args, **kw), lab)

    def fir(x, f2d, up, down, padding, *args, **kw):
        n, ch, h, w = x.shape
        lab = f'fir  up{up[0]} down{down[0]} N{n} C{ch:4d} {h:3d}x{w:<3d} pitch{x.stride(2)}'
        return timed(lambda: orig_fir(x, f2d, up, down, padding, *args, **kw), lab)

    orig_fir8, orig_cvt, orig_firn = uf._fir4_split8, cg.split8_from_nchw, uf._fir4_split8_nchw

    def fir8(x, *args, **kw):
        n, ch, h, w = x.shape
        return timed(lambda: orig_fir8(x, *args, **kw), f'fir4 c8->split8 N{n} C{ch:4d} {h:3d}x{w:<3d}')

    def cvt(x, *args, **kw):
        n, ch, h, w = x.shape
        return timed(lambda: orig_cvt(x, *args, **kw), f'nchw->split8 N{n} C{ch:4d} {h:3d}x{w:<3d}')

    def firn(x, *args, **kw):
        n, ch, h, w = x.shape
        return timed(lambda: orig_firn(x, *args, **kw), f'fir4 nchw->split8 N{n} C{ch:4d} {h:3d}x{w:<3d}')

    cg.conv_launch, uf._launch, uf._fir4_split8, cg.split8_from_nchw, uf._fir4_split8_nchw = conv, fir, fir8, cvt, firn
    for it in range(3):
        on[0] = it == 2
        ws = G.mapping(z, c_cond, truncation_psi=0.7, truncation_cutoff=14)
        G.synthesis(ws, c, v, neural_rendering_resolution=64, noise_mode='const', force_fp32=True)
        torch.cuda.synchronize()
    tot = sum(t for _, t in rows)
    for lab, t in rows:
        extra = ''
        if 'gflop=' in lab:
            extra = f'  {float(lab.split("gflop=")[1]) / t:7.1f} TF'
        print(f'{t * 1e3:8.1f} us  {lab}{extra}')
    print(f'total conv+fir {tot:.3f} ms over {len(rows)} launches')


if __name__ == '__main__':
    main()
