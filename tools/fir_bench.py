#!/usr/bin/env python3
"""tools/fir_bench.py — the FIR kernels that write split8, one launch at a time (HIP events, 20 launches each): the up-sampling
layers' c8 -> split8 FIR and the stride-2 layers' NCHW -> split8 pre-filter at the generator's shapes (batch 4), separable vs 16-tap.
    python tools/fir_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from next3d_amd import _lib                                   # noqa: E402
from next3d_amd.torch_utils.ops import upfirdn2d as uf        # noqa: E402


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    dev = torch.device('cuda:0')
    f = uf.setup_filter([1, 3, 3, 1]).to(dev)
    print('c8 -> split8 (up-sampling layers: FIR + noise + bias + lrelu + next style)')
    for n, c, h in [(4, 512, 64), (4, 256, 128), (4, 128, 256), (4, 256, 256), (4, 128, 512)]:
        z = _lib.C8(n, c, h + 1, h + 1, dev)
        z.data.normal_()
        style, bias, noise = torch.randn(n, c, device=dev), torch.randn(c, device=dev), torch.randn(h, h, device=dev)
        epi = _lib.make_epilogue(noise=noise, noise_strength=torch.tensor(0.1, device=dev), bias=bias, act='lrelu', gain=1.41)
        mb = 4 * n * c * ((h + 1) ** 2 + h * h) / 1e6
        row = []
        for sep in ('1', '0'):
            uf.FIR_SEP = ('all' if sep == '1' else False)
            us = timed(lambda: uf._fir4_split8(z, f, 4, epi, style))
            row.append(f'sep={sep}: {us:7.1f} us {mb / us:5.2f} TB/s')
        print(f'  [{n},{c},{h + 1},{h + 1}] {mb:7.1f} MB   ' + '   '.join(row))
    uf.FIR_SEP = 'all'
    print('NCHW -> split8, padding 2 (stride-2 layers\' pre-filter)')
    for n, c, h in [(4, 128, 256), (4, 256, 128), (4, 512, 64), (4, 512, 32), (4, 256, 64)]:
        x = torch.randn(n, c, h, h, device=dev)
        mb = 4 * n * c * (h * h + (h + 1) ** 2) / 1e6
        row = []
        for sep in ('1', '0'):
            uf.FIR_SEP = ('all' if sep == '1' else False)
            us = timed(lambda: uf._fir4_split8_nchw(x, f, 2))
            row.append(f'sep={sep}: {us:7.1f} us {mb / us:5.2f} TB/s')
        print(f'  [{n},{c},{h},{h}] {mb:7.1f} MB   ' + '   '.join(row))


if __name__ == '__main__':
    main()
