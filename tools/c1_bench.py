#!/usr/bin/env python3
"""tools/c1_bench.py — the 1x1 split-bf16 kernels one launch at a time (HIP events, 20 launches) at the generator's toRGB / fromrgb shapes
(batch 4): bytes = x + y (+ side output + low-resolution skip image) -> TB/s.     python tools/c1_bench.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from next3d_amd import _lib                                        # noqa: E402
from next3d_amd.torch_utils.ops import conv2d_gradfix as cg       # noqa: E402
from next3d_amd.torch_utils.ops import upfirdn2d as uf            # noqa: E402


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
    # (N, I, O, H, skip image, side output, split8 result)
    for n, i, o, h, skip, side, s8 in [(4, 128, 96, 256, True, False, False), (4, 256, 96, 128, True, True, False), (4, 512, 96, 64, True, True, False),
                                       (4, 128, 3, 256, True, False, False), (4, 256, 3, 128, True, True, False), (4, 128, 3, 512, True, False, False),
                                       (4, 256, 3, 256, True, True, False), (4, 32, 128, 256, False, False, True), (4, 32, 256, 128, False, False, True),
                                       (4, 32, 128, 256, False, False, False)]:
        x = torch.randn(n, i, h, h, device=dev)
        wt16 = cg.prep_weight_bf16x3(torch.randn(o, i, 1, 1, device=dev) / np.sqrt(i))
        st, st2, b = torch.randn(n, i, device=dev), torch.randn(n, i, device=dev), torch.randn(o, device=dev)
        low = torch.randn(n, o, h // 2, h // 2, device=dev)
        epi = _lib.make_epilogue(bias=b, residual=low, residual_up_filter=f) if skip else _lib.make_epilogue(bias=b)
        mb = 4 * n * h * h * (i + o + (i if side else 0) + (o / 4 if skip else 0)) / 1e6
        us = timed(lambda: cg.conv_launch(x, wt16, 1, 0, o, style=None if s8 else st, epilogue=epi, bf16x3=True, side_style=st2 if side else None,
                                          out_split8=s8))
        print(f'  [{n},{i}->{o},{h}x{h}] skip={int(skip)} side={int(side)} split8_out={int(s8)}  {mb:7.1f} MB {us:7.1f} us {mb / us:5.2f} TB/s')


if __name__ == '__main__':
    main()
