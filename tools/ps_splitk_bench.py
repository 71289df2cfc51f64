#!/usr/bin/env python3
"""Few-tile 3x3 stride-1 layers: the register-staged split-bf16 kernel with host-chosen split-K (NCHW input) against one conversion
pass + the pre-split LDS-DMA kernel with the library's split-K (n3d_conv2d_split8_ksplit) — in-stream time per layer (events around 20
back-to-back launches).  Also the transposed pre-split kernel on the benchmark's shapes.  Usage (GPU box): python tools/ps_splitk_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from next3d_amd import _lib
    from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
    dev = torch.device('cuda', 0)

    def timeit(fn, n=20):
        fn(); fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    print('--- stride 1: register-staged + split-K  vs  conversion + pre-split kernel (library split-K)')
    shapes = [] if not hasattr(_lib._handle(), 'n3d_conv2d_split8_ksplit') else [(4, 512, 512, 32, 32), (4, 1024, 512, 32, 32), (1, 512, 512, 64, 64), (1, 1024, 512, 64, 64), (1, 256, 256, 128, 128), (1, 512, 256, 128, 128),
                            (2, 512, 512, 32, 32), (8, 512, 512, 32, 32)]
    for (N, I, O, H, W) in shapes:
        x = torch.randn(N, I, H, W, device=dev)
        st = torch.rand(N, I, device=dev) + 0.5
        wt16 = cg.prep_weight_bf16x3(torch.randn(O, I, 3, 3, device=dev) / (3 * I ** 0.5))
        gf = 2 * N * O * I * 9 * H * W / 1e9
        ks_ps = _lib.lib().n3d_conv2d_split8_ksplit(N, I, O, H, W)
        t_reg = timeit(lambda: cg.conv_launch(x, wt16, 3, 0, O, style=st, bf16x3=True))
        line = f'N{N} {I:4d}->{O:4d} {H:3d}x{W:<3d} {gf:6.1f} GF | register-staged ksplit {cg.pick_ksplit_bf16x3(N, I, O, H, W, 0)}: {t_reg:7.1f} us {gf / t_reg * 1e3:6.1f} TF'
        if ks_ps > 0:
            xs = cg.split8_from_nchw(x, st)
            t_cv = timeit(lambda: cg.split8_from_nchw(x, st))
            t_ps = timeit(lambda: cg.conv_launch(xs, wt16, 3, 0, O, bf16x3=True))
            y0, y1 = cg.conv_launch(x, wt16, 3, 0, O, style=st, bf16x3=True), cg.conv_launch(xs, wt16, 3, 0, O, bf16x3=True)
            line += f' | pre-split ksplit {ks_ps}: {t_ps:7.1f} us {gf / t_ps * 1e3:6.1f} TF (+ conversion {t_cv:5.1f} us)  max diff {float((y0 - y1).abs().max()):.1e}'
        print(line)
    print('--- few-position transposed layers: register-staged + split-K + reduce (NCHW)  vs  conversion + pre-split kernel writing NCHW')
    for (N, I, O, H, W) in [(4, 512, 512, 4, 4), (4, 512, 512, 8, 8), (4, 512, 512, 16, 16), (4, 512, 512, 32, 32), (1, 512, 512, 16, 16), (1, 512, 512, 32, 32),
                            (1, 512, 256, 64, 64), (1, 256, 128, 128, 128), (2, 512, 256, 64, 64)]:
        x = torch.randn(N, I, H, W, device=dev)
        st = torch.rand(N, I, device=dev) + 0.5
        dco = torch.rand(N, O, device=dev) + 0.5
        wt16 = cg.prep_weight_bf16x3(torch.randn(O, I, 3, 3, device=dev) / (3 * I ** 0.5))
        gf = 2 * N * O * I * 9 * H * W / 1e9
        t_reg = timeit(lambda: cg.conv_launch(x, wt16, 3, 2, O, style=st, epilogue=_lib.make_epilogue(row_scale=dco), bf16x3=True, row_pitch=True))
        xs = cg.split8_from_nchw(x, st)
        t_cv = timeit(lambda: cg.split8_from_nchw(x, st))
        t_ps = timeit(lambda: cg.conv_launch(xs, wt16, 3, 2, O, epilogue=_lib.make_epilogue(row_scale=dco), bf16x3=True, row_pitch=True))
        y0 = cg.conv_launch(x, wt16, 3, 2, O, style=st, epilogue=_lib.make_epilogue(row_scale=dco), bf16x3=True, row_pitch=True)
        y1 = cg.conv_launch(xs, wt16, 3, 2, O, epilogue=_lib.make_epilogue(row_scale=dco), bf16x3=True, row_pitch=True)
        print(f'N{N} {I:4d}->{O:4d} {H:3d}x{W:<3d} {gf:6.2f} GF | register-staged ksplit {cg.pick_ksplit_bf16x3(N, I, O, H, W, 2)}: {t_reg:7.1f} us | pre-split -> NCHW: {t_ps:7.1f} us '
              f'(+ conversion {t_cv:5.1f} us)  max diff {float((y0 - y1).abs().max()):.1e}')
    print('--- transposed pre-split kernel (c8 out)')
    for (N, I, O, H, W) in [(4, 512, 256, 64, 64), (4, 256, 128, 128, 128), (4, 256, 128, 256, 256), (4, 512, 512, 32, 32), (4, 32, 256, 128, 128), (1, 512, 256, 64, 64),
                            (1, 256, 128, 128, 128)]:
        x = torch.randn(N, I, H, W, device=dev)
        st = torch.rand(N, I, device=dev) + 0.5
        wt16 = cg.prep_weight_bf16x3(torch.randn(O, I, 3, 3, device=dev) / (3 * I ** 0.5))
        gf = 2 * N * O * I * 9 * H * W / 1e9
        xs = cg.split8_from_nchw(x, st)
        t = timeit(lambda: cg.conv_launch(xs, wt16, 3, 2, O, bf16x3=True, out_c8=True))
        print(f'N{N} {I:4d}->{O:4d} {H:3d}x{W:<3d} {gf:6.1f} GF | {t:7.1f} us {gf / t * 1e3:6.1f} TF')


if __name__ == '__main__':
    main()
