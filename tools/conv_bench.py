#!/usr/bin/env python3
"""Per-layer micro-benchmark of the conv2d / upfirdn2d kernels on the generator's real layer shapes (GPU box only).
Prints TFLOP/s (conv) or GB/s (FIR) per shape.  Usage: python tools/conv_bench.py [N]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from next3d_amd import _lib
from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
from next3d_amd.torch_utils.ops import upfirdn2d as uf

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device('cuda')
# (I, O, H, W, k, mode, count-per-frame)
LAYERS = [
    (128, 128, 512, 512, 3, 0, 1), (256, 256, 256, 256, 3, 0, 1), (128, 128, 256, 256, 3, 0, 4), (256, 256, 128, 128, 3, 0, 4),
    (512, 512, 64, 64, 3, 0, 5), (1024, 512, 64, 64, 3, 0, 1), (512, 256, 128, 128, 3, 0, 1), (512, 512, 32, 32, 3, 0, 5),
    (1024, 512, 32, 32, 3, 0, 1), (512, 512, 16, 16, 3, 0, 4), (512, 512, 8, 8, 3, 0, 4), (512, 512, 4, 4, 3, 0, 4),
    (256, 128, 256, 256, 3, 2, 1), (32, 256, 128, 128, 3, 2, 1), (256, 128, 128, 128, 3, 2, 4), (512, 256, 64, 64, 3, 2, 4),
    (512, 512, 32, 32, 3, 2, 4), (512, 512, 16, 16, 3, 2, 3), (512, 512, 8, 8, 3, 2, 3), (512, 512, 4, 4, 3, 2, 3),
    (128, 256, 257, 257, 3, 1, 1), (256, 512, 129, 129, 3, 1, 1), (512, 512, 65, 65, 3, 1, 2), (512, 512, 33, 33, 3, 1, 1),
    (128, 32, 256, 256, 1, 0, 3), (128, 96, 256, 256, 1, 0, 1), (128, 3, 512, 512, 1, 0, 1), (256, 3, 256, 256, 1, 0, 1),
    (32, 128, 256, 256, 1, 0, 1), (32, 512, 64, 64, 1, 0, 2), (512, 32, 64, 64, 1, 0, 3),
]


def timeit(fn, iters=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


tot_ms = tot_fl = 0
for (I, O, H, W, k, mode, cnt) in LAYERS:
    x = torch.randn(N, I, H, W, device=dev)
    w = torch.randn(O, I, k, k, device=dev)
    wt = cg.prep_weight(w)
    s = torch.randn(N, I, device=dev)
    oh, ow = cg.out_shape(H, W, mode)
    y = torch.empty(N, O, oh, ow, device=dev)
    epi = _lib.make_epilogue(act='lrelu')
    ms = timeit(lambda: cg.conv_launch(x, wt, k, mode, O, out=y, style=s, epilogue=epi))
    fl = 2.0 * N * O * I * k * k * (H * W if mode == 2 else oh * ow)
    tot_ms += ms * cnt; tot_fl += fl * cnt
    print(f'conv I={I:4d} O={O:4d} {H:3d}x{W:3d} k={k} mode={mode} x{cnt}: {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s  ksplit={cg.pick_ksplit(N, I, O, (H + 1 if mode == 2 else oh), (W + 1 if mode == 2 else ow), k, mode)}')
print('--- bf16x3 (split-bf16)')
tot16_ms = tot16_fl = 0
for (I, O, H, W, k, mode, cnt) in LAYERS:
    if not cg.bf16x3_eligible(I, H, W, k, mode):
        continue
    x = torch.randn(N, I, H, W, device=dev); w = torch.randn(O, I, k, k, device=dev); s = torch.randn(N, I, device=dev)
    wt16 = cg.prep_weight_bf16x3(w)
    oh, ow = cg.out_shape(H, W, mode)
    y = torch.empty(N, O, oh, ow, device=dev)
    epi = _lib.make_epilogue(act='lrelu')
    ms = timeit(lambda: cg.conv_launch(x, wt16, 3, mode, O, out=y, style=s, epilogue=epi, bf16x3=True))
    fl = 2.0 * N * O * I * 9 * H * W
    tot16_ms += ms * cnt; tot16_fl += fl * cnt
    print(f'bf16x3 I={I:4d} O={O:4d} {H:3d}x{W:3d} mode={mode} x{cnt}: {ms:8.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s(fp32-equiv)')
print(f'bf16x3-eligible total: {tot16_ms:.2f} ms, {tot16_fl / tot16_ms / 1e9:.1f} TFLOP/s fp32-equiv')
print(f'conv total (weighted by per-frame count): {tot_ms:.2f} ms, {tot_fl / tot_ms / 1e9:.1f} TFLOP/s')

f = uf.setup_filter([1, 3, 3, 1]).to(dev)
for (C, H, W, up, down, pad, gain) in [(128, 513, 513, 1, 1, [1, 1, 1, 1], 4), (256, 257, 257, 1, 1, [1, 1, 1, 1], 4), (128, 257, 257, 1, 1, [1, 1, 1, 1], 4),
                                       (32, 128, 128, 2, 1, [2, 1, 2, 1], 4), (96, 128, 128, 2, 1, [2, 1, 2, 1], 4), (3, 256, 256, 2, 1, [2, 1, 2, 1], 4),
                                       (128, 256, 256, 1, 1, [2, 2, 2, 2], 1), (32, 256, 256, 1, 2, [1, 1, 1, 1], 1)]:
    x = torch.randn(N, C, H, W, device=dev)
    b = torch.randn(C, device=dev)
    epi = _lib.make_epilogue(bias=b, act='lrelu')
    y = uf.upfirdn2d(x, f, up=up, down=down, padding=pad, gain=gain, _epilogue=epi)
    ms = timeit(lambda: uf.upfirdn2d(x, f, up=up, down=down, padding=pad, gain=gain, _epilogue=epi))
    by = 4.0 * (x.numel() + y.numel())
    print(f'fir C={C:4d} {H:3d}x{W:3d} up={up} down={down}: {ms:8.3f} ms  {by / ms / 1e6:8.1f} GB/s')
