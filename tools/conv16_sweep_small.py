#!/usr/bin/env python3
"""Split-K sweep (1..32) of the split-bf16 3x3 kernels on the low-resolution layers of the generator (N=4, 512 channels):
which factor minimises conv + reduce time.  Usage (GPU box): python tools/conv16_sweep_small.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from next3d_amd import _lib
from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
dev = torch.device('cuda')
N = 4
for (I, O, H, mode) in [(512, 512, 4, 0), (512, 512, 8, 0), (512, 512, 16, 0), (512, 512, 32, 0), (1024, 512, 8, 0), (1024, 512, 16, 0), (1024, 512, 32, 0),
                        (512, 512, 4, 2), (512, 512, 8, 2), (512, 512, 16, 2), (512, 512, 32, 2)]:
    x = torch.randn(N, I, H, H, device=dev); w = torch.randn(O, I, 3, 3, device=dev); s = torch.randn(N, I, device=dev)
    wt16 = cg.prep_weight_bf16x3(w)
    epi = _lib.make_epilogue(act="lrelu") if mode != 2 else _lib.make_epilogue(row_scale=torch.rand(N, O, device=dev) + 0.5)
    row = [f'I{I} O{O} {H}x{H} mode{mode}: blocks={_lib.lib().n3d_conv2d_bf16x3_blocks(N, O, H, H, mode)} auto={cg.pick_ksplit_bf16x3(N, I, O, H, H, mode)} |']
    for ks in (1, 2, 4, 8, 16, 32):
        if I // ks < 16:
            continue
        for _ in range(3):
            cg.conv_launch(x, wt16, 3, mode, O, style=s, epilogue=epi, bf16x3=True, ksplit=ks, row_pitch=mode == 2)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            cg.conv_launch(x, wt16, 3, mode, O, style=s, epilogue=epi, bf16x3=True, ksplit=ks, row_pitch=mode == 2)
        e1.record(); torch.cuda.synchronize()
        row.append(f'ks{ks}: {e0.elapsed_time(e1) / 30 * 1e3:6.1f}')
    print(' '.join(row), flush=True)
