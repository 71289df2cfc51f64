#!/usr/bin/env python3
"""Time one split-bf16 conv shape over several split-K factors: conv16_sweep.py I O H W mode [N]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from next3d_amd import _lib
from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
I, O, H, W, mode = [int(a) for a in sys.argv[1:6]]
N = int(sys.argv[6]) if len(sys.argv) > 6 else 4
dev = torch.device('cuda')
x = torch.randn(N, I, H, W, device=dev); w = torch.randn(O, I, 3, 3, device=dev); s = torch.randn(N, I, device=dev)
wt16 = cg.prep_weight_bf16x3(w)
epi = _lib.make_epilogue(act="lrelu") if mode != 2 else _lib.make_epilogue(row_scale=torch.rand(N, O, device=dev) + 0.5)
gf = 2.0 * N * I * O * 9 * H * W / 1e9
print(f'I{I} O{O} {H}x{W} mode{mode} N{N}: blocks={_lib.lib().n3d_conv2d_bf16x3_blocks(N, O, H, W, mode)} auto ksplit={cg.pick_ksplit_bf16x3(N, I, O, H, W, mode)}')
for ks in (1, 2, 4, 8):
    if I // ks < 64:
        continue
    for _ in range(3):
        cg.conv_launch(x, wt16, 3, mode, O, style=s, epilogue=epi, bf16x3=True, ksplit=ks, row_pitch=mode == 2)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        cg.conv_launch(x, wt16, 3, mode, O, style=s, epilogue=epi, bf16x3=True, ksplit=ks, row_pitch=mode == 2)
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20
    print(f'  ksplit {ks}: {t * 1e3:7.1f} us  {gf / t:6.1f} TF')
