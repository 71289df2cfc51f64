#!/usr/bin/env python3
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from next3d_amd import _lib
from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
I, O, H, W, mode = [int(a) for a in sys.argv[1:6]]
N = 4
dev = torch.device('cuda')
x = torch.randn(N, I, H, W, device=dev); w = torch.randn(O, I, 3, 3, device=dev); s = torch.randn(N, I, device=dev)
wt16 = cg.prep_weight_bf16x3(w)
oh, ow = cg.out_shape(H, W, mode)
y = torch.empty(N, O, oh, ow, device=dev)
epi = _lib.make_epilogue(act='lrelu')
for _ in range(3):
    cg.conv_launch(x, wt16, 3, mode, O, out=y, style=s, epilogue=epi, bf16x3=True)
torch.cuda.synchronize()
