#!/usr/bin/env python3
"""Run one conv2d shape a few times (for rocprofv3 --pmc / --kernel-trace).  Usage: conv_one.py I O H W k mode [N] [iters]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from next3d_amd import _lib
from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
I, O, H, W, k, mode = [int(a) for a in sys.argv[1:7]]
N = int(sys.argv[7]) if len(sys.argv) > 7 else 4
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 3
dev = torch.device('cuda')
x = torch.randn(N, I, H, W, device=dev); w = torch.randn(O, I, k, k, device=dev); s = torch.randn(N, I, device=dev)
wt = cg.prep_weight(w)
oh, ow = cg.out_shape(H, W, mode)
y = torch.empty(N, O, oh, ow, device=dev)
epi = _lib.make_epilogue(act='lrelu')
for _ in range(iters):
    cg.conv_launch(x, wt, k, mode, O, out=y, style=s, epilogue=epi)
torch.cuda.synchronize()
