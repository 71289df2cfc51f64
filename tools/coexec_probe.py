#!/usr/bin/env python3
"""Do MFMA-bound 3x3 convolution workgroups and HBM-bound LDS-free kernels of ANOTHER stream really share CUs (both progress), or does the chip time-slice them?
Per pair: the convolution alone, the streaming kernel alone (20 launches each, one stream), then both at once on two streams; `overlap` = (t_a + t_b - t_both) / min(t_a, t_b):
1 = the shorter one ran entirely in the other's shadow, 0 = they ran one after the other.  Tuning build (N3D_PS_NBUF forces the one- / two-workgroups-per-CU form):
    tools/build_variant.sh tune conv2d_ps_bf16x3.hip -DN3D_TUNING && N3D_LIB=tools/probe/libn3d_tune.so python tools/coexec_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from next3d_amd import _lib
from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
from next3d_amd.torch_utils.ops import upfirdn2d as uf
dev = torch.device('cuda', 0)
REP = 20
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def t_ms(jobs):
    """jobs: [(stream, fn)]; every fn REP times on its stream, all streams at once; wall time from a common start event to the last stream's end"""
    for s, fn in jobs:
        with torch.cuda.stream(s):
            fn(); fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        start = torch.cuda.Event(enable_timing=True)
        start.record()
        ends = []
        for s, fn in jobs:
            s.wait_event(start)
            with torch.cuda.stream(s):
                for _ in range(REP):
                    fn()
                e = torch.cuda.Event(enable_timing=True); e.record(); ends.append(e)
        torch.cuda.synchronize()
        best = min(best, max(start.elapsed_time(e) for e in ends))
    return best / REP


big = torch.randn(64 * 1024 * 1024, device=dev); big2 = torch.empty_like(big)          # 256 MB each
def copy(): big2.copy_(big)                                                            # LDS-free streaming kernel (ATen copy): 512 MB of traffic
xc = torch.randn(4, 128, 257, 257, device=dev)
def combine_like(): torch.add(big, 1.0, out=big2)

for (N, I, O, H, nbuf) in [(4, 512, 512, 64, 2), (4, 256, 256, 128, 2), (4, 256, 256, 128, 1), (4, 128, 128, 256, 2), (4, 128, 128, 256, 1)]:
    os.environ['N3D_PS_NBUF'] = str(nbuf)
    wt = cg.prep_weight_bf16x3(torch.randn(O, I, 3, 3, device=dev) / (3 * I ** 0.5))
    xs = cg.split8_from_nchw(torch.randn(N, I, H, H, device=dev))
    y = torch.empty(N, O, H, H, device=dev)
    conv = lambda: cg.conv_launch(xs, wt, 3, 0, O, out=y, bf16x3=True)
    ta, tb = t_ms([(sa, conv)]), t_ms([(sb, copy)])
    tab = t_ms([(sa, conv), (sb, copy)])
    print(f'stride-1 N{N} I{I} O{O} {H}x{H} nbuf{nbuf}: conv alone {ta * 1e3:7.1f} us, 512 MB copy alone {tb * 1e3:7.1f} us, both {tab * 1e3:7.1f} us per pair -> overlap {(ta + tb - tab) / min(ta, tb):5.2f}', flush=True)
