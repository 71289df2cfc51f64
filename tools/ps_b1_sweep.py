#!/usr/bin/env python3
"""Batch-1 layers on the pre-split stride-1 kernel: split-K factor (N3D_PS_KS) x LDS-buffer form (N3D_PS_NBUF: 1 = two workgroups per CU, 2 = one), conv + reduce launch, HIP graph of 20,
microseconds per layer (tuning build).  tools/build_variant.sh tune conv2d_ps_bf16x3.hip -DN3D_TUNING && N3D_LIB=tools/probe/libn3d_tune.so python tools/ps_b1_sweep.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from next3d_amd import _lib
from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
dev = torch.device('cuda', 0)
REP = 20


def graph_us(fn):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), _lib.ticket_pools(_lib.new_ticket_pools(dev)):
        for _ in range(REP):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / REP)
    return best


for (N, I, O, H) in [(1, 512, 512, 32), (1, 1024, 512, 32), (1, 512, 512, 64), (1, 1024, 512, 64), (1, 256, 256, 128), (1, 512, 256, 128), (1, 128, 128, 256), (1, 256, 256, 256), (1, 128, 128, 512)]:
    wts = [cg.prep_weight_bf16x3(torch.randn(O, I, 3, 3, device=dev) / (3 * I ** 0.5)) for _ in range(4)]      # (rotating weights: colder than one layer back to back)
    xs = cg.split8_from_nchw(torch.randn(N, I, H, H, device=dev))
    y = torch.empty(N, O, H, H, device=dev)
    epi = _lib.make_epilogue(act='lrelu', row_scale=torch.rand(N, O, device=dev) + 0.5)
    k = [0]
    def run():
        k[0] += 1
        return cg.conv_launch(xs, wts[k[0] % 4], 3, 0, O, out=y, epilogue=epi, bf16x3=True)
    gf = 2.0 * N * O * I * 9 * H * H / 1e9
    os.environ.pop('N3D_PS_KS', None); os.environ.pop('N3D_PS_NBUF', None)
    if cg.split8_eligible(N, I, O, H, H):
        row = [f'N{N} I{I} O{O} {H}x{H} ({gf:5.1f} GF): default ks{cg.split8_ksplit(N, I, O, H, H)} {graph_us(run):6.1f} us |']
    else:          # (the model runs this layer on the few-pixel / register-staged kernels from float32 NCHW input)
        xn = torch.randn(N, I, H, H, device=dev); st = torch.randn(N, I, device=dev)
        t0 = graph_us(lambda: cg.conv_launch(xn, wts[0], 3, 0, O, out=y, style=st, epilogue=epi, bf16x3=True))
        row = [f'N{N} I{I} O{O} {H}x{H} ({gf:5.1f} GF): not pre-split by default (NCHW route {t0:6.1f} us) |']
    for nbuf in (2, 1):
        for ks in (1, 2, 4, 8, 16):
            if (I // 16) % ks or I // (16 * ks) < 2:
                continue
            os.environ['N3D_PS_KS'] = str(ks); os.environ['N3D_PS_NBUF'] = str(nbuf)
            t = graph_us(run)
            row.append(f'nbuf{nbuf} ks{ks}: {t:6.1f} ({gf / t * 1e3:3.0f} TF)')
    print(' '.join(row), flush=True)
