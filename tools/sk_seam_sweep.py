#!/usr/bin/env python3
"""Few-pixel 3x3 layers (conv2d_sk_bf16x3.hip): K slices over workgroups (KS, reduced by the last arriver inside the launch) x pixel groups per tile (PT),
per layer shape of the generator at batch 4 and batch 1.  Each configuration is a HIP graph of 20 back-to-back launches (no host time between them),
replayed 6 times; the figure is microseconds per launch including the ~1.5 us kernel boundary.  Needs the tuning build:
    tools/build_variant.sh tune conv2d_sk_bf16x3.hip -DN3D_TUNING && N3D_LIB=tools/probe/libn3d_tune.so python tools/sk_seam_sweep.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from next3d_amd import _lib
from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
dev = torch.device('cuda', 0)
REP = 20


def graph_us(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fn(); fn()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    pools = _lib.ticket_pools(_lib.new_ticket_pools(dev))
    with torch.cuda.graph(g), pools:
        for _ in range(REP):
            fn()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / REP)
    assert all(int(p.tickets.abs().sum()) == 0 for p in pools.all), 'arrival counters not re-armed'
    return best


shapes = []
for N in (4, 1):
    shapes += [(N, 512, 512, 4, 0), (N, 512, 512, 8, 0), (N, 512, 512, 16, 0), (N, 1024, 512, 8, 0), (N, 1024, 512, 16, 0),
               (N, 512, 512, 4, 2), (N, 512, 512, 8, 2), (N, 512, 512, 9, 1), (N, 512, 512, 17, 1)]
shapes += [(1, 512, 512, 32, 0), (1, 1024, 512, 32, 0), (1, 512, 512, 16, 2), (1, 512, 512, 33, 1)]
for (N, I, O, H, mode) in shapes:
    x = torch.randn(N, I, H, H, device=dev); w = torch.randn(O, I, 3, 3, device=dev) / (3 * I ** 0.5); s = torch.randn(N, I, device=dev)
    wt16 = cg.prep_weight_bf16x3(w)
    epi = _lib.make_epilogue(act='lrelu', row_scale=torch.rand(N, O, device=dev) + 0.5)
    elig = {0: cg.sk_eligible, 1: cg.sk_s2_eligible, 2: cg.up_sk_eligible}[mode]
    os.environ.pop('N3D_SK_KS', None); os.environ.pop('N3D_SK_PT', None)
    gf = 2.0 * N * O * I * 9 * (H * H if mode != 1 else ((H - 3) // 2 + 1) ** 2) / 1e9
    row = [f'N{N} I{I} O{O} {H}x{H} mode{mode} ({gf:5.2f} GF):']
    run = lambda: cg.conv_launch(x, wt16, 3, mode, O, style=s, epilogue=epi, bf16x3=True)
    if not elig(N, I, O, H, H):
        print(' '.join(row), 'not a few-pixel layer:', f'{graph_us(run):6.1f} us on the general kernels', flush=True)
        continue
    cg.SK_SEAM = False
    ref = run().clone()
    row.append(f'round-5 launch {graph_us(run):6.1f} us |')
    cg.SK_SEAM = True
    for pt in ((1, 2) if mode == 0 else (0,)):
        for ks in (1, 2, 4, 8):
            if (I // 16) % (8 * ks):
                continue
            os.environ['N3D_SK_KS'] = str(ks); os.environ['N3D_SK_PT'] = str(pt)
            if mode == 0 and not cg.sk_eligible(N, I, O, H, H):
                continue
            y = run()
            err = float((y - ref).abs().max()) / max(1.0, float(ref.abs().max()))
            assert err < 2e-5, (pt, ks, err)
            row.append(f'pt{pt} ks{ks}: {graph_us(run):6.1f}')
    os.environ.pop('N3D_SK_KS', None); os.environ.pop('N3D_SK_PT', None)
    row.append(f'| auto: {graph_us(run):6.1f}')
    print(' '.join(row), flush=True)
