#!/usr/bin/env python3
"""Few-pixel 3x3 layers with COLD weights (every launch another layer's 9.4 MB, 48 layers = 453 MB: beyond the 256 MB Infinity Cache), as in a forward —
against the same launches with the NEXT layer's weights being read by a second stream meanwhile (a prefetch into the Infinity Cache), and against hot weights.
One HIP graph per variant (fork / join through events), microseconds per layer."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from next3d_amd import _lib
from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
dev = torch.device('cuda', 0)
L = 48


def run_graph(body):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        body()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g), _lib.ticket_pools(_lib.new_ticket_pools(dev)):
        body()
    g.replay(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / L)
    return best


for (N, I, O, H, mode) in [(1, 512, 512, 4, 0), (1, 512, 512, 8, 0), (1, 512, 512, 16, 0), (4, 512, 512, 4, 0), (4, 512, 512, 8, 0), (1, 512, 512, 8, 2), (1, 512, 512, 32, 0)]:
    ws = [cg.prep_weight_bf16x3(torch.randn(O, I, 3, 3, device=dev) / (3 * I ** 0.5)) for _ in range(L)]
    x = torch.randn(N, I, H, H, device=dev); s = torch.randn(N, I, device=dev)
    epi = _lib.make_epilogue(act='lrelu', row_scale=torch.rand(N, O, device=dev) + 0.5)
    sink = torch.zeros(L, device=dev)
    side = torch.cuda.Stream()

    def cold():
        for k in range(L):
            cg.conv_launch(x, ws[k], 3, mode, O, style=s, epilogue=epi, bf16x3=True)

    def hot():
        for k in range(L):
            cg.conv_launch(x, ws[0], 3, mode, O, style=s, epilogue=epi, bf16x3=True)

    def prefetched():
        cur = torch.cuda.current_stream()
        for k in range(L):
            if k + 1 < L:                          # the next layer's weights, read by another stream while this layer runs
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    sink[k] = ws[k + 1].view(torch.int32).sum()
            cg.conv_launch(x, ws[k], 3, mode, O, style=s, epilogue=epi, bf16x3=True)
        cur.wait_stream(side)

    print(f'N{N} I{I} O{O} {H}x{H} mode{mode}: hot weights {run_graph(hot):6.1f} us, cold {run_graph(cold):6.1f} us, cold + next layer prefetched by a side stream {run_graph(prefetched):6.1f} us per layer', flush=True)
