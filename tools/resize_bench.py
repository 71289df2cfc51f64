#!/usr/bin/env python3
"""n3d_resize_aa's four calls of a batch-4 frame, one by one (microseconds per launch): the mouth crop (box -> 64^2), the paste back (256^2 -> the box of the front plane), and the
super-resolution input resizes (64^2 -> 128^2, 32 + 3 channels).  GPU box: python tools/resize_bench.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from next3d_amd import _lib
dev = torch.device('cuda', 0)
L = _lib.lib()
N = 4
bbox = torch.tensor([[82, 122, 108, 148]] * N, dtype=torch.int32, device=dev)
front, crop = torch.randn(N, 32, 256, 256, device=dev), torch.empty(N, 32, 64, 64, device=dev)
mouths, stitch = torch.randn(N, 32, 256, 256, device=dev), torch.randn(N, 32, 256, 256, device=dev)
feat, up = torch.randn(N, 32, 64, 64, device=dev), torch.empty(N, 32, 128, 128, device=dev)
rgb_up = torch.empty(N, 3, 128, 128, device=dev)
calls = {
    'crop  box(40x40) -> 64x64, 32 ch': lambda: L.n3d_resize_aa(_lib.ptr(front), _lib.ptr(crop), _lib.ptr(bbox), None, N, 32, 256, 256, 64, 64, 0, _lib.stream()),
    'paste 256x256 -> box(40x40), 32 ch': lambda: L.n3d_resize_aa(_lib.ptr(mouths), _lib.ptr(stitch), None, _lib.ptr(bbox), N, 32, 256, 256, 256, 256, 1, _lib.stream()),
    'SR    64x64 -> 128x128, 32 ch': lambda: L.n3d_resize_aa(_lib.ptr(feat), _lib.ptr(up), None, None, N, 32, 64, 64, 128, 128, 0, _lib.stream()),
    'SR    64x64 -> 128x128, 3 of 32 ch': lambda: L.n3d_resize_aa_strided(_lib.ptr(feat), feat.stride(0), _lib.ptr(rgb_up), None, None, N, 3, 64, 64, 128, 128, 0, _lib.stream()),
}
for name, fn in calls.items():
    for _ in range(5):
        _lib.check(fn())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        fn()
    e1.record(); torch.cuda.synchronize()
    print(f'{os.environ.get("N3D_LIB", "shipped"):28s} {name:38s} {e0.elapsed_time(e1) / 50 * 1e3:7.1f} us', flush=True)
print('paste checksum', float(stitch.double().sum()))
