#!/usr/bin/env python3
"""Rasteriser under a co-resident 8-wave convolution of ANOTHER stream (round-1 finding).  For every
tools/probe/libraster_v<K>.so (csrc/raster.hip built with -DRASTER_VARIANT=K by tools/build_raster_variants.sh) run the
rasteriser 12 times while `conv2d_bf16x3` kernels run on a side stream, compare every output with a quiet run, and CLASSIFY:
  missing  = z-buffer entry still cleared where the quiet run has a face     wrong = another face / depth won
  faces / blocks = distinct faces with missing pixels and the raster_faces workgroups they belong to
  grid_nan = output pixels never written (outputs are poisoned with NaN first)
  static   = one set of scratch / output buffers for all iterations (no allocator ping-pong)
One process, one line per configuration (a diagnostic, not a test).   usage: python tools/dbg_race3.py [variants...]"""
import ctypes, glob, os, re, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from next3d_amd import _lib, mesh
from next3d_amd.generator import RENDERING_VIEWS, angle2matrix
from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
dev = torch.device('cuda', 0)
g = np.load(os.path.join(ROOT, 'tests/golden/demo_inputs.npz'))
mb = mesh.mesh_buffers(g['faces'], g['uvs'], g['uvfaces'])
faces = mb['faces'][0][:, [0, 2, 1]].to(torch.int32).contiguous().to(dev)
face_uv = mb['face_uvcoords'][0][:, [0, 2, 1]].contiguous().to(dev)
rot = torch.cat([angle2matrix(a) for a in RENDERING_VIEWS], 0).contiguous().to(dev)
uv_mask = torch.nn.functional.interpolate(mesh.synthetic_uv_face_mask().float(), [256, 256])[0, 0].contiguous().to(dev)
d = np.load(os.path.join(ROOT, 'tests/golden/case_r64_s48.npz'))
v = torch.from_numpy(d['v']).to(dev)
vv, lms = v[:, :5023].contiguous(), v[:, 5023:].contiguous()
N, V, Lm, F, views, H, W = 2, 5023, 68, faces.shape[0], 4, 256, 256
f32 = dict(dtype=torch.float32, device=dev)
names = ['tv', 'zbuf', 'grid', 'alpha', 'lm2d']


def alloc():
    return (torch.empty(N * views * V * 3, **f32), torch.empty(N * views * H * W, dtype=torch.int64, device=dev),
            torch.empty(N * views, H, W, 2, **f32), torch.empty(N, views, H, W, **f32), torch.empty(N, Lm, 2, **f32))


def load(path):
    L = ctypes.CDLL(path)
    L.n3d_rasterize_views.restype = ctypes.c_int
    L.n3d_rasterize_views.argtypes = _lib.lib().n3d_rasterize_views.argtypes
    return L


x0 = torch.randn(4, 256, 128, 128, device=dev); x1 = torch.randn(4, 256, 129, 129, device=dev)
w = torch.randn(256, 256, 3, 3, device=dev) / 48
wt = cg.prep_weight_bf16x3(w)
LOADS = {'mode0': lambda: cg.conv_launch(x0, wt, 3, 0, 256, bf16x3=True), 's2': lambda: cg.conv_launch(x1, wt, 3, 1, 256, bf16x3=True),
         'up': lambda: cg.conv_launch(x0, wt, 3, 2, 256, bf16x3=True), 'none': lambda: None}
side, main = torch.cuda.Stream(), torch.cuda.Stream()
torch.cuda.set_stream(main)


def run(tag, L, kind='mode0', static=False, fill=0, iters=12):
    keep = alloc()

    def raster():
        bufs = keep if static else alloc()
        tv, zbuf, grid, alpha4, lm2d = bufs
        grid.fill_(float('nan')); alpha4.fill_(float('nan'))
        _lib.check(L.n3d_rasterize_views(_lib.ptr(vv), _lib.ptr(lms), _lib.ptr(rot), _lib.ptr(faces), _lib.ptr(face_uv), _lib.ptr(uv_mask), 256, 256,
                                         _lib.ptr(tv), _lib.ptr(zbuf), _lib.ptr(grid), _lib.ptr(alpha4), _lib.ptr(lm2d), N, V, Lm, F, views, H, W,
                                         0.0, -0.01, -0.01, 5.0, fill, 1, _lib.stream()))
        return [t.clone() for t in bufs] if static else list(bufs)
    torch.cuda.synchronize()
    ref = [t.clone() for t in raster()]; torch.cuda.synchronize()
    bad, tot, detail = 0, {n: 0 for n in names}, []
    for it in range(iters):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(4):
                LOADS[kind]()
        r = raster()
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        ne = lambda a, b: ((a != b) & ~((a != a) & (b != b))) if a.is_floating_point() else (a != b)
        diff = {n: int(ne(a, b).sum()) for n, a, b in zip(names, r, ref)}
        for n in names:
            tot[n] += diff[n]
        if any(diff.values()):
            bad += 1
            if len(detail) < 2:
                zb, zr = r[1], ref[1]
                mism = zb != zr
                missing = mism & (zb == -1)
                fm = zr[missing] & 0xFFFFFFFF
                thread = (torch.nonzero(missing).flatten() // (H * W)) * F + fm
                uvbad = ne(r[2], ref[2]).any(-1).flatten() & ~mism
                detail.append(dict(it=it, **diff, missing=int(missing.sum()), wrong=int((mism & (zb != -1)).sum()), faces=int(torch.unique(thread).numel()),
                                   blocks=int(torch.unique(thread // 256).numel()), grid_nan=int((r[2] != r[2]).sum()), uv_bad_right_zbuf=int(uvbad.sum())))
    print(f'{tag:28s} co-resident={kind:5s} static={int(static)} fill={fill}: corrupted {bad:2d} of {iters}; mismatching elements {tot}', flush=True)
    for dd in detail:
        print('      ', dd, flush=True)
    return bad


libs = sorted(glob.glob(os.path.join(ROOT, 'tools/probe/libraster_v*.so')), key=lambda p: int(re.findall(r'_v(\d+)\.so', p)[0]))
want = set(sys.argv[1:])
for path in libs:
    k = re.findall(r'_v(\d+)\.so', path)[0]
    if want and k not in want:
        continue
    run(f'variant {k}', load(path))
base = load(os.path.join(ROOT, 'tools/probe/libraster_v0.so'))
run('variant 0', base, static=True)
run('variant 0', base, kind='none')
run('variant 0', base, kind='s2')
run('variant 0', base, kind='up')
ship = _lib.lib()
run('libn3d.so (shipped)', ship)
run('libn3d.so (shipped)', ship, kind='s2')
run('libn3d.so (shipped)', ship, fill=1)
