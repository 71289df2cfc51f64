import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from next3d_amd import demo, _lib
from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
dev = torch.device('cuda', 0)
G, _ = demo.build_generator(dev)
S = G._prep()
d = np.load('tests/golden/case_r64_s48.npz')
v = torch.from_numpy(d['v']).to(dev)
vv, lms = v[:, :5023].contiguous(), v[:, 5023:].contiguous()
N, V, Lm, F, views, H, W = 2, 5023, 68, S.faces.shape[0], 4, 256, 256
f32 = dict(dtype=torch.float32, device=dev)
import ctypes
RL = None
if os.environ.get('RASTER_LIB'):       # the rasteriser from another build / code object (bisecting)
    RL = ctypes.CDLL(os.environ['RASTER_LIB'])
    RL.n3d_rasterize_views.restype = ctypes.c_int
    RL.n3d_rasterize_views.argtypes = _lib.lib().n3d_rasterize_views.argtypes
def raster():
    tv = torch.empty(N * views * V * 3, **f32); zbuf = torch.empty(N * views * H * W, dtype=torch.int64, device=dev)
    grid = torch.empty(N * views, H, W, 2, **f32); alpha4 = torch.empty(N, views, H, W, **f32); lm2d = torch.empty(N, Lm, 2, **f32)
    sh = G.orth_shift.reshape(-1).tolist()
    fn = RL.n3d_rasterize_views if RL is not None else _lib.lib().n3d_rasterize_views
    _lib.check(fn(_lib.ptr(vv), _lib.ptr(lms), _lib.ptr(S.rot), _lib.ptr(S.faces), _lib.ptr(S.face_uv), _lib.ptr(S.uv_mask),
               S.uv_mask.shape[0], S.uv_mask.shape[1], _lib.ptr(tv), _lib.ptr(zbuf), _lib.ptr(grid), _lib.ptr(alpha4), _lib.ptr(lm2d), N, V, Lm, F, views, H, W,
               sh[0], sh[1], sh[2], float(G.orth_scale.item()), 0, 1, _lib.stream()))
    return tv, zbuf, grid, alpha4, lm2d
ref = raster(); torch.cuda.synchronize()
side = torch.cuda.Stream()
kind = sys.argv[2] if len(sys.argv) > 2 else 'mode0'
x = torch.randn(4, 256, 128, 128, device=dev); w = torch.randn(256, 256, 3, 3, device=dev) / 48
if kind == 'fp32':
    wt = cg.prep_weight(w); launch = lambda: cg.conv_launch(x, wt, 3, 0, 256)
elif kind == 'up':
    wt = cg.prep_weight_bf16x3(w); launch = lambda: cg.conv_launch(x, wt, 3, 2, 256, bf16x3=True, row_pitch=True)
elif kind == 's2':
    x = torch.randn(4, 256, 129, 129, device=dev); wt = cg.prep_weight_bf16x3(w); launch = lambda: cg.conv_launch(x, wt, 3, 1, 256, bf16x3=True)
elif kind == 'mode0_nw4':
    x = torch.randn(4, 256, 32, 32, device=dev); wt = cg.prep_weight_bf16x3(w); launch = lambda: [cg.conv_launch(x, wt, 3, 0, 256, bf16x3=True, ksplit=1) for _ in range(6)]
elif kind == 'matmul':
    a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16); launch = lambda: a @ a
else:
    wt = cg.prep_weight_bf16x3(w); launch = lambda: cg.conv_launch(x, wt, 3, 0, 256, bf16x3=True)
main = torch.cuda.Stream() if len(sys.argv) > 1 else torch.cuda.current_stream()
torch.cuda.set_stream(main)
badn = 0
for it in range(12):
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(4):
            launch()
    r = raster()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    names = ['tv', 'zbuf', 'grid', 'alpha', 'lm2d']
    diff = {n: int((a != b).sum()) for n, a, b in zip(names, r, ref)}
    if any(diff.values()) and it < 3: print('  ', diff)
    badn += int(any(diff.values()))
    if any(diff.values()):
        import time; time.sleep(0.05); torch.cuda.synchronize()
        late = int((r[1] != ref[1]).sum())
        print('  zbuf mismatches right after sync:', diff['zbuf'], ' 50 ms later:', late)
print(kind, 'corrupted', badn, 'of 12')
