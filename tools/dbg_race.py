import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from next3d_amd import demo, _lib
from next3d_amd.torch_utils.ops import conv2d_gradfix as cg, upfirdn2d as uf
dev = torch.device('cuda', 0)
G, _ = demo.build_generator(dev)
d = np.load('tests/golden/case_r64_s48.npz')
v = torch.from_numpy(d['v']).to(dev)
vv, lms = v[:, :5023].contiguous(), v[:, 5023:].contiguous()
G._prep()
ref = [t.clone() for t in G.raster_geometry(vv, lms)]
torch.cuda.synchronize()
side = torch.cuda.Stream()
N = 4
def mk(I, O, H, W, k):
    x = torch.randn(N, I, H, W, device=dev); w = torch.randn(O, I, k, k, device=dev) / (I * k * k) ** 0.5
    return x, cg.prep_weight_bf16x3(w), O, k
cases = {
 'mode0_big': (mk(256, 256, 128, 128, 3), 0), 'mode0_32': (mk(512, 512, 32, 32, 3), 0), 'flat16': (mk(512, 512, 16, 16, 3), 0), 'flat4': (mk(512, 512, 4, 4, 3), 0),
 'up64': (mk(512, 256, 64, 64, 3), 2), 'up16': (mk(512, 512, 16, 16, 3), 2), 'up4': (mk(512, 512, 4, 4, 3), 2),
 'c1x1_big': (mk(128, 96, 256, 256, 1), 0), 'c1x1_small': (mk(512, 96, 16, 16, 1), 0), 's2': (mk(128, 256, 129, 129, 3), 1),
}
fir = G._prep().static.fir
def run_case(name):
    (x, wt, O, k), mode = cases[name]
    bad = 0
    for it in range(30):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(4):
                y = cg.conv_launch(x, wt, k, mode, O, bf16x3=True, row_pitch=(mode == 2))
                if mode == 2:
                    uf.upfirdn2d(y, fir, padding=[1, 1, 1, 1], gain=4)
        g = G.raster_geometry(vv, lms)
        torch.cuda.current_stream().wait_stream(side)
        ok = all(torch.equal(a, b) for a, b in zip(g, ref))
        if not ok and bad == 0:
            dg = (g[0] - ref[0]).abs().view(-1, 4, 256, 256, 2).amax(-1)
            print('   first corruption: grid px per (n,view)', [[int((dg[n, w] > 0).sum()) for w in range(4)] for n in range(dg.shape[0])], 'alpha diff', int(((g[1] - ref[1]).abs() > 0).sum()), 'bbox eq', bool(torch.equal(g[2], ref[2])))
        bad += int(not ok)
    print(f'{name:12s} corrupted raster runs: {bad}/30')
names = sys.argv[1].split(',') if len(sys.argv) > 1 else list(cases)
for name in names:
    run_case(name)
