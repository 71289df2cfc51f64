#!/usr/bin/env python3
"""The renderer against the oracle (teacher-forced importance depths) over shapes whose LAST decode pass is partly filled in every way — few samples, one ray per wave, a
two-sample importance pass — for both decoders; prints the rays beyond 1e-3 and their per-channel errors.  The sweep that exposed (and now guards) the tail handling of round 6:
profiles/r06_render_split_ab.txt.  GPU box: python tools/render_tail_sweep.py"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from next3d_amd import _lib, demo as camera_utils
from oracle import renderer, cases
dev = torch.device('cuda', 0)
def _gen(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed); return torch.randn(shape, generator=g) * scale
def run(R, Sc, Sf, PH, PW, split):
    N = 2
    planes = _gen((N, 3, 32, PH, PW), 60 + R, 2.0)
    P = {'decoder.net.0.weight': _gen((64, 32), 61), 'decoder.net.0.bias': _gen((64,), 62, 0.3), 'decoder.net.2.weight': _gen((33, 64), 63), 'decoder.net.2.bias': _gen((33,), 64, 0.3)}
    w1 = (P['decoder.net.0.weight'] / np.sqrt(32)).contiguous()
    w2t = torch.cat([(P['decoder.net.2.weight'] / np.sqrt(64)).t(), torch.zeros(64, 1)], 1).contiguous()
    c = torch.cat([camera_utils.demo_camera_params(angle_y=a, angle_p=-0.2)[0] for a in (0.35, -0.3)], 0).float()
    ray_o, ray_d = renderer.ray_sampler(c[:, :16].reshape(N, 4, 4), c[:, 16:25].reshape(N, 3, 3), R)
    jitter, u = cases.rng_inputs(N, R, Sc, max(Sf, 1)); u = u[:, :Sf]
    opts = dict(depth_resolution=Sc, depth_resolution_importance=Sf, ray_start=2.25, ray_end=3.3, box_warp=1)
    fine = []
    rgb, depth, wsum = renderer.importance_renderer(P, 'decoder', planes, ray_o, ray_d, opts, jitter, u, fine_depths_out=fine)
    t = dict(dtype=torch.float32, device=dev)
    d = [x.contiguous().to(dev) for x in (planes.permute(0, 1, 3, 4, 2).contiguous(), c[:, :16], c[:, 16:25], torch.linspace(2.25, 3.3, Sc), jitter, u, w1, P['decoder.net.0.bias'], w2t, P['decoder.net.2.bias'])]
    feat, dep, ws_, bounds = torch.empty(N, 32, R, R, **t), torch.empty(N, 1, R, R, **t), torch.empty(N, R * R, **t), torch.empty(2, **t)
    ro = _lib.RenderOpts(); ro.ray_start, ro.ray_end, ro.box_side = 2.25, 3.3, 1.0
    fin = fine[0].reshape(N, R * R, Sf).contiguous().to(dev)
    ro.fine_depths_in = _lib.ptr(fin); ro.decoder_split_bf16 = split
    _lib.check(_lib.lib().n3d_render_rays_ex(*[_lib.ptr(x) for x in d], _lib.ptr(feat), _lib.ptr(dep), _lib.ptr(ws_), _lib.ptr(bounds), N, R, Sc, Sf, PH, PW, float((3.3 - 2.25) / (Sc - 1)), 2.0, ro, _lib.stream()))
    e_rgb = (feat.cpu().reshape(N, 32, R * R).permute(0, 2, 1) - rgb).abs()
    e_dep = (dep.cpu().reshape(N, R * R) - depth[..., 0]).abs()
    e_w = (ws_.cpu() - wsum.reshape(N, R * R)).abs()
    bad = (e_rgb.amax(-1) > 1e-3).nonzero().tolist()
    print(f'R{R} {Sc}+{Sf} split {split}: rgb max {float(e_rgb.max()):.2e}, depth max {float(e_dep.max()):.2e} wsum max {float(e_w.max()):.2e}; bad rays {bad[:20]} ({len(bad)})')
    for n_, r_ in bad[:3]:
        print('   ray', n_, r_, 'channel errors', [f'{float(x):.0e}' for x in e_rgb[n_, r_]], 'depth err', float(e_dep[n_, r_]), 'w err', float(e_w[n_, r_]))
for cfg in [(5, 24, 24, 16, 40), (3, 20, 12, 24, 24), (7, 40, 8, 32, 32), (6, 16, 48, 32, 32), (2, 48, 48, 32, 32), (5, 48, 48, 32, 32), (4, 33, 31, 24, 24), (6, 8, 8, 16, 16), (6, 24, 24, 16, 40), (4, 64, 17, 24, 24), (3, 47, 2, 16, 16), (1, 48, 48, 16, 16)]:
    run(*cfg, 0); run(*cfg, 1)
