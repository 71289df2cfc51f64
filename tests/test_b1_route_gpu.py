"""-m gpu tests of the OPERATOR boundary (B1) as an un-reloaded reference pickle drives it: grouped (per-sample-weight) convolutions
in one launch, the per-group weight re-tile, the weight cache — against ATen on the CPU — and the whole generator forward through
`oracle/b1_route.py` (the reference's code pattern on next3d_amd.torch_utils.ops + next3d_amd.shims) against the goldens the REAL
reference produced (tests/golden/case_*.npz): <= 1e-3 max-abs on rendered RGB (BASELINE.json north_star)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from next3d_amd import mesh, spec
from oracle import cases

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def _gen(shape, seed, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _md(a, b):
    return float((torch.as_tensor(a).float().cpu() - torch.as_tensor(b).float().cpu()).abs().max())


@pytest.mark.parametrize('kind', [0, 1, 2])
@pytest.mark.parametrize('G,O,I,k,transposed', [(4, 512, 512, 3, False), (3, 96, 64, 3, True), (2, 3, 128, 1, False), (1, 40, 48, 3, False),
                                                (5, 64, 16, 3, True)])
def test_prep_weight_grouped_equals_per_group_preparation(dev, kind, G, O, I, k, transposed):
    """n3d_conv2d_prep_weight_grouped (all groups in one launch, F.conv2d or F.conv_transpose2d weight layout, float32 or float16 in)
    writes, group by group, exactly what the per-model preparation kernels write for that group's [O, I, k, k] weights."""
    from next3d_amd import _lib
    from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
    if kind == 2 and k != 3:
        pytest.skip('float16 tiles are 3x3')
    w = _gen((G, O, I, k, k), 5, 0.1)                                      # reference order per group
    wd = (w.transpose(1, 2).reshape(G * I, O, k, k) if transposed else w.reshape(G * O, I, k, k)).contiguous().to(dev)
    for dt in (torch.float32, torch.float16):
        src = wd.to(dt)
        out, per, o, i = cg.prep_weight_grouped(src, G, transposed, kind)
        assert (o, i) == (O, I)
        wq = src.float().reshape(G, I, O, k, k).transpose(1, 2) if transposed else src.float().reshape(G, O, I, k, k)
        for g in range(G):
            grp = out.view(torch.uint8)[g * per:(g + 1) * per]
            if kind == 0:
                ref = cg.prep_weight(wq[g].contiguous())
            elif kind == 1:
                ref = cg.prep_weight_bf16x3(wq[g].contiguous())
            else:                                                       # [9][I/16][2][O][8] float16 (sr_f16.hip modulate_row's layout)
                ref = wq[g].permute(2, 3, 1, 0).reshape(9, I // 16, 2, 8, O).permute(0, 1, 2, 4, 3).contiguous().to(torch.float16)
            assert torch.equal(grp, ref.reshape(-1).view(torch.uint8)), (kind, dt, g)


CONV_CASES = [  # N, I, O, H, W, k, mode
    (4, 512, 512, 8, 8, 3, 0), (4, 256, 128, 64, 64, 3, 0), (2, 64, 96, 33, 47, 3, 0), (3, 512, 3, 32, 32, 1, 0), (4, 128, 3, 128, 128, 1, 0),
    (4, 512, 512, 16, 16, 3, 2), (2, 128, 64, 64, 64, 3, 2), (1, 32, 256, 128, 128, 3, 2), (3, 48, 40, 20, 36, 3, 2), (1, 512, 512, 4, 4, 3, 0),
]


@pytest.mark.parametrize('precision', ['bf16x3', 'fp32'])
@pytest.mark.parametrize('N,I,O,H,W,k,mode', CONV_CASES)
def test_grouped_conv_one_launch_matches_aten(dev, precision, N, I, O, H, W, k, mode):
    """conv2d_gradfix.conv2d / conv_transpose2d with groups = batch on per-sample weights — the call the FUSED modulated convolution
    makes (networks_stylegan2.py:82-88) — against ATen's grouped convolution on the CPU.  One launch (+ one re-tile launch)."""
    from next3d_amd import _lib, layers
    from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
    layers.set_precision(precision)
    try:
        x = _gen((1, N * I, H, W), 1)
        w = _gen((N * O, I, k, k), 2) / np.sqrt(I * k * k)
        if mode == 2:
            wt = w.reshape(N, O, I, k, k).transpose(1, 2).reshape(N * I, O, k, k).contiguous()
            ref = F.conv_transpose2d(x, wt, stride=2, groups=N)
            _lib.prof_enable(True); _lib.prof_reset()
            y = cg.conv_transpose2d(x.to(dev), wt.to(dev), stride=2, groups=N)
        else:
            ref = F.conv2d(x, w, padding=k // 2, groups=N)
            _lib.prof_enable(True); _lib.prof_reset()
            y = cg.conv2d(x.to(dev), w.to(dev), padding=k // 2, groups=N)
        prof = _lib.prof_read()
        _lib.prof_enable(False)
        launches = sum(prof[f]['launches'] for f in ('conv2d', 'conv2d_bf16x3', 'conv1x1_bf16x3'))
        assert launches == 1, prof                                       # the whole batch in ONE convolution launch
        assert tuple(y.shape) == tuple(ref.shape)
        tol = 2e-5 if precision == 'fp32' else 1e-4
        err = _md(y, ref)
        print(f'grouped {precision} N{N} {I}->{O} {H}x{W} k{k} mode{mode}: max abs {err:.2e} (|ref| max {float(ref.abs().max()):.2f})')
        assert err <= tol * max(1.0, float(ref.abs().max()))
    finally:
        layers.set_precision('bf16x3')


@pytest.mark.parametrize('N,I,O,H,W,k,mode', [(4, 32, 256, 128, 128, 3, 2), (2, 256, 256, 64, 64, 3, 0), (4, 128, 3, 64, 64, 1, 0), (2, 64, 64, 16, 32, 3, 0),
                                              (3, 48, 40, 20, 36, 3, 0)])
def test_grouped_conv_float16_matches_aten_half_semantics(dev, N, I, O, H, W, k, mode):
    """float16 operands (a reference fp16 block): the result is the float16 rounding of the float32-accumulated sum of exact
    float16 x float16 products — compared with ATen on the CPU evaluating the same float16 values in float32, rounded once."""
    from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
    x = _gen((1, N * I, H, W), 3).half()
    w = (_gen((N * O, I, k, k), 4) / np.sqrt(I * k * k)).half()
    if mode == 2:
        wt = w.reshape(N, O, I, k, k).transpose(1, 2).reshape(N * I, O, k, k).contiguous()
        ref = F.conv_transpose2d(x.float(), wt.float(), stride=2, groups=N)
        y = cg.conv_transpose2d(x.to(dev), wt.to(dev), stride=2, groups=N)
    else:
        ref = F.conv2d(x.float(), w.float(), padding=k // 2, groups=N)
        y = cg.conv2d(x.to(dev), w.to(dev), padding=k // 2, groups=N)
    assert y.dtype == torch.float16 and tuple(y.shape) == tuple(ref.shape)
    d = (y.float().cpu() - ref).abs()
    ulp = float(ref.abs().max()) * 2.0 ** -10
    print(f'grouped f16 N{N} {I}->{O} {H}x{W} k{k} mode{mode}: max {float(d.max()):.2e} mean {float(d.mean()):.2e} (one float16 ulp of the largest value {ulp:.2e})')
    assert float(d.max()) <= 1.01 * ulp                                  # one rounding of a sum that differs only in accumulation order


def test_prepared_weight_cache_follows_the_tensor_object(dev):
    """groups == 1: a persistent weight tensor is re-tiled once, an in-place update re-tiles it, and a NEW tensor at a recycled
    address (Conv2dLayer's `self.weight * self.weight_gain` temporaries) never gets another tensor's tiles."""
    from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
    x = _gen((2, 32, 24, 40), 7).to(dev)
    w = (_gen((48, 32, 3, 3), 8) / 17).to(dev)
    y0 = cg.conv2d(x, w, padding=1)
    assert id(w) in cg._PREP_CACHE
    y1 = cg.conv2d(x, w, padding=1)
    assert torch.equal(y0, y1)
    w.mul_(2.0)
    assert _md(cg.conv2d(x, w, padding=1), 2 * y0) <= 1e-5 * float(y0.abs().max())
    outs = []
    for s in (0.5, 3.0, -1.0):                                            # temporaries of equal shape: the allocator hands out the same block again
        outs.append(cg.conv2d(x, w * s, padding=1))
    for s, o in zip((0.5, 3.0, -1.0), outs):
        assert _md(o, 2 * s * y0) <= 2e-5 * float(y0.abs().max()) * abs(s)
    ref = F.conv2d(x.cpu(), w.cpu(), padding=1)
    assert _md(cg.conv2d(x, w, padding=1), ref) <= 1e-4 * float(ref.abs().max())


@pytest.fixture(scope='module')
def route(dev):
    from oracle import b1_route
    d = np.load(os.path.join(GOLDEN, 'demo_inputs.npz'))
    P = spec.synthetic_state_dict(0)
    P.update(mesh.mesh_buffers(d['faces'], d['uvs'], d['uvfaces']))
    r = b1_route.Route(dev)
    return r, r.to_device(P), torch.nn.functional.interpolate(mesh.synthetic_uv_face_mask().float(), [256, 256])


@pytest.mark.parametrize('precision', ['bf16x3', 'fp32'])
@pytest.mark.parametrize('case', ['case_r32_s24', 'case_r64_s48_b4'])
def test_b1_route_matches_reference_golden(route, dev, case, precision):
    """The reference's code pattern on the operator layer + shims — what an un-reloaded pickle runs — against the reference's own
    outputs, float32 route (force_fp32=True: what the goldens pin)."""
    from next3d_amd import layers
    r, P, mask = route
    layers.set_precision(precision)
    try:
        d = np.load(os.path.join(GOLDEN, case + '.npz'))
        N, R, Sc, Sf = d['z'].shape[0], int(d['R']), int(d['Sc']), int(d['Sf'])
        rk = dict(depth_resolution=Sc, depth_resolution_importance=Sf, ray_start=2.25, ray_end=3.3, box_warp=1, c_scale=1.0, c_gen_conditioning_zero=True)
        jitter, u = cases.rng_inputs(N, R, Sc, Sf)
        with torch.no_grad():
            ws = r.mapping(P, torch.from_numpy(d['z']).float(), torch.from_numpy(d['c_cond']), rk, truncation_psi=float(d['psi']), truncation_cutoff=int(d['cutoff']))
            out, st = r.synthesis(P, ws, torch.from_numpy(d['c']), torch.from_numpy(d['v']), mask, rk, jitter, u, neural_rendering_resolution=R,
                                  return_stages=True, force_fp32=True)
        rep = {'ws': _md(ws, d['ws']), 'mouth_mask': _md(st['mouth_mask'], d['mouth_mask']),
               'alpha': _md((st['alpha'] * 255).round(), d['alpha'].astype(np.float32)),
               'textures': _md(st['textures'][..., ::8, ::8], d['textures_sub8']), 'static_plane': _md(st['static_plane'].reshape(N, 96, 256, 256)[..., ::8, ::8], d['static_plane_sub8']),
               'image_raw': _md(out['image_raw'], d['image_raw']), 'image_depth': _md(out['image_depth'], d['image_depth']),
               'image': _md(out['image'][..., ::4, ::4], d['image_sub4'])}
        print('B1 route', case, precision, ' '.join(f'{k}={v:.3e}' for k, v in rep.items()))
        assert rep['ws'] <= 1e-4 and rep['mouth_mask'] == 0 and rep['alpha'] == 0
        assert rep['textures'] <= 1e-3 and rep['static_plane'] <= 1e-3
        assert rep['image_raw'] <= 1e-3 and rep['image'] <= 1e-3 and rep['image_depth'] <= 1e-3, rep      # north_star
    finally:
        layers.set_precision('bf16x3')


@pytest.mark.parametrize('case', ['case_r32_s24', 'case_r64_s48_b4'])
def test_b1_default_route_float16_blocks_against_reference_fp16_run(route, dev, case):
    """The scripts' default call (no force_fp32): the super-resolution blocks as REAL float16 tensors through the operator layer —
    f16 matrix-core convolutions, float16 FIR / bias_act — against the reference's own float16 run (tests/golden/*_fp16sr.npz), on
    the reference's (rgb, features, ws).  Tolerance: tests/test_cpu_oracle.py::FP16_SR_TOL['cuda'] (bias_act rounds once here, as
    bias_act.cu does; the fixture was produced with _bias_act_ref's per-step rounding)."""
    from test_cpu_oracle import FP16_SR_TOL
    r, P, _ = route
    g = np.load(os.path.join(GOLDEN, case + '_fp16sr.npz'))
    ref = torch.from_numpy(g['image'])
    step = int(g['image_step']) if 'image_step' in g else 1
    rgb, feat, ws = (torch.from_numpy(g[k]).to(dev) for k in ('rgb_in', 'feat_in', 'ws_in'))
    with torch.no_grad(), torch.device(dev):
        out = r.networks.superresolution(P, 'superresolution', rgb, feat, ws, force_fp32=False).cpu()[..., ::step, ::step]
    d = (out - ref).abs()
    print('B1 float16 blocks', case, f'max {float(d.max()):.3e} mean {float(d.mean()):.3e}')
    assert float(d.max()) <= FP16_SR_TOL['cuda'][0] and float(d.mean()) <= FP16_SR_TOL['cuda'][1]
