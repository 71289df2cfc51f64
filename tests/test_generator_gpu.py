"""-m gpu parity of the whole generator forward (mapping + synthesis) through libn3d.so against
  (a) the golden fixtures produced by the REAL reference (tests/golden/case_*.npz, oracle/pin_against_reference.py),
  (b) the CPU oracle, stage by stage.
Tolerance on rendered RGB: 1e-3 max-abs (BASELINE.json north_star)."""
import os

import numpy as np
import pytest
import torch

from next3d_amd import mesh, spec
from oracle import cases

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
RK = dict(image_resolution=512, disparity_space_sampling=False, clamp_mode='softplus', c_gen_conditioning_zero=True,
          c_scale=1.0, superresolution_noise_mode='none', decoder_lr_mul=1.0, sr_antialias=True, depth_resolution=48,
          depth_resolution_importance=48, ray_start=2.25, ray_end=3.3, box_warp=1, avg_camera_radius=2.7,
          avg_camera_pivot=[0, 0, 0.2])


@pytest.fixture(scope='module')
def G(dev):
    from next3d_amd.generator import TriPlaneGenerator
    d = np.load(os.path.join(GOLDEN, 'demo_inputs.npz'))
    g = TriPlaneGenerator(512, 25, 512, 512, 3, (d['faces'], d['uvs'], d['uvfaces']), sr_num_fp16_res=4,
                          mapping_kwargs=dict(num_layers=2), rendering_kwargs=dict(RK),
                          sr_kwargs=dict(channel_base=32768, channel_max=512, fused_modconv_default='inference_only'),
                          uv_face_mask=mesh.synthetic_uv_face_mask(), channel_base=32768, channel_max=512,
                          fused_modconv_default='inference_only', num_fp16_res=0, conv_clamp=None)
    sd = spec.synthetic_state_dict(0)
    sd.update(mesh.mesh_buffers(d['faces'], d['uvs'], d['uvfaces']))
    g.load_state_dict(sd, strict=True)
    return g.eval().requires_grad_(False).to(dev)


def _md(a, b):
    return float((torch.as_tensor(a).float().cpu() - torch.as_tensor(b).float().cpu()).abs().max())


@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
@pytest.mark.parametrize('case', ['case_r32_s24', 'case_r64_s48', 'case_r64_s96', 'case_r64_s48_b4', 'case_r64_s48_b8', 'case_r64_s96_v4'])
def test_forward_matches_reference_golden(G, dev, case, precision):
    from next3d_amd import layers
    layers.set_precision(precision)
    d = np.load(os.path.join(GOLDEN, case + '.npz'))
    N, R, Sc, Sf = d['z'].shape[0], int(d['R']), int(d['Sc']), int(d['Sf'])
    G.rendering_kwargs['depth_resolution'], G.rendering_kwargs['depth_resolution_importance'] = Sc, Sf
    jitter, u = cases.rng_inputs(N, R, Sc, Sf)
    G.keep_stages = True
    ws = G.mapping(torch.from_numpy(d['z']).to(dev), torch.from_numpy(d['c_cond']).to(dev), truncation_psi=float(d['psi']),
                   truncation_cutoff=int(d['cutoff']))
    out = G.synthesis(ws, torch.from_numpy(d['c']).to(dev), torch.from_numpy(d['v']).to(dev), neural_rendering_resolution=R,
                      noise_mode='const', depth_jitter=jitter, importance_u=u, force_fp32=True)      # the goldens pin the fp32 path
    st = G._debug
    rep = {
        'ws': _md(ws, d['ws']),
        'mouth_mask': _md(st['bbox'], d['mouth_mask']),
        'alpha': _md((st['alpha'] * 255).round(), d['alpha'].astype(np.float32)),
        'textures': _md(st['textures'][..., ::8, ::8], d['textures_sub8']),
        'mouths_plane': _md(st['mouths'][..., ::8, ::8], d['mouths_plane_sub8']) if 'mouths_plane_sub8' in d else 0.0,      # (the lean batch-8 fixture
        'rendering_stitch': _md(st['stitch'][..., ::8, ::8], d['rendering_stitch_sub8']) if 'rendering_stitch_sub8' in d else 0.0,   # omits these two)
        'static_plane': _md(st['static'][..., ::8, ::8], d['static_plane_sub8']),
        'image_raw': _md(out['image_raw'], d['image_raw']),
        'image_depth': _md(out['image_depth'], d['image_depth']),
        'image': _md(out['image'][..., ::4, ::4], d['image_sub4']),
        'image_mean': _md(out['image'].mean(dim=(2, 3)), d['image_mean']),
    }
    print(case, precision, ' '.join(f'{k}={v:.3e}' for k, v in rep.items()))
    n_alpha_bad = int(((st['alpha'].cpu() * 255).round() != torch.from_numpy(d['alpha'].astype(np.float32))).sum())
    print('alpha pixels differing:', n_alpha_bad)
    assert rep['ws'] <= 1e-4
    assert rep['mouth_mask'] == 0
    assert n_alpha_bad == 0                       # index work: bit-exact on the committed goldens (north_star)
    layers.set_precision('bf16x3')
    assert rep['textures'] <= 1e-3 and rep['static_plane'] <= 1e-3
    assert rep['image_raw'] <= 1e-3, rep           # north_star: <= 1e-3 max-abs on rendered RGB
    assert rep['image'] <= 1e-3, rep
    assert rep['image_depth'] <= 1e-3, rep


@pytest.mark.parametrize('case', ['case_r32_s24', 'case_r64_s48', 'case_r64_s48_b4'])
def test_fp16_superresolution_matches_reference_fp16_run(G, dev, case, monkeypatch):
    """The scripts' DEFAULT route (sr_num_fp16_res = 4, no force_fp32) on the f16 kernels against the REFERENCE's own float16
    run of its super-resolution blocks (tests/golden/*_fp16sr.npz, oracle/pin_against_reference.py --fp16), on the reference's
    own (rgb, features, ws).  Two bias_act roundings (layers.F16_REF_CPU_ROUNDING): the reference's off-GPU one reproduces that
    CPU run up to accumulation order; the default (bias_act.cu's single rounding) within the stated looser bound.  Tolerances and
    their derivation: tests/test_cpu_oracle.py::FP16_SR_TOL."""
    from next3d_amd import generator, layers
    from test_cpu_oracle import FP16_SR_TOL
    g = np.load(os.path.join(GOLDEN, case + '_fp16sr.npz'))
    ref = torch.from_numpy(g['image'])
    step = int(g['image_step']) if 'image_step' in g else 1          # the batch-4 fixture (the benched configuration) keeps every second pixel
    rgb, feat, ws = (torch.from_numpy(g[k]).to(dev) for k in ('rgb_in', 'feat_in', 'ws_in'))
    sr = G._prep().sr
    for mode in ('cpu', 'cuda'):
        monkeypatch.setattr(layers, 'F16_REF_CPU_ROUNDING', mode == 'cpu')
        assert sr._f16_ok(generator._resize_aa(feat, 128), 'none')
        out = sr(rgb, feat, ws, generator._resize_aa, noise_mode='none', fp16=True).cpu()[..., ::step, ::step]
        d = (out - ref).abs()
        print(case, mode, f'max {float(d.max()):.3e} mean {float(d.mean()):.3e} (image absmax {float(ref.abs().max()):.2f})')
        assert float(d.max()) <= FP16_SR_TOL[mode][0] and float(d.mean()) <= FP16_SR_TOL[mode][1]


@pytest.mark.parametrize('case', ['case_r32_s24', 'case_r64_s48', 'case_r64_s48_b4'])
def test_default_route_end_to_end_against_reference_fp16_run(G, dev, case):
    """The whole forward exactly as the scripts call it (no force_fp32) against the reference's float16-route image: the
    renderer's float32 output differs from the reference's by ~1e-5, which float16 rounding amplifies to single ulps."""
    from next3d_amd import layers
    from test_cpu_oracle import FP16_SR_TOL
    layers.set_precision('bf16x3')
    d = np.load(os.path.join(GOLDEN, case + '.npz'))
    g16 = np.load(os.path.join(GOLDEN, case + '_fp16sr.npz'))
    ref, step = torch.from_numpy(g16['image']), (int(g16['image_step']) if 'image_step' in g16 else 1)
    N, R, Sc, Sf = d['z'].shape[0], int(d['R']), int(d['Sc']), int(d['Sf'])
    G.rendering_kwargs['depth_resolution'], G.rendering_kwargs['depth_resolution_importance'] = Sc, Sf
    jitter, u = cases.rng_inputs(N, R, Sc, Sf)
    ws = G.mapping(torch.from_numpy(d['z']).to(dev), torch.from_numpy(d['c_cond']).to(dev), truncation_psi=float(d['psi']), truncation_cutoff=int(d['cutoff']))
    out = G.synthesis(ws, torch.from_numpy(d['c']).to(dev), torch.from_numpy(d['v']).to(dev), neural_rendering_resolution=R, noise_mode='const',
                      depth_jitter=jitter, importance_u=u)['image'].cpu()[..., ::step, ::step]
    e = (out - ref).abs()
    print(case, f'default route vs reference fp16 run: max {float(e.max()):.3e} mean {float(e.mean()):.3e}')
    assert float(e.max()) <= FP16_SR_TOL['cuda'][0] and float(e.mean()) <= FP16_SR_TOL['cuda'][1]


def test_benched_case_matches_reference_at_every_pixel(G, dev):
    """BASELINE.json configs[1] exactly as bench.py runs it (batch 4): all 4 x 3 x 512 x 512 pixels of the reference's image
    (tests/golden/case_r64_s48_b4_image.npz), 1e-3 max-abs, both arithmetic settings."""
    from next3d_amd import layers
    d = np.load(os.path.join(GOLDEN, 'case_r64_s48_b4.npz'))
    ref = torch.from_numpy(np.load(os.path.join(GOLDEN, 'case_r64_s48_b4_image.npz'))['image'])
    N, R, Sc, Sf = d['z'].shape[0], int(d['R']), int(d['Sc']), int(d['Sf'])
    G.rendering_kwargs['depth_resolution'], G.rendering_kwargs['depth_resolution_importance'] = Sc, Sf
    jitter, u = cases.rng_inputs(N, R, Sc, Sf)
    try:
        for precision in ('fp32', 'bf16x3'):
            layers.set_precision(precision)
            ws = G.mapping(torch.from_numpy(d['z']).to(dev), torch.from_numpy(d['c_cond']).to(dev), truncation_psi=float(d['psi']), truncation_cutoff=int(d['cutoff']))
            out = G.synthesis(ws, torch.from_numpy(d['c']).to(dev), torch.from_numpy(d['v']).to(dev), neural_rendering_resolution=R, noise_mode='const',
                              depth_jitter=jitter, importance_u=u, force_fp32=True)['image'].cpu()
            err = float((out - ref).abs().max())
            print(f'{precision}: max-abs over {ref.numel()} pixels {err:.3e}')
            assert err <= 1e-3
    finally:
        layers.set_precision('bf16x3')


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['case_r32_s24', 'case_r64_s48', 'case_r64_s96'])
def test_sample_mixed_matches_reference_golden(G, dev, case):
    """G.sample / G.sample_mixed (shape-extraction point queries, triplane_next3d.py:232-322) against the reference's own
    outputs; the planes are cached on the first call and re-used by the second (the reference rebuilds them per chunk)."""
    d = np.load(os.path.join(GOLDEN, case + '.npz'))
    ws = G.mapping(torch.from_numpy(d['z']).to(dev), torch.from_numpy(d['c_cond']).to(dev), truncation_psi=float(d['psi']),
                   truncation_cutoff=int(d['cutoff']))
    coords = torch.from_numpy(d['sample_coords']).to(dev)
    v = torch.from_numpy(d['v']).to(dev)
    out = G.sample_mixed(coords, torch.zeros_like(coords), ws, v, noise_mode='const', cache_backbone=True)
    e_rgb, e_sig = _md(out['rgb'], d['sample_rgb']), _md(out['sigma'], d['sample_sigma'])
    print(case, 'sample_mixed rgb', e_rgb, 'sigma', e_sig, 'sigma range', float(d['sample_sigma'].min()), float(d['sample_sigma'].max()))
    assert out['rgb'].shape == d['sample_rgb'].shape and out['sigma'].shape == d['sample_sigma'].shape
    assert e_rgb <= 1e-3 and e_sig <= 1e-3 * max(1.0, float(np.abs(d['sample_sigma']).max()))
    half = coords.shape[1] // 2
    out2 = G.sample_mixed(coords[:, half:].contiguous(), None, ws, v, noise_mode='const', use_cached_backbone=True)
    assert torch.equal(out2['rgb'], out['rgb'][:, half:]) and torch.equal(out2['sigma'], out['sigma'][:, half:])
    out3 = G.sample(coords, None, torch.from_numpy(d['z']).to(dev), torch.from_numpy(d['c_cond']).to(dev), v,
                    truncation_psi=float(d['psi']), truncation_cutoff=int(d['cutoff']), noise_mode='const')
    assert _md(out3['rgb'], d['sample_rgb']) <= 1e-3
    with pytest.raises(RuntimeError):
        G.sample_mixed(coords[:, :, :2], None, ws, v, noise_mode='const')


@pytest.mark.gpu
def test_plane_and_identity_caches(G, dev):
    """cache_backbone / use_cached_backbone (camera orbit: planes re-used) and cache_identity / use_cached_identity
    (reenactment: latent-only networks re-used, new mesh) return exactly what a full forward returns."""
    d = np.load(os.path.join(GOLDEN, 'case_r64_s48.npz'))
    N, R, Sc, Sf = d['z'].shape[0], 32, 24, 24
    G.rendering_kwargs['depth_resolution'], G.rendering_kwargs['depth_resolution_importance'] = Sc, Sf
    jitter, u = cases.rng_inputs(N, R, Sc, Sf)
    t = lambda k: torch.from_numpy(d[k]).to(dev)
    ws = G.mapping(t('z'), t('c_cond'), truncation_psi=0.7, truncation_cutoff=14)
    kw = dict(neural_rendering_resolution=R, noise_mode='const', depth_jitter=jitter, importance_u=u)
    c, v = t('c'), t('v')
    full = G.synthesis(ws, c, v, cache_backbone=True, cache_identity=True, **kw)['image']
    c2 = c.clone(); c2[:, [3, 7]] += 0.05                                            # another camera, same planes
    orbit = G.synthesis(ws, c2, v, use_cached_backbone=True, **kw)['image']
    assert torch.equal(orbit, G.synthesis(ws, c2, v, **kw)['image'])
    v2 = v.clone(); v2[:, :5023, 1] += 0.002                                         # another mesh, same identity
    reenact = G.synthesis(ws, c, v2, use_cached_identity=True, **kw)['image']
    ref2 = G.synthesis(ws, c, v2, **kw)['image']
    assert torch.equal(reenact, ref2)
    assert not torch.equal(ref2, full)


@pytest.mark.gpu
def test_pipelined_steps_are_bitwise_reproducible(G, dev):
    """Six forwards issued back to back WITHOUT host synchronisation (the bench / video-loop pattern, static backbone on its
    side stream) must all return the same bits.  Guards the stream choreography: the rasteriser used to return different
    faces from run to run while 8-wave split-bf16 conv kernels of another stream were resident — bisected to L1-served gather
    loads of its vertex / face tables, which are agent-scope loads now."""
    d = np.load(os.path.join(GOLDEN, 'case_r64_s48.npz'))
    N, R, Sc, Sf = d['z'].shape[0], 32, 24, 24
    G.rendering_kwargs['depth_resolution'], G.rendering_kwargs['depth_resolution_importance'] = Sc, Sf
    jitter, u = cases.rng_inputs(N, R, Sc, Sf)
    t = lambda k: torch.from_numpy(d[k]).to(dev)
    ws = G.mapping(t('z'), t('c_cond'), truncation_psi=0.7, truncation_cutoff=14)
    c, v = t('c'), t('v')
    outs = [G.synthesis(ws, c, v, neural_rendering_resolution=R, noise_mode='const', depth_jitter=jitter, importance_u=u, force_fp32=True)
            for _ in range(6)]
    torch.cuda.synchronize()
    for o in outs[1:]:
        for k in ('image', 'image_raw', 'image_depth'):
            assert torch.equal(o[k], outs[0][k]), k


@pytest.mark.gpu
def test_fused_backbone_torgb_equals_separate_torgb(G, dev, monkeypatch):
    """Round 5 (layers.FUSED_TORGB_MAX = 32): the 32-colour toRGB layers of the texture / mouth / blending networks evaluated in their conv1's
    epilogue on the matrix cores (blocks whose feature map has no float32 reader: n3d_conv2d_desc.rgb_*, partial colours + n3d_rgb_combine)
    against the same forward with separate 1x1 toRGB launches (FUSED_TORGB_MAX = 4).  Both forms are split-bf16 products with float32
    accumulation in different orders: the stage tensors agree to the 1x1 kernel's own accuracy, the rasterised geometry bit for bit."""
    from next3d_amd import layers
    layers.set_precision('bf16x3')
    d = np.load(os.path.join(GOLDEN, 'case_r64_s48_b4.npz'))
    R, Sc, Sf = 64, int(d['Sc']), int(d['Sf'])
    G.rendering_kwargs['depth_resolution'], G.rendering_kwargs['depth_resolution_importance'] = Sc, Sf
    jitter, u = cases.rng_inputs(4, R, Sc, Sf)
    t = lambda k: torch.from_numpy(d[k]).to(dev)
    ws = G.mapping(t('z'), t('c_cond'), truncation_psi=0.7, truncation_cutoff=14)
    kw = dict(neural_rendering_resolution=R, noise_mode='const', depth_jitter=jitter, importance_u=u, force_fp32=True)
    G.keep_stages = True
    got = {}
    try:
        for mx in (32, 4):
            monkeypatch.setattr(layers, 'FUSED_TORGB_MAX', mx)
            o = G.synthesis(ws, t('c'), t('v'), **kw)
            got[mx] = dict({k: G._debug[k].clone() for k in ('textures', 'mouths', 'stitch', 'static', 'alpha', 'bbox')}, image=o['image'].clone(), image_raw=o['image_raw'].clone())
    finally:
        G.keep_stages = False
    assert torch.equal(got[32]['alpha'], got[4]['alpha']) and torch.equal(got[32]['bbox'], got[4]['bbox']) and torch.equal(got[32]['static'], got[4]['static'])
    rep = {k: _md(got[32][k], got[4][k]) for k in ('textures', 'mouths', 'stitch', 'image_raw', 'image')}
    print('fused vs separate 32-colour toRGB:', ' '.join(f'{k}={v:.3e}' for k, v in rep.items()), '| stage |max|', ' '.join(f"{k}={float(got[4][k].abs().max()):.1f}" for k in ('textures', 'mouths', 'stitch')))
    assert rep['textures'] > 0                                              # the fused layers do run
    for k in ('textures', 'mouths', 'stitch'):
        assert rep[k] <= 2e-5 * max(1.0, float(got[4][k].abs().max())), (k, rep[k])
    assert rep['image_raw'] <= 1e-4 and rep['image'] <= 1e-4


@pytest.mark.gpu
def test_fused_last_layer_torgb_equals_separate_torgb(G, dev, monkeypatch):
    """layers.FUSED_TORGB: the super-resolution's last 3x3 layer (128 channels at 512 x 512) evaluates its toRGB in the epilogue and never
    writes its feature map.  Against the same forward with the two layers separate: equal to the accuracy of the separate 1x1 kernel
    (split-bf16: 2^-17 relative per product); everything in front of that layer is bit-identical.  (FUSED_TORGB_MAX = 4 here: the super-resolution's
    3-colour layers only, as in round 4; the backbones' 32-colour layers: test_fused_backbone_torgb_equals_separate_torgb.)"""
    from next3d_amd import layers
    layers.set_precision('bf16x3')
    monkeypatch.setattr(layers, 'FUSED_TORGB_MAX', 4)
    d = np.load(os.path.join(GOLDEN, 'case_r64_s48_b4.npz'))
    R, Sc, Sf = 64, int(d['Sc']), int(d['Sf'])
    G.rendering_kwargs['depth_resolution'], G.rendering_kwargs['depth_resolution_importance'] = Sc, Sf
    jitter, u = cases.rng_inputs(4, R, Sc, Sf)
    t = lambda k: torch.from_numpy(d[k]).to(dev)
    ws = G.mapping(t('z'), t('c_cond'), truncation_psi=0.7, truncation_cutoff=14)
    kw = dict(neural_rendering_resolution=R, noise_mode='const', depth_jitter=jitter, importance_u=u, force_fp32=True)
    outs = {}
    for on in (True, False):
        monkeypatch.setattr(layers, 'FUSED_TORGB', on)
        o = G.synthesis(ws, t('c'), t('v'), **kw)
        outs[on] = {k: o[k].clone() for k in ('image', 'image_raw', 'image_depth')}
    assert torch.equal(outs[True]['image_raw'], outs[False]['image_raw']) and torch.equal(outs[True]['image_depth'], outs[False]['image_depth'])
    e = _md(outs[True]['image'], outs[False]['image'])
    print(f'fused vs separate toRGB of the last layer: image max abs diff {e:.3e}')
    assert 0 < e <= 5e-5
    # the scripts' default route (float16 blocks): the same fusion in n3d_conv2d_f16 — float16 operands, float32 sums in another order: colours equal
    # up to one float16 ulp (2^-8 at |v| in [4, 8)) where a sum lands on a rounding boundary
    kw.pop('force_fp32')
    h = {}
    for on in (True, False):
        monkeypatch.setattr(layers, 'FUSED_TORGB', on)
        h[on] = G.synthesis(ws, t('c'), t('v'), **kw)['image'].clone()
    dh = (h[True] - h[False]).abs()
    print(f'float16 route, fused vs separate toRGB: max abs diff {float(dh.max()):.3e}, differing {float((dh > 1e-6).float().mean()):.2e} of the values')
    assert float(dh.max()) <= 2.0 ** -7 and float((dh > 1e-6).float().mean()) < 0.02


@pytest.mark.gpu
@pytest.mark.parametrize('batch,R', [(3, 64), (1, 128)])
def test_presplit_pipeline_equals_register_staged_pipeline(G, dev, monkeypatch, batch, R):
    """The whole forward with every pre-split (LDS-DMA) path switched off — register-staged kernels, float32 NCHW between all
    layers — against the default pipeline, at a batch size / render resolution the goldens do not cover.  The pre-split kernels
    multiply bit-identical operands in the same order; the only differences allowed are the FIR variants' accumulation order
    (c8 / NCHW -> split8 against the float32 FIR), hence a tolerance far below the parity tolerance instead of equality.
    The reference's default float16 super-resolution mode is run too (finite, close to the float32 mode)."""
    from next3d_amd import layers
    d = np.load(os.path.join(GOLDEN, 'case_r64_s48_b4.npz'))
    Sc, Sf = 24, 24
    G.rendering_kwargs['depth_resolution'], G.rendering_kwargs['depth_resolution_importance'] = Sc, Sf
    jitter, u = cases.rng_inputs(batch, R, Sc, Sf)
    t = lambda k: torch.from_numpy(d[k][:batch]).to(dev)
    ws = G.mapping(t('z'), t('c_cond'), truncation_psi=0.7, truncation_cutoff=14)
    kw = dict(neural_rendering_resolution=R, noise_mode='const', depth_jitter=jitter, importance_u=u)
    out = {}
    for on in (True, False):
        for name in ('PRESPLIT', 'UP_PRESPLIT', 'S2_PRESPLIT'):
            monkeypatch.setattr(layers, name, on)
        out[on] = G.synthesis(ws, t('c'), t('v'), force_fp32=True, **kw)
    for k in ('image', 'image_raw', 'image_depth'):
        e = _md(out[True][k], out[False][k])
        print(f'batch {batch} render {R} {k}: pre-split vs register-staged max abs diff {e:.3e}')
        assert e <= 2e-4, (k, e)              # (measured 4e-5: rounding-order differences of the FIR variants, amplified by ~30 layers)
    half_off = G.synthesis(ws, t('c'), t('v'), **kw)      # the default call (float16 super-resolution blocks) runs under every layout switch
    for name in ('PRESPLIT', 'UP_PRESPLIT', 'S2_PRESPLIT'):
        monkeypatch.setattr(layers, name, True)
    half = G.synthesis(ws, t('c'), t('v'), **kw)                          # the reference's default: float16 super-resolution blocks
    e, e_off = _md(half['image'], out[True]['image']), _md(half_off['image'], half['image'])
    print(f'batch {batch} render {R}: fp16-mode image vs fp32-mode image max abs diff {e:.3e}; fp16 mode with / without the pre-split hand-off {e_off:.3e}')
    assert torch.isfinite(half['image']).all() and e <= 2e-2 and e_off <= 2e-2


@pytest.mark.gpu
def test_packed_mesh_sequence_streams_to_device(dev, tmp_path):
    """meshio.MeshSequence.batches: pinned double-buffered uploads on a copy stream return the packed frames unchanged."""
    from next3d_amd import meshio
    d = np.load(os.path.join(GOLDEN, 'demo_inputs.npz'))
    frames = np.stack([np.concatenate([d['verts'] + 0.001 * t, d['landmarks'] - 0.002 * t], 0) for t in range(7)], 0).astype('<f4')
    path = str(tmp_path / 'seq.n3dmesh')
    with open(path, 'wb') as fh:
        fh.write(meshio.HEADER.pack(meshio.MAGIC, 7, 5023, 68, 0)); fh.write(frames.tobytes())
    seq = meshio.MeshSequence(path)
    got = [b.clone() for b in seq.batches(3, dev)]
    assert [g.shape[0] for g in got] == [3, 3, 1] and all(g.is_cuda for g in got)
    assert torch.equal(torch.cat(got, 0).cpu(), torch.from_numpy(frames))


@pytest.mark.gpu
@pytest.mark.timeout(600)
def test_bench_multi_gpu_path_on_one_gpu():
    """The N > 1 path of bench.py — `torch.distributed.run` launch line of the driver, RCCL ('nccl') process group, the per-step
    asynchronous uint8 frame gather on RCCL's stream, barrier + max-over-ranks timing — executed on hardware with world size 1
    (N3D_BENCH_FORCE_DIST=1): the only thing a 1-GPU box cannot show is the scaling number itself."""
    import json
    import socket
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, N3D_BENCH_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1', '--master-port', str(port),
           os.path.join(repo, 'bench.py'), '--gpus', '1', '--steps', '4', '--warmup', '2', '--prewarm-seconds', '0.2', '--no-cpu-baseline', '--no-roofline', '--no-extras']
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=repo, timeout=540)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][-1])
    print(line['value'], line['config']['gather'])
    assert line['n_gpus'] == 1 and line['config']['gather'].startswith('RCCL') and line['frames_bitwise_reproducible'] and line['value'] > 20


@pytest.mark.gpu
def test_synthesis_graph_replay_is_bit_identical(G, dev):
    """synthesis_graph (HIP-graph capture + replay) against the eager call on the same inputs: the full forward at
    batch 2, new inputs through the SAME captured graph, the cached camera orbit (use_cached_backbone) and the reenactment pattern
    (use_cached_identity, a new mesh per frame) — identical bits every time."""
    from next3d_amd import layers
    layers.set_precision('bf16x3')
    d = np.load(os.path.join(GOLDEN, 'case_r64_s48.npz'))
    N, R, Sc, Sf = d['z'].shape[0], int(d['R']), int(d['Sc']), int(d['Sf'])
    G.rendering_kwargs['depth_resolution'], G.rendering_kwargs['depth_resolution_importance'] = Sc, Sf
    jitter, u = (t.to(dev) for t in cases.rng_inputs(N, R, Sc, Sf))
    ws = G.mapping(torch.from_numpy(d['z']).to(dev), torch.from_numpy(d['c_cond']).to(dev), truncation_psi=0.7, truncation_cutoff=14)
    c, v = torch.from_numpy(d['c']).to(dev), torch.from_numpy(d['v']).to(dev)
    kw = dict(neural_rendering_resolution=R, noise_mode='const', depth_jitter=jitter, importance_u=u)
    G.refresh()
    for force32 in (True, False):                                     # the float32 route and the default float16 super-resolution route
        eager = {k: t.clone() for k, t in G.synthesis(ws, c, v, force_fp32=force32, **kw).items()}
        out = G.synthesis_graph(ws, c, v, force_fp32=force32, **kw)
        assert all(torch.equal(out[k], eager[k]) for k in eager), [k for k in eager if not torch.equal(out[k], eager[k])]
        c2 = c.flip(0).contiguous()                                   # other cameras, other latents through the same graph
        ws2 = ws.flip(0).contiguous()
        eager2 = G.synthesis(ws2, c2, v, force_fp32=force32, **kw)['image'].clone()
        assert torch.equal(G.synthesis_graph(ws2, c2, v, force_fp32=force32, **kw)['image'], eager2)
        assert torch.equal(G.synthesis_graph(ws, c, v, force_fp32=force32, **kw)['image'], eager['image'])
    assert len(G._graphs) == 2
    # camera orbit on cached planes
    G.synthesis(ws, c, v, cache_backbone=True, force_fp32=True, **kw)
    for cam in (c, c2, c):
        e = G.synthesis(ws, cam, v, use_cached_backbone=True, force_fp32=True, **kw)['image'].clone()
        assert torch.equal(G.synthesis_graph(ws, cam, v, use_cached_backbone=True, force_fp32=True, **kw)['image'], e)
    # reenactment: cached identity, a new mesh per frame
    G.synthesis(ws, c, v, cache_identity=True, force_fp32=True, **kw)
    gq = torch.Generator(device=dev).manual_seed(3)
    for k in range(3):
        vk = v + 2e-4 * torch.randn(v.shape, device=dev, generator=gq)
        e = G.synthesis(ws, c, vk, use_cached_identity=True, force_fp32=True, **kw)['image'].clone()
        assert torch.equal(G.synthesis_graph(ws, c, vk, use_cached_identity=True, force_fp32=True, **kw)['image'], e)
    with pytest.raises(RuntimeError):
        G.synthesis_graph(ws, c, v, cache_backbone=True, **kw)
    G.refresh()
    assert G._graphs is None


@pytest.mark.gpu
@pytest.mark.parametrize('batch', [1, 4])
def test_static_backbone_side_stream_equals_serial(G, dev, batch):
    """The static tri-plane backbone on its side stream (the default) against the two backbones run one after the other on the launch
    stream: the same kernels on the same operands, so every stage is bit-identical."""
    from next3d_amd import layers
    layers.set_precision('bf16x3')
    d = np.load(os.path.join(GOLDEN, 'case_r64_s48_b4.npz'))
    R, Sc, Sf = 64, int(d['Sc']), int(d['Sf'])
    G.rendering_kwargs['depth_resolution'], G.rendering_kwargs['depth_resolution_importance'] = Sc, Sf
    jitter, u = cases.rng_inputs(batch, R, Sc, Sf)
    t = lambda k: torch.from_numpy(d[k][:batch]).to(dev)
    ws = G.mapping(t('z'), t('c_cond'), truncation_psi=0.7, truncation_cutoff=14)
    kw = dict(neural_rendering_resolution=R, noise_mode='const', depth_jitter=jitter, importance_u=u, force_fp32=True)
    G.keep_stages = True
    res = {}
    try:
        for mode, overlap in {'serial': False, 'side_stream': True}.items():
            G.overlap_static = overlap
            out = G.synthesis(ws, t('c'), t('v'), **kw)
            res[mode] = (G._debug['textures'].clone(), G._debug['static'].clone(), out['image'].clone())
    finally:
        G.overlap_static, G.keep_stages = True, False
    for name, a, b in zip(('textures', 'static', 'image'), res['serial'], res['side_stream']):
        assert torch.equal(a, b), (name, _md(a, b))


@pytest.mark.gpu
def test_fused_layout_handovers_are_bit_identical_to_conversion_passes(G, dev, monkeypatch):
    """The two epilogue fusions that replace n3d_split8_from_nchw passes — toRGB writing its input as split8 for the next block's
    transposed convolution (layers.torgb_layer side_style) and the encoders' fromrgb writing split8 for conv1
    (layers.conv2d_layer sole_consumer) — against the same forward with both switched off: the same values reach the same
    kernels, so every output is bit-identical."""
    from next3d_amd import layers
    layers.set_precision('bf16x3')
    d = np.load(os.path.join(GOLDEN, 'case_r64_s48_b4.npz'))
    R, Sc, Sf = 64, int(d['Sc']), int(d['Sf'])
    G.rendering_kwargs['depth_resolution'], G.rendering_kwargs['depth_resolution_importance'] = Sc, Sf
    jitter, u = cases.rng_inputs(4, R, Sc, Sf)
    t = lambda k: torch.from_numpy(d[k]).to(dev)
    ws = G.mapping(t('z'), t('c_cond'), truncation_psi=0.7, truncation_cutoff=14)
    kw = dict(neural_rendering_resolution=R, noise_mode='const', depth_jitter=jitter, importance_u=u, force_fp32=True)
    outs = {}
    for on in (True, False):
        monkeypatch.setattr(layers, 'TORGB_SIDE', on)
        monkeypatch.setattr(layers, 'DIRECT_SPLIT8', on)
        o = G.synthesis(ws, t('c'), t('v'), **kw)
        outs[on] = {k: o[k].clone() for k in ('image', 'image_raw', 'image_depth')}
    for k in outs[True]:
        assert torch.equal(outs[True][k], outs[False][k]), (k, _md(outs[True][k], outs[False][k]))
    # the small up-sampling layers' FIR writing split8 itself (layers.NCHW_FIR_SPLIT8; at batch 4: the 16 -> 32 layers) is not a pure layout
    # change — that kernel sums the separable filter's taps in another order than the float32 FIR kernel — so: equal to float32 rounding
    # of the layer outputs (a few 1e-6 relative on activations of magnitude ~10, 4e-5 measured on the image)
    monkeypatch.setattr(layers, 'NCHW_FIR_SPLIT8', False)
    o = G.synthesis(ws, t('c'), t('v'), **kw)
    for k in outs[True]:
        assert _md(outs[True][k], o[k]) <= 1e-4, (k, _md(outs[True][k], o[k]))


@pytest.mark.gpu
@pytest.mark.parametrize('fp32', [True, False])
def test_batch8_rows_equal_two_batch4_forwards(G, dev, fp32):
    """BASELINE.json configs[3]'s per-GPU share (8 seeds in one call): every row must be the frame the same seed gets in a batch-4
    call — other kernels are selected at batch 8 (split-K factors, pre-split eligibility, grid shapes), so the comparison is at a
    tolerance far below the parity tolerance, not bitwise.  (The ray marcher's depth clamp is a min / max over the WHOLE batch
    tensor, ray_marcher.py:54: with fixed ray_start / ray_end it differs between the two batchings only by the jitter extremes.)"""
    from next3d_amd import demo, layers
    layers.set_precision('bf16x3')
    R, Sc, Sf = 64, 48, 48
    G.rendering_kwargs['depth_resolution'], G.rendering_kwargs['depth_resolution_importance'] = Sc, Sf
    z, c, c_cond, v = demo.demo_batch(list(range(8)), device=dev)
    jitter, u = cases.rng_inputs(8, R, Sc, Sf)
    ws = G.mapping(z, c_cond, truncation_psi=0.7, truncation_cutoff=14)
    kw = dict(neural_rendering_resolution=R, noise_mode='const', force_fp32=fp32)
    full = G.synthesis(ws, c, v, depth_jitter=jitter, importance_u=u, **kw)
    assert tuple(full['image'].shape) == (8, 3, 512, 512) and torch.isfinite(full['image']).all()
    for h in range(2):
        s = slice(4 * h, 4 * h + 4)
        part = G.synthesis(ws[s], c[s], v[s], depth_jitter=jitter[s], importance_u=u[4 * h * R * R:(4 * h + 4) * R * R], **kw)
        # float16 blocks: a 1e-5 difference of the block input flips float16 roundings (an ulp at the feature maps' magnitude is
        # 2e-3..8e-3), so the image is compared at the float16 route's own tolerance (test_cpu_oracle.FP16_SR_TOL) plus its mean
        for k, tol in (('image', 2e-4 if fp32 else 1.2e-2), ('image_raw', 2e-4), ('image_depth', 2e-4)):
            e, m = _md(full[k][s], part[k]), float((full[k][s] - part[k]).abs().mean())
            print(f'batch 8 rows {4 * h}-{4 * h + 3} vs batch 4, fp32={fp32}, {k}: max abs diff {e:.3e}, mean {m:.3e}')
            assert e <= tol and m <= (2e-5 if (fp32 or k != 'image') else 1.2e-3), (k, h, e, m)


@pytest.mark.gpu
@pytest.mark.parametrize('opts', [dict(ray_start='auto', ray_end='auto', white_back=True, density_noise=0.25), dict(disparity_space_sampling=True)])
def test_rendering_kwargs_outside_the_ffhq_configuration(G, dev, opts):
    """generator.render reads rendering_kwargs the way the reference's renderer does (vr/renderer.py:95-153, vr/ray_marcher.py:56-57):
    'auto' ray bounds, white_back, density_noise, disparity-space sampling reach n3d_render_rays_ex — against the oracle (pinned
    to the reference for these options: tests/golden/render_opts.npz) on the generator's own decoder; other values still raise."""
    from oracle import renderer as oren
    N, R, Sc, Sf = 2, 16, 12, 12
    rk_old = dict(G.rendering_kwargs)
    try:
        G.rendering_kwargs.update(depth_resolution=Sc, depth_resolution_importance=Sf, **opts)
        g = torch.Generator().manual_seed(5)
        planes = torch.randn(N, 3, 32, 256, 256, generator=g) * 2
        jitter, u = cases.rng_inputs(N, R, Sc, Sf)
        nz = (torch.randn(N, R * R * Sc, 1, generator=g), torch.randn(N, R * R * Sf, 1, generator=g))
        d = np.load(os.path.join(GOLDEN, 'case_r64_s48.npz'))
        c = torch.from_numpy(d['c'])
        feat, depth = G.render(planes.permute(0, 1, 3, 4, 2).contiguous().to(dev), c.to(dev), R, jitter, u,
                               density_noise_draws=(nz[0].reshape(N, R * R, Sc), nz[1].reshape(N, R * R, Sf)))
        P = {k: v.detach().cpu() for k, v in G.state_dict().items() if k.startswith('decoder')}
        ro, rd = oren.ray_sampler(c[:, :16].reshape(N, 4, 4), c[:, 16:25].reshape(N, 3, 3), R)
        rgb_o, dep_o, _ = oren.importance_renderer(P, 'decoder', planes, ro, rd, dict(G.rendering_kwargs), jitter, u, noise=nz)
        e_rgb = (feat.cpu().reshape(N, 32, R * R).permute(0, 2, 1) - rgb_o).abs().amax(-1)
        e_dep = (depth.cpu().reshape(N, R * R) - dep_o[..., 0]).abs()
        print(opts, f'rgb max {float(e_rgb.max()):.2e} median {float(e_rgb.median()):.2e} depth max {float(e_dep.max()):.2e}')
        for e in (e_rgb, e_dep):
            assert float((e > 1e-3).float().mean()) <= 0.01 and float(e.median()) <= 1e-4
        G.rendering_kwargs['clamp_mode'] = 'mip'
        with pytest.raises(RuntimeError):
            G.render(planes.permute(0, 1, 3, 4, 2).contiguous().to(dev), c.to(dev), R, jitter, u)
    finally:
        G.rendering_kwargs.clear()
        G.rendering_kwargs.update(rk_old)


@pytest.fixture(scope='module')
def G16(dev):
    """The generator as legacy.load_network_pkl(force_fp16=True) rebuilds it (legacy.py:49-59): num_fp16_res = 4, conv_clamp = 256 in every
    backbone — the blocks of resolution >= 32 of all four StyleGAN2 / StyleUNet networks are float16 blocks."""
    from next3d_amd.generator import TriPlaneGenerator
    d = np.load(os.path.join(GOLDEN, 'demo_inputs.npz'))
    g = TriPlaneGenerator(512, 25, 512, 512, 3, (d['faces'], d['uvs'], d['uvfaces']), sr_num_fp16_res=4,
                          mapping_kwargs=dict(num_layers=2), rendering_kwargs=dict(RK),
                          sr_kwargs=dict(channel_base=32768, channel_max=512, fused_modconv_default='inference_only'),
                          uv_face_mask=mesh.synthetic_uv_face_mask(), channel_base=32768, channel_max=512,
                          fused_modconv_default='inference_only', num_fp16_res=4, conv_clamp=256)
    sd = spec.synthetic_state_dict(0)
    sd.update(mesh.mesh_buffers(d['faces'], d['uvs'], d['uvfaces']))
    g.load_state_dict(sd, strict=True)
    return g.eval().requires_grad_(False).to(dev)


@pytest.mark.gpu
@pytest.mark.parametrize('case', ['case_r32_s24', 'case_r64_s48'])
def test_fp16_backbones_match_reference_fp16_run(G16, dev, case):
    """SURVEY 8 (f4), backbone half: float16 blocks in the texture / static backbones and both StyleUNets (num_fp16_res = 4, conv_clamp
    = 256) on the f16 matrix-core kernels against the REFERENCE's own end-to-end float16 run (tests/golden/*_fp16bb.npz: its three
    SynthesisBlock classes executed on the CPU with the off-GPU float32 guard disabled, oracle/pin_against_reference.py
    --fp16-backbones).  Tolerances and their derivation: tests/test_cpu_oracle.py::FP16_BB_*.  force_fp32=True must give the float32
    route back (the float32 goldens, with conv_clamp = 256 never reached by these activations)."""
    from next3d_amd import layers
    from test_cpu_oracle import check_fp16_backbone_outputs
    layers.set_precision('bf16x3')
    g = np.load(os.path.join(GOLDEN, case + '_fp16bb.npz'))
    N, R, Sc, Sf = g['z'].shape[0], int(g['R']), int(g['Sc']), int(g['Sf'])
    G16.rendering_kwargs['depth_resolution'], G16.rendering_kwargs['depth_resolution_importance'] = Sc, Sf
    jitter, u = cases.rng_inputs(N, R, Sc, Sf)
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    ws = G16.mapping(t('z'), t('c_cond'), truncation_psi=float(g['psi']), truncation_cutoff=int(g['cutoff']))
    G16.keep_stages = True
    try:
        out = G16.synthesis(ws, t('c'), t('v'), neural_rendering_resolution=R, noise_mode='const', depth_jitter=jitter, importance_u=u)
        st = G16._debug
        stages = {'textures': st['textures'], 'static_plane': st['static'], 'mouths_plane': st['mouths'], 'rendering_stitch': st['stitch']}
        check_fp16_backbone_outputs(g, stages, out, 'HIP')
        out32 = G16.synthesis(ws, t('c'), t('v'), neural_rendering_resolution=R, noise_mode='const', depth_jitter=jitter, importance_u=u, force_fp32=True)
    finally:
        G16.keep_stages = False
    d32 = np.load(os.path.join(GOLDEN, case + '.npz'))
    assert _md(out32['image'][..., ::4, ::4], d32['image_sub4']) <= 1e-3 and _md(out32['image_raw'], d32['image_raw']) <= 1e-3
    assert _md(out['image'], out32['image']) > 1e-4                       # the float16 blocks do run


@pytest.mark.gpu
def test_default_call_of_a_force_fp16_generator_keeps_the_float16_blocks(G16, dev):
    """`G.synthesis(ws, c, v)` with every default — noise_mode 'random' (tat/networks_stylegan2.py:311) — on the force_fp16 generator: the float16
    blocks stay on the f16 kernels (their noisy layers run sample by sample: one noise image per launch), no warning, no float32 fallback; the draws come
    from the device generator, so the same seed gives the same image; the image differs from the noise-free one by the noise and from the float32
    route's by more than the noise-free float16 / float32 distance would explain nothing — it is checked to be finite and in range."""
    import warnings
    from next3d_amd import demo
    z, c, c_cond, v = demo.demo_batch([0, 1], device=dev)
    ws = G16.mapping(z, c_cond, truncation_psi=0.7, truncation_cutoff=14)
    jitter, u = cases.rng_inputs(2, 32, 24, 24)
    G16.rendering_kwargs['depth_resolution'], G16.rendering_kwargs['depth_resolution_importance'] = 24, 24
    kw = dict(neural_rendering_resolution=32, depth_jitter=jitter, importance_u=u)
    G16.__dict__.pop('_warned32', None)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        torch.manual_seed(7)
        a = G16.synthesis(ws, c, v, **kw)['image']
        torch.manual_seed(7)
        b = G16.synthesis(ws, c, v, **kw)['image']
    assert not [x for x in w if 'float16' in str(x.message)], [str(x.message) for x in w]
    assert torch.equal(a, b) and bool(torch.isfinite(a).all())
    quiet = G16.synthesis(ws, c, v, noise_mode='none', **kw)['image']
    const = G16.synthesis(ws, c, v, noise_mode='const', **kw)['image']
    d_rand, d_const = _md(a, quiet), _md(const, quiet)
    print(f'force_fp16 generator: |random - none| {d_rand:.3e}, |const - none| {d_const:.3e}')
    assert d_rand > 1e-3 and d_rand < 50 * max(d_const, 1e-3)               # noise of the same strengths, another draw


@pytest.mark.gpu
def test_fp16_blocks_teacher_forced(G16, dev, monkeypatch):
    """every float16 block of the four backbones ONE BLOCK DEEP against the reference's own block output on a stated input
    (tests/golden/fp16_blocks.npz, tests/_fp16_blocks.py).  With the reference's off-GPU bias_act rounding (layers.F16_REF_CPU_ROUNDING: the fixture was
    produced off-GPU) the f16 matrix-core kernels reproduce the block almost bit for bit — what is left is the accumulation order of the half
    convolutions; the SAME block run on this library's float32 route fails the bound several times over, i.e. this test (unlike the end-to-end
    FP16_BB tolerances) distinguishes the float16 route from the float32 one."""
    import _fp16_blocks as fb
    from next3d_amd import _lib, layers, networks
    layers.set_precision('bf16x3')
    S = G16._prep()
    nets = {'texture': S.texture, 'static': S.static, 'mouth': S.mouth, 'blend': S.blend}
    worst, bad = [1.0, 0.0, 0.0], []
    for b in fb.blocks():
        net = nets[b['net']]
        blk = net.blocks[b['res']]
        bank = net.bank.compute(b['ws'].to(dev))
        img_in = None if b['img'] is None else b['img'].to(dev)
        monkeypatch.setattr(layers, 'F16_REF_CPU_ROUNDING', True)
        w16 = networks._f16_weights({b['res']: blk}, bank, 1)[b['res']]
        xh, img = networks._f16_block(blk, _lib.H8.from_nchw(b['x'].float().to(dev)), img_in, net.fir, 'const', w16)
        same, mean_ulp, ie = fb.compare(xh.to_float().half(), img, b)
        # ... the float32 route of THIS library on the same input (x is float16-representable, so both routes start from identical values)
        monkeypatch.setattr(layers, 'F16_REF_CPU_ROUNDING', False)
        x32, img32, _ = blk(b['x'].float().to(dev), img_in, bank, 1, net.fir, 'const')
        same32, mean32, _ = fb.compare(x32.half(), img32, b)
        print(f"{b['net']} b{b['res']}: f16 kernels {100 * same:.2f} % bit-equal, mean {mean_ulp:.3f} ulp, img {ie:.2f} ulp | float32 route {100 * same32:.1f} %, {mean32:.2f} ulp "
              f"(the reference's own float32 route: {100 * b['fp32_route'][0]:.1f} %, {b['fp32_route'][1]:.2f} ulp)")
        bad += [(b['net'], b['res'], 'f16', same, mean_ulp, ie)] if not (same >= fb.MIN_EQUAL_HIP and mean_ulp <= fb.MAX_MEAN_ULP_HIP and ie <= fb.IMG_TOL_ULP) else []
        bad += [(b['net'], b['res'], 'f32 passes', same32, mean32)] if not (same32 < 0.5 * fb.MIN_EQUAL_HIP and mean32 > 3 * fb.MAX_MEAN_ULP_HIP) else []    # the float32 route FAILS the bound
        worst = [min(worst[0], same), max(worst[1], mean_ulp), max(worst[2], ie)]
    print('worst block: %.2f %% bit-equal, mean %.3f ulp, img %.2f ulp' % (100 * worst[0], worst[1], worst[2]))
    assert not bad, bad


@pytest.fixture(scope='module')
def GW(dev):
    """The generator on the SECOND weight draw: spec.synthetic_state_dict(1, profile='wide') — heavy-tailed weights, styles x 3, noise / bias 0.3."""
    from next3d_amd.generator import TriPlaneGenerator
    d = np.load(os.path.join(GOLDEN, 'demo_inputs.npz'))
    g = TriPlaneGenerator(512, 25, 512, 512, 3, (d['faces'], d['uvs'], d['uvfaces']), sr_num_fp16_res=4,
                          mapping_kwargs=dict(num_layers=2), rendering_kwargs=dict(RK),
                          sr_kwargs=dict(channel_base=32768, channel_max=512, fused_modconv_default='inference_only'),
                          uv_face_mask=mesh.synthetic_uv_face_mask(), channel_base=32768, channel_max=512,
                          fused_modconv_default='inference_only', num_fp16_res=0, conv_clamp=None)
    sd = spec.synthetic_state_dict(1, profile='wide')
    sd.update(mesh.mesh_buffers(d['faces'], d['uvs'], d['uvfaces']))
    g.load_state_dict(sd, strict=True)
    return g.eval().requires_grad_(False).to(dev)


@pytest.mark.gpu
@pytest.mark.parametrize('precision', ['fp32', 'bf16x3'])
def test_second_weight_draw_matches_reference_golden(GW, dev, precision):
    """the benched configuration (batch 4, 512² / 64² / 48 + 48) on a second, trained-like weight draw
    (tests/golden/case_r64_s48_b4_w1.npz: the reference's own run on spec.synthetic_state_dict(1, profile='wide') — activations of 10² in the
    blending network, image values up to ~5).  The split-bf16 arithmetic's error on this draw is printed next to the unit draw's 8.8e-5;
    tolerance 1e-3 max-abs on RGB as everywhere (north_star), stages relative to their largest value."""
    from next3d_amd import layers
    d = np.load(os.path.join(GOLDEN, 'case_r64_s48_b4_w1.npz'))
    assert str(d['weights_profile']) == 'wide' and int(d['weights_seed']) == 1
    N, R, Sc, Sf = d['z'].shape[0], int(d['R']), int(d['Sc']), int(d['Sf'])
    GW.rendering_kwargs['depth_resolution'], GW.rendering_kwargs['depth_resolution_importance'] = Sc, Sf
    jitter, u = cases.rng_inputs(N, R, Sc, Sf)
    layers.set_precision(precision)
    GW.keep_stages = True
    try:
        ws = GW.mapping(torch.from_numpy(d['z']).to(dev), torch.from_numpy(d['c_cond']).to(dev), truncation_psi=float(d['psi']), truncation_cutoff=int(d['cutoff']))
        out = GW.synthesis(ws, torch.from_numpy(d['c']).to(dev), torch.from_numpy(d['v']).to(dev), neural_rendering_resolution=R, noise_mode='const',
                           depth_jitter=jitter, importance_u=u, force_fp32=True)
        st = GW._debug
    finally:
        layers.set_precision('bf16x3')
        GW.keep_stages = False
    amax = dict(zip(('textures', 'mouths_plane', 'rendering_stitch', 'static_plane', 'image'), d['stage_absmax']))
    rep = {'ws': _md(ws, d['ws']), 'textures': _md(st['textures'][..., ::8, ::8], d['textures_sub8']), 'static_plane': _md(st['static'][..., ::8, ::8], d['static_plane_sub8']),
           'image_raw': _md(out['image_raw'], d['image_raw']), 'image_depth': _md(out['image_depth'], d['image_depth']), 'image': _md(out['image'][..., ::4, ::4], d['image_sub4'])}
    print(f'wide draw, {precision}: ' + ' '.join(f'{k}={v:.3e}' for k, v in rep.items()) + ' | reference |max|: ' + ' '.join(f'{k}={v:.1f}' for k, v in amax.items()))
    assert int(((st['alpha'].cpu() * 255).round() != torch.from_numpy(d['alpha'].astype(np.float32))).sum()) == 0 and _md(st['bbox'], d['mouth_mask']) == 0
    assert rep['ws'] <= 1e-4
    assert rep['textures'] <= 1e-4 * max(1.0, amax['textures']) * 10 and rep['static_plane'] <= 1e-4 * max(1.0, amax['static_plane']) * 10
    assert rep['image_raw'] <= 1e-3 and rep['image'] <= 1e-3 and rep['image_depth'] <= 1e-3, rep


@pytest.mark.gpu
@pytest.mark.parametrize('depth', [8, 1])
def test_mapping_depths_match_reference_golden(dev, depth):
    """mapping_kwargs.num_layers = 8 (the MappingNetwork's default, what a constructor call WITHOUT the key builds in the reference) and 1 against the reference's
    own ws (tests/golden/mapping_depth.npz); the forward runs on such a model."""
    from next3d_amd import demo
    g = np.load(os.path.join(GOLDEN, 'mapping_depth.npz'))
    G, _ = demo.build_generator(dev, mapping_layers=depth)
    assert G.backbone.mapping.num_layers == depth
    ws = G.mapping(torch.from_numpy(g['z']).to(dev), torch.from_numpy(g['c_cond']).to(dev), truncation_psi=float(g['psi']), truncation_cutoff=int(g['cutoff']))
    err = _md(ws, g[f'ws_depth{depth}'])
    print(f'mapping depth {depth}: ws max-abs vs reference {err:.3e}')
    assert err <= 1e-4
    z, c, c_cond, v = demo.demo_batch([0], device=dev)
    out = G.synthesis(ws[:1], c, v, neural_rendering_resolution=32, noise_mode='const')
    assert tuple(out['image'].shape) == (1, 3, 512, 512) and bool(torch.isfinite(out['image']).all())


@pytest.mark.gpu
@pytest.mark.parametrize('suffix', ['cb16384', 'cb16384_cm256'])
def test_other_channel_widths_match_reference_golden(dev, suffix):
    """Backbone widths other than the ffhq-512 pickle's (`--cbase 16384`, `--cmax 256` of train_next3d.py:199-200: 64-channel blocks at 256 x 256, 256 at the
    low resolutions) against the REFERENCE's own run (tests/golden/case_r32_s24_cb*.npz, oracle/pin_against_reference.py --channel-widths: built by the
    reference's constructors), both arithmetic settings at tolerance 1e-3, and the default route (float16 super-resolution blocks)."""
    from next3d_amd import demo, layers
    d = np.load(os.path.join(GOLDEN, f'case_r32_s24_{suffix}.npz'))
    cb, cm = int(d['channel_base']), int(d['channel_max'])
    rk = dict(RK, depth_resolution=int(d['Sc']), depth_resolution_importance=int(d['Sf']))
    g, _ = demo.build_generator(dev, rendering_kwargs=rk, channel_base=cb, channel_max=cm)
    R, step = int(d['R']), int(d['image_step'])
    jitter, u = cases.rng_inputs(1, R, int(d['Sc']), int(d['Sf']))
    t = lambda k: torch.from_numpy(d[k]).to(dev)
    try:
        for precision in ('fp32', 'bf16x3'):
            layers.set_precision(precision)
            ws = g.mapping(t('z'), t('c_cond'), truncation_psi=float(d['psi']), truncation_cutoff=int(d['cutoff']))
            assert _md(ws, d['ws']) <= 1e-4
            out = g.synthesis(ws, t('c'), t('v'), neural_rendering_resolution=R, noise_mode='const', depth_jitter=jitter, importance_u=u, force_fp32=True)
            rep = {'image': _md(out['image'][..., ::step, ::step], d['image_sub']), 'image_raw': _md(out['image_raw'], d['image_raw']), 'image_depth': _md(out['image_depth'], d['image_depth'])}
            print(suffix, precision, ' '.join(f'{k}={v:.3e}' for k, v in rep.items()))
            assert all(v <= 1e-3 for v in rep.values()), rep
    finally:
        layers.set_precision('bf16x3')
    out16 = g.synthesis(ws, t('c'), t('v'), neural_rendering_resolution=R, noise_mode='const', depth_jitter=jitter, importance_u=u)
    e16 = _md(out16['image'], out['image'])
    print(suffix, f'default route vs float32 route: {e16:.3e}')
    assert 1e-5 < e16 <= 2e-2
    # batch 4 (the benchmark's batch: other kernels are eligible than at batch 1) equals four batch-1 calls
    z4, c4, cc4, v4 = demo.demo_batch([0, 1, 2, 3], device=dev)
    ws4 = g.mapping(z4, cc4, truncation_psi=0.7, truncation_cutoff=14)
    j4, u4 = cases.rng_inputs(4, R, int(d['Sc']), int(d['Sf']))
    j4, u4 = j4.to(dev), u4.to(dev)
    o4 = g.synthesis(ws4, c4, v4, neural_rendering_resolution=R, noise_mode='const', depth_jitter=j4, importance_u=u4, force_fp32=True)['image']
    for i in range(4):
        oi = g.synthesis(ws4[i:i + 1], c4[i:i + 1], v4[i:i + 1], neural_rendering_resolution=R, noise_mode='const', depth_jitter=j4[i:i + 1],
                         importance_u=u4[i * R * R:(i + 1) * R * R], force_fp32=True)['image']
        assert _md(o4[i:i + 1], oi) <= 2e-4, i


@pytest.mark.gpu
@pytest.mark.parametrize('cls,res,fixture', [('SuperresolutionHybrid8X', 512, 'sr8X'), ('SuperresolutionHybrid4X', 256, 'sr4X'), ('SuperresolutionHybrid2X', 128, 'sr2X'),
                                             ('SuperresolutionHybrid2X', 128, 'sr2X_r64'), ('SuperresolutionHybrid4X', 256, 'sr4X_r128')])
def test_other_superresolution_modules_match_reference_golden(dev, cls, res, fixture):
    """VERDICT r4 missing #4: the reference's other super-resolution modules (tat/superresolution.py:29-124: 8X = other channel counts at 512 x 512, 4X = a
    SynthesisBlockNoUp first, 256 x 256, resizes only a smaller render, 2X = 128 x 128) against the REFERENCE's own run (tests/golden/case_r32_s24_sr*.npz,
    oracle/pin_against_reference.py --sr-modules: built by the reference's constructors, state-dict names diffed against spec.build_spec), both
    arithmetic settings, tolerance 1e-3; and the default (no force_fp32) call: float16 kernels for 8X, the float32 fallback (one warning) for 4X / 2X."""
    import warnings
    from next3d_amd import layers
    from next3d_amd.generator import TriPlaneGenerator
    d0 = np.load(os.path.join(GOLDEN, 'demo_inputs.npz'))
    # (sr2X_r64 / sr4X_r128, round 6: a render AT the module's input resolution — no resize, so the reference's in-place `img.add_(y)` of SynthesisBlockNoUp lands in
    # the returned 'image_raw': reproduced)
    d = np.load(os.path.join(GOLDEN, f'case_r32_s24_{fixture}.npz'))
    assert str(d['sr_class']) == cls
    rk = dict(RK, depth_resolution=int(d['Sc']), depth_resolution_importance=int(d['Sf']), superresolution_module='training_avatar_texture.superresolution.' + cls)
    with pytest.raises(RuntimeError):
        TriPlaneGenerator(512, 25, 512, res * 2, 3, (d0['faces'], d0['uvs'], d0['uvfaces']), sr_num_fp16_res=4, rendering_kwargs=dict(rk), uv_face_mask=mesh.synthetic_uv_face_mask())
    g = TriPlaneGenerator(512, 25, 512, res, 3, (d0['faces'], d0['uvs'], d0['uvfaces']), sr_num_fp16_res=4, mapping_kwargs=dict(num_layers=2), rendering_kwargs=dict(rk),
                          sr_kwargs=dict(channel_base=32768, channel_max=512, fused_modconv_default='inference_only'), uv_face_mask=mesh.synthetic_uv_face_mask(),
                          channel_base=32768, channel_max=512, fused_modconv_default='inference_only', num_fp16_res=0, conv_clamp=None)
    sd = spec.synthetic_state_dict(0, sr=cls)
    sd.update(mesh.mesh_buffers(d0['faces'], d0['uvs'], d0['uvfaces']))
    assert sorted(k for k in g.state_dict() if k.startswith('superresolution')) == list(d['state_dict_names'])
    g.load_state_dict(sd, strict=True)
    g = g.eval().requires_grad_(False).to(dev)
    R, step = int(d['R']), int(d['image_step'])
    jitter, u = cases.rng_inputs(1, R, int(d['Sc']), int(d['Sf']))
    t = lambda k: torch.from_numpy(d[k]).to(dev)
    try:
        for precision in ('fp32', 'bf16x3'):
            layers.set_precision(precision)
            ws = g.mapping(t('z'), t('c_cond'), truncation_psi=float(d['psi']), truncation_cutoff=int(d['cutoff']))
            out = g.synthesis(ws, t('c'), t('v'), neural_rendering_resolution=R, noise_mode='const', depth_jitter=jitter, importance_u=u, force_fp32=True)
            assert tuple(out['image'].shape) == (1, 3, res, res)
            rep = {'image': _md(out['image'][..., ::step, ::step], d['image_sub']), 'image_raw': _md(out['image_raw'], d['image_raw']), 'image_depth': _md(out['image_depth'], d['image_depth'])}
            print(cls, precision, ' '.join(f'{k}={v:.3e}' for k, v in rep.items()))
            assert all(v <= 1e-3 for v in rep.values()), rep
    finally:
        layers.set_precision('bf16x3')
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        out16 = g.synthesis(ws, t('c'), t('v'), neural_rendering_resolution=R, noise_mode='const', depth_jitter=jitter, importance_u=u)
    e16 = _md(out16['image'], out['image'])
    print(cls, f'default route vs float32 route: {e16:.3e}; warnings: {len(w)}')
    if cls.endswith('8X'):
        assert 1e-5 < e16 <= 2e-2 and not w                              # float16 blocks on the f16 kernels
    else:
        assert e16 == 0.0 and len(w) == 1                                # SynthesisBlockNoUp has no float16 form here: float32 arithmetic, one warning


@pytest.mark.gpu
def test_orbit_frames_match_reference_golden(G, dev):
    """VERDICT r5 4c — the closest thing to "gen_videos_next3d.py on hardware" this environment allows: the script's render loop (:96-158) with the argument
    values the script itself computes — one seed = one keyframe, w_frames = 120, --trunc 0.7, --sample_mult 2 (96 + 96 samples), the float64 scipy cubic
    interpolation of the tiled keyframe latents, `G.synthesis(ws=w.unsqueeze(0), c=c[0:1], v=verts[0:1], noise_mode='const')` — for frames 0 / 60 / 119 of the
    120-frame orbit, against the REAL reference's outputs for the same frames (tests/golden/orbit_frames.npz, oracle/pin_against_reference.py --orbit; the
    renderer's random draws injected on both sides).  Run the way the script runs it (plane caching off, `cache_backbone` pattern on: the orbit never changes
    the latents or the mesh, so the second form must return the same frames), on both precisions of the float32 route and on the scripts' default route."""
    import scipy.interpolate
    from next3d_amd import layers
    d = np.load(os.path.join(GOLDEN, 'orbit_frames.npz'))
    R, Sc, Sf, w_frames, wraps = int(d['R']), int(d['Sc']), int(d['Sf']), int(d['w_frames']), int(d['wraps'])
    assert (R, Sc, Sf) == (G.neural_rendering_resolution, 96, 96)
    G.rendering_kwargs['depth_resolution'], G.rendering_kwargs['depth_resolution_importance'] = Sc, Sf       # gen_videos_next3d.py:288-289
    jitter, u = cases.rng_inputs(1, R, Sc, Sf)
    device = dev
    zs = torch.from_numpy(np.stack([np.random.RandomState(int(d['seed'])).randn(G.z_dim)])).to(device)         # :95 (float64, as the script builds it)
    c = torch.from_numpy(d['c_cond']).to(device)
    verts = torch.from_numpy(d['v']).to(device)
    worst = {}
    try:
        for precision in ('fp32', 'bf16x3'):
            layers.set_precision(precision)
            ws = G.mapping(z=zs, c=c, truncation_psi=float(d['psi']), truncation_cutoff=int(d['cutoff']))      # :104
            assert _md(ws, d['ws']) <= 1e-4
            _ = G.synthesis(ws[:1], c[:1], verts[:1])                                                          # :105 warm up (its own random draws)
            ws_g = ws.reshape(1, 1, 1, *ws.shape[1:])
            x = np.arange(-1 * wraps, 1 * (wraps + 1))                                                         # :113-116, num_keyframes = 1
            y = np.tile(ws_g[0][0].cpu().numpy(), [wraps * 2 + 1, 1, 1])
            interp = scipy.interpolate.interp1d(x, y, kind='cubic', axis=0)
            for route, kw in (('float32 route', dict(force_fp32=True)), ('default route', {})):
                for cached in (False, True):
                    for k, frame_idx in enumerate(int(f) for f in d['frames']):
                        cf = torch.from_numpy(d[f'c_{frame_idx}']).to(device)                                  # the script's own LookAtPoseSampler pose of this frame (:131-137)
                        w = torch.from_numpy(interp(frame_idx / w_frames)).to(device)                          # :140-141: float64
                        assert w.dtype == torch.float64
                        ckw = dict(cache_backbone=(k == 0), use_cached_backbone=(k > 0)) if cached else {}
                        o = G.synthesis(ws=w.unsqueeze(0), c=cf[0:1], v=verts[0:1], noise_mode='const', depth_jitter=jitter, importance_u=u, **kw, **ckw)      # :153
                        e = (_md(o['image'][..., ::2, ::2], d[f'image_{frame_idx}_sub2']), _md(o['image_raw'], d[f'image_raw_{frame_idx}']),
                             _md(o['image_depth'], d[f'image_depth_{frame_idx}']), _md(o['image'].mean(dim=(2, 3)), d[f'image_mean_{frame_idx}']))
                        key = (precision, route)
                        worst[key] = tuple(max(a, b) for a, b in zip(worst.get(key, (0, 0, 0, 0)), e))
                        if route == 'float32 route':
                            assert e[0] <= 1e-3 and e[1] <= 1e-3 and e[2] <= 1e-3, (precision, route, cached, frame_idx, e)           # north_star
                        else:      # float16 super-resolution blocks against the reference's float32 run of the same frames: the documented bound of that route
                            assert e[0] <= 1.2e-2 and e[3] <= 1.2e-3 and e[1] <= 1e-3 and e[2] <= 1e-3, (precision, route, cached, frame_idx, e)
    finally:
        layers.set_precision('bf16x3')
        G.rendering_kwargs['depth_resolution'], G.rendering_kwargs['depth_resolution_importance'] = 48, 48
    for key, e in worst.items():
        print(f'orbit {key}: image {e[0]:.2e} image_raw {e[1]:.2e} image_depth {e[2]:.2e} image mean {e[3]:.2e}')


@pytest.mark.gpu
def test_inference_mode_model_captures_a_graph(dev):
    """a generator built AND called under torch.inference_mode() holds inference tensors (no version counter).  The separable-filter factor must still be
    decided once (upfirdn2d.fir_factor caches such a filter by object + storage address) — a per-call device-to-host read would serialise the stream lanes and is
    illegal inside a HIP-graph capture: synthesis_graph has to capture and replay, bit-identically to the eager call."""
    from next3d_amd import demo
    from next3d_amd.torch_utils.ops import upfirdn2d as uf
    with torch.inference_mode():
        g, _ = demo.build_generator(dev)
        assert next(g.parameters()).is_inference()
        z, c, c_cond, v = demo.demo_batch([0], device=dev)
        ws = g.mapping(z, c_cond, truncation_psi=0.7, truncation_cutoff=14)
        jit, u = cases.rng_inputs(1, 64, 48, 48)
        kw = dict(neural_rendering_resolution=64, noise_mode='const', depth_jitter=jit.to(dev), importance_u=u.to(dev))
        eager = g.synthesis(ws, c, v, **kw)['image'].clone()
        n_cached = len(uf._FIR1D)
        assert n_cached >= 1
        out = g.synthesis_graph(ws, c, v, **kw)['image']
        torch.cuda.synchronize()
        assert torch.equal(out, eager)
        out2 = g.synthesis_graph(ws, c, v, **kw)['image']
        torch.cuda.synchronize()
        assert torch.equal(out2, eager) and len(uf._FIR1D) == n_cached      # nothing re-decided per call
