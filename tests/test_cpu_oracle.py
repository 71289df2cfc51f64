"""CPU (-m "not gpu") tests: the oracle against the golden fixtures produced by the REAL reference
(oracle/pin_against_reference.py) and against closed-form cases."""
import os

import numpy as np
import pytest
import torch

from next3d_amd import mesh, spec
from oracle import cases, generator as ogen, ops as O, raster, renderer

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
RK = dict(c_gen_conditioning_zero=True, c_scale=1.0, depth_resolution=24, depth_resolution_importance=24, ray_start=2.25,
          ray_end=3.3, box_warp=1)


@pytest.fixture(scope='module')
def state():
    d = np.load(os.path.join(GOLDEN, 'demo_inputs.npz'))
    sd = spec.synthetic_state_dict(0)
    sd.update(mesh.mesh_buffers(d['faces'], d['uvs'], d['uvfaces']))
    return sd


@pytest.mark.parametrize('case', ['case_r32_s24', 'case_r64_s96'])
def test_oracle_reproduces_reference_golden(state, case):
    """configs[0] of BASELINE.json (1 seed, R=32, 24+24 samples, fp32 CPU) and the 96+96-sample case with two perturbed meshes
    (mouth boxes of different sizes): oracle == reference outputs, bit for bit."""
    g = np.load(os.path.join(GOLDEN, case + '.npz'))
    R, Sc, Sf = int(g['R']), int(g['Sc']), int(g['Sf'])
    jitter, u = cases.rng_inputs(g['z'].shape[0], R, Sc, Sf)
    rk = dict(RK, depth_resolution=Sc, depth_resolution_importance=Sf)
    ws = ogen.mapping(state, torch.from_numpy(g['z']), torch.from_numpy(g['c_cond']), rk, truncation_psi=float(g['psi']),
                      truncation_cutoff=int(g['cutoff']))
    assert np.abs(ws.numpy() - g['ws']).max() == 0
    out, st = ogen.synthesis(state, ws, torch.from_numpy(g['c']), torch.from_numpy(g['v']), mesh.synthetic_uv_face_mask(), rk,
                             jitter, u, neural_rendering_resolution=R, return_stages=True)
    assert np.abs(out['image_raw'].numpy() - g['image_raw']).max() <= 1e-6
    assert np.abs(out['image_depth'].numpy() - g['image_depth']).max() <= 1e-6
    assert np.abs(out['image'][..., ::4, ::4].numpy() - g['image_sub4']).max() <= 1e-6
    assert np.abs(out['image'].mean(dim=(2, 3)).numpy() - g['image_mean']).max() <= 1e-6
    assert np.array_equal((st['alpha'].numpy() * 255).round().astype(np.uint8), g['alpha'])
    # point queries (G.sample_mixed, triplane_next3d.py:278) on the same planes
    smp = ogen.run_model(state, st['blended_planes'], torch.from_numpy(g['sample_coords']), rk)
    assert np.abs(smp['rgb'].numpy() - g['sample_rgb']).max() <= 1e-6
    assert np.abs(smp['sigma'].numpy() - g['sample_sigma']).max() <= 1e-6
    assert np.array_equal(st['mouth_mask'].numpy(), g['mouth_mask'])
    assert np.abs(st['textures'][..., ::8, ::8].numpy() - g['textures_sub8']).max() <= 1e-6
    assert np.abs(st['blended_planes'][..., ::8, ::8].numpy() - g['blended_planes_sub8']).max() <= 1e-6


def test_operator_oracle_reproduces_reference_ref_ops():
    """oracle/ops.py against the outputs of the reference's own _bias_act_ref / _upfirdn2d_ref / _filtered_lrelu_ref
    (tests/golden/ref_ops.npz, oracle/pin_ops_against_reference.py): all 9 activations with gain / clamp / alpha and the bias
    along dim 0 / 1 / 3, 2-D / separable / asymmetric filters, per-axis up x down, negative padding, flips — bit for bit."""
    import _ref_ops
    mod = {'bias_act': O.bias_act, 'upfirdn2d': O.upfirdn2d, 'filtered_lrelu': O.filtered_lrelu}
    cases_ = _ref_ops.load()
    assert len(cases_) >= 49 and {c[1] for c in cases_} == {'bias_act', 'upfirdn2d', 'filtered_lrelu'}
    for i, op, kw, t, y_ref in cases_:
        y = _ref_ops.run(mod, op, kw, t, O.setup_filter)
        assert y.shape == y_ref.shape and torch.equal(y, y_ref), (i, op, kw)


def test_upfirdn2d_against_direct_definition():
    """upfirdn2d == explicit zero-insert / pad / correlate-with-flipped-filter / decimate, on ragged shapes."""
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1, 2, 5, 7, generator=g)
    f = torch.randn(3, 4, generator=g)
    up, down, pad = 2, 3, (2, 1, 0, 3)
    y = O.upfirdn2d(x, f, up=up, down=down, padding=list(pad))
    z = torch.zeros(1, 2, 5 * up, 7 * up)
    z[:, :, ::up, ::up] = x
    z = torch.nn.functional.pad(z, [pad[0], pad[1], pad[2], pad[3]])
    H, W = z.shape[2] - 3 + 1, z.shape[3] - 4 + 1
    ref = torch.zeros(1, 2, H, W)
    ff = f.flip([0, 1])
    for i in range(H):
        for j in range(W):
            ref[:, :, i, j] = (z[:, :, i:i + 3, j:j + 4] * ff).sum(dim=(2, 3))
    assert torch.allclose(y, ref[:, :, ::down, ::down], atol=1e-5)


def test_rasterizer_and_floodfill_closed_form():
    verts = torch.tensor([[[-0.5, -0.5, 1.0], [0.5, -0.5, 1.0], [0.0, 0.5, 2.0],        # front-facing
                           [-0.9, -0.9, 0.5], [-0.9, -0.8, 0.5], [-0.8, -0.9, 0.5]]])    # back-facing (culled)
    faces = torch.tensor([[0, 1, 2], [3, 4, 5]])
    area = lambda a, b, c: (a[0] - b[0]) * (c[1] - b[1]) - (a[1] - b[1]) * (c[0] - b[0])
    v = verts[0]
    front = [i for i, f in enumerate(faces.tolist()) if area(v[f[0]], v[f[1]], v[f[2]]) > 0]
    p2f, zbuf, bary = raster.rasterize_meshes(verts, faces, image_size=64)
    hit = p2f[0, :, :, 0] >= 0
    assert set(p2f[0, :, :, 0][hit].unique().tolist()) == set(front) and len(front) == 1
    big = 0 in front                                    # exactly one of the two windings survives back-face culling
    assert abs(int(hit.sum()) - (512 if big else 5)) < (80 if big else 6)      # NDC area 0.5 -> 0.5 * 64*64/4 = 512 px
    b = bary[0, :, :, 0][hit]
    assert torch.allclose(b.sum(-1), torch.ones_like(b[:, 0]), atol=1e-5) and bool((b > 0).all())
    zhit = zbuf[0, :, :, 0][hit]
    assert float(zhit.min()) >= 0.5 - 1e-6 and float(zhit.max()) <= 2.0 + 1e-6
    assert bool((p2f[0, :, :, 0][~hit] == -1).all())
    if big:   # +Y is UP in PyTorch3D NDC: the apex (y=+0.5) lands in the top rows (fewer covered pixels there)
        rows = hit.any(dim=1).nonzero().flatten()
        assert hit[rows.min()].sum() < hit[rows.max()].sum()
    # z_invalid (PyTorch3D CheckPointOutsideBoundingBox: zlims.x < kEpsilon): a face with ONE vertex at / behind the camera plane covers nothing
    # although its zmax is positive; the same face moved in front of the plane is drawn
    vz, fv = verts.clone(), int(faces[front[0]][0])                        # one vertex of the face that IS drawn
    vz[0, fv, 2] = -0.25
    assert int((raster.rasterize_meshes(vz, faces, image_size=64)[0] >= 0).sum()) == 0
    vz[0, fv, 2] = 2e-8
    assert int((raster.rasterize_meshes(vz, faces, image_size=64)[0] >= 0).sum()) == int(hit.sum())
    img = np.zeros((8, 8), np.float32)
    img[2:6, 2:6] = 255.0
    img[3:5, 3:5] = 0.0                                                     # enclosed hole must NOT be filled
    raster.floodfill_fixed_range(img, 255.0, 0.0, 254.0)
    assert img[0, 0] == 255.0 and img[3, 3] == 0.0 and (img[2, 2:6] == 255.0).all()
    a = torch.zeros(1, 1, 8, 8)
    a[0, 0, 2:6, 2:6] = 1.0
    a[0, 0, 3:5, 3:5] = 0.0
    filled = raster.fill_mouth(a)
    assert float(filled[0, 0, 3, 3]) == 1.0 and float(filled[0, 0, 0, 0]) == 0.0


def test_renderer_properties():
    """Size-independent properties: weights form a sub-convex combination; importance samples stay inside the coarse range;
    constant planes give a constant colour."""
    g = torch.Generator().manual_seed(3)
    N, R, Sc, Sf = 1, 8, 12, 12
    P = {'decoder.net.0.weight': torch.randn(64, 32, generator=g), 'decoder.net.0.bias': torch.zeros(64),
         'decoder.net.2.weight': torch.randn(33, 64, generator=g), 'decoder.net.2.bias': torch.zeros(33)}
    planes = torch.randn(1, 1, 32, 1, 1, generator=g).expand(N, 3, 32, 16, 16).contiguous()
    c2w = torch.eye(4)[None].clone()
    c2w[0, 2, 3] = -2.7
    K = torch.tensor([[[4.26, 0, 0.5], [0, 4.26, 0.5], [0, 0, 1]]])
    ray_o, ray_d = renderer.ray_sampler(c2w, K, R)
    assert torch.allclose(ray_d.norm(dim=-1), torch.ones(N, R * R), atol=1e-6)
    jitter, u = cases.rng_inputs(N, R, Sc, Sf)
    opts = dict(depth_resolution=Sc, depth_resolution_importance=Sf, ray_start=2.25, ray_end=3.3, box_warp=1)
    rgb, depth, wsum = renderer.importance_renderer(P, 'decoder', planes, ray_o, ray_d, opts, jitter, u)
    assert float(wsum.max()) <= 1.0 + 1e-5 and float(wsum.min()) >= 0.0
    assert float(depth.min()) >= 2.25 and float(depth.max()) <= 3.3 + (3.3 - 2.25) / (Sc - 1)
    inside = (rgb + 1) / 2 / wsum.clamp_min(1e-6)          # constant colour field -> composite = colour * sum(w)
    assert float((inside - inside.mean(dim=1, keepdim=True)).abs().max()) < 5e-2


RENDER_OPT_CASES = {'white_back': dict(ray_start=2.25, ray_end=3.3, white_back=True),
                    'disparity': dict(ray_start=2.25, ray_end=3.3, disparity_space_sampling=True),
                    'auto': dict(ray_start='auto', ray_end='auto'), 'auto_wide_fov': dict(ray_start='auto', ray_end='auto'),
                    'density_noise': dict(ray_start=2.25, ray_end=3.3, density_noise=0.5),
                    'all': dict(ray_start='auto', ray_end='auto', white_back=True, density_noise=0.25)}


def load_render_opts_golden():
    """tests/golden/render_opts.npz (oracle/pin_renderer_options.py: the REFERENCE's ImportanceRenderer on seeded inputs, one run per
    option set) -> (inputs dict, {case: (opts, cams, rgb, depth, wsum)})."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'render_opts.npz'))
    Sc, Sf = int(g['Sc']), int(g['Sf'])
    inp = dict(planes=torch.from_numpy(g['planes']), jitter=torch.from_numpy(g['jitter']), u=torch.from_numpy(g['u']), R=int(g['R']), Sc=Sc, Sf=Sf,
               noise=(torch.from_numpy(g['noise_c']), torch.from_numpy(g['noise_f'])),
               P={k.replace('__', '.'): torch.from_numpy(g[k]) for k in g.files if k.startswith('decoder')})
    out = {}
    for name in [str(c) for c in g['cases']]:
        opts = dict(dict(depth_resolution=Sc, depth_resolution_importance=Sf, box_warp=1), **RENDER_OPT_CASES[name])
        out[name] = (opts, torch.from_numpy(g[name + '_cams']), torch.from_numpy(g[name + '_rgb']), torch.from_numpy(g[name + '_depth']),
                     torch.from_numpy(g[name + '_wsum']))
    inp['fine'] = {name: torch.from_numpy(g[name + '_fine']) for name in out}        # the REFERENCE's importance depths [N, R*R, Sf, 1] (round 6: teacher forcing)
    return inp, out


def test_renderer_options_match_reference_golden():
    """The rendering options beyond the ffhq configuration — 'auto' ray bounds (incl. the repair of rays that miss the box),
    disparity-space sampling, white_back, density noise — in oracle/renderer.py against the reference's own outputs."""
    inp, cases_ = load_render_opts_golden()
    N = inp['planes'].shape[0]
    for name, (opts, cams, rgb, depth, wsum) in cases_.items():
        ro, rd = renderer.ray_sampler(cams[:, :16].reshape(N, 4, 4), cams[:, 16:25].reshape(N, 3, 3), inp['R'])
        fine = []
        o = renderer.importance_renderer(inp['P'], 'decoder', inp['planes'], ro, rd, opts, inp['jitter'], inp['u'], noise=inp['noise'], fine_depths_out=fine)
        d = [float((a - b).abs().max()) for a, b in zip(o, (rgb, depth, wsum))]
        assert torch.equal(fine[0], inp['fine'][name])                   # the importance depths themselves: the reference's, bit for bit
        assert max(d) <= 1e-5, (name, d)
        if name == 'auto_wide_fov':
            rs, re = renderer.ray_limits_box(ro, rd, 1)
            assert 0.05 < float((re > rs).float().mean()) < 0.9        # the case does exercise the repair branch


# Tolerances of the float16 super-resolution route against the REFERENCE's own float16 run (tests/golden/*_fp16sr.npz: its float16
# SynthesisBlocks executed on the CPU with the off-GPU float32 guard disabled, oracle/pin_against_reference.py --fp16), max-abs /
# mean-abs on the 512x512 image (values up to ~8 with the synthetic weights; float16 ulp 2^-11 relative):
#   'cpu'  : bias_act rounded like the reference's off-GPU _bias_act_ref (float16 tensor ops) — only the accumulation order of the
#            half convolutions is left free (measured 2.4e-3 / 1.3e-4 oracle vs reference, 4.1e-3 / 2.8e-4 HIP kernels vs reference: their
#            per-sample weights additionally differ from ATen's by a float16 ulp on ~0.1 % of the entries — reduction order of the demodulation sum);
#   'cuda' : bias_act as bias_act.cu (float32 inside, one rounding: what the reference does on a GPU) — about one float16 ulp of a
#            hidden activation on 30 % of the elements away from the CPU run (measured 4.6e-3 / 6.4e-4).
FP16_SR_TOL = {'cpu': (6e-3, 4e-4), 'cuda': (1.2e-2, 1.2e-3)}


@pytest.mark.parametrize('case', ['case_r32_s24', 'case_r64_s48', 'case_r64_s48_b4'])
def test_oracle_fp16_superresolution_against_reference_fp16_run(case):
    """The oracle's float16 blocks (oracle/networks.py::synthesis_block_fp16) against the reference's own float16 branch: with the
    off-GPU bias_act rounding the oracle reproduces the reference's CPU run up to convolution accumulation order; the GPU-side
    rounding (bias_act.cu) — the one the HIP kernels implement by default — stays within the looser, stated bound."""
    from oracle import networks as ON
    g = np.load(os.path.join(GOLDEN, case + '_fp16sr.npz'))
    sd = spec.synthetic_state_dict(0, only=lambda n: n.startswith('superresolution'))
    ref = torch.from_numpy(g['image'])
    step = int(g['image_step']) if 'image_step' in g else 1          # the batch-4 fixture keeps every second pixel
    args = [torch.from_numpy(g[k]) for k in ('rgb_in', 'feat_in', 'ws_in')]
    for mode in ('cpu', 'cuda'):
        out = ON.superresolution(sd, 'superresolution', *args, force_fp32=False, cpu_rounding=(mode == 'cpu'))[..., ::step, ::step]
        d = (out - ref).abs()
        print(case, mode, f'max {float(d.max()):.3e} mean {float(d.mean()):.3e} (image absmax {float(ref.abs().max()):.2f})')
        assert float(d.max()) <= FP16_SR_TOL[mode][0] and float(d.mean()) <= FP16_SR_TOL[mode][1]
    out32 = ON.superresolution(sd, 'superresolution', *args, force_fp32=True)[..., ::step, ::step]
    assert float((out32 - ref).abs().mean()) > FP16_SR_TOL['cpu'][1]          # the float32 route is NOT within the tight bound


# Float16 blocks in the BACKBONES (num_fp16_res = 4, conv_clamp = 256: what legacy.load_network_pkl(force_fp16=True) builds) against the
# reference's own end-to-end float16 run on the CPU (tests/golden/*_fp16bb.npz, oracle/pin_against_reference.py --fp16-backbones).  Two
# implementations of such a network agree to about ONE float16 ulp of the largest activation of a stage (accumulation order moves
# values across rounding boundaries; 36 float16 layers deep the roundings decorrelate) — which is also the distance between the float16
# and the float32 route (5e-3 .. 8e-3 on the image).  Stage tolerance: 2.5 ulp (2^-10 relative) of the stage's largest value; image: the
# super-resolution route's bounds.
FP16_BB_STAGE_ULPS = 2.5
FP16_BB_IMAGE_TOL = (1.5e-2, 1.5e-3)       # max-abs, mean-abs on the 512 x 512 image
FP16_BB_RAW_TOL = 5e-3                     # max-abs on image_raw


def check_fp16_backbone_outputs(g, stages, out, who):
    """stages: dict textures / static_plane / mouths_plane / rendering_stitch (full resolution, float32), out: image / image_raw."""
    sub = {'textures': ('textures_sub4', 4), 'static_plane': ('static_plane_sub8', 8), 'mouths_plane': ('mouths_plane_sub4', 4),
           'rendering_stitch': ('rendering_stitch_sub4', 4)}
    absmax = dict(zip(('textures', 'static_plane', 'mouths_plane', 'rendering_stitch'), g['stage_absmax']))
    for k, (key, step) in sub.items():
        ref = torch.from_numpy(g[key])
        d = float((stages[k].float().cpu().reshape(ref.shape[0], ref.shape[1], *stages[k].shape[-2:])[..., ::step, ::step] - ref).abs().max())
        tol = FP16_BB_STAGE_ULPS * 2.0 ** -10 * float(absmax[k])
        print(f'{who} {k}: max-abs {d:.3e} (tolerance {tol:.3e} = {FP16_BB_STAGE_ULPS} float16 ulps of {float(absmax[k]):.1f})')
        assert d <= tol, (k, d, tol)
    d_raw = float((out['image_raw'].cpu() - torch.from_numpy(g['image_raw'])).abs().max())
    d_img = (out['image'].cpu()[..., ::2, ::2] - torch.from_numpy(g['image_sub2'])).abs()
    print(f'{who} image_raw max-abs {d_raw:.3e}; image max-abs {float(d_img.max()):.3e} mean {float(d_img.mean()):.3e}')
    assert d_raw <= FP16_BB_RAW_TOL and float(d_img.max()) <= FP16_BB_IMAGE_TOL[0] and float(d_img.mean()) <= FP16_BB_IMAGE_TOL[1]


def test_oracle_fp16_backbones_against_reference_fp16_run():
    """oracle/networks.py's emulation of float16 blocks inside the four backbones against the reference's own float16 run."""
    import os
    from next3d_amd import mesh, spec
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'case_r32_s24_fp16bb.npz'))
    d = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'demo_inputs.npz'))
    sd = spec.synthetic_state_dict(0)
    sd.update(mesh.mesh_buffers(d['faces'], d['uvs'], d['uvfaces']))
    R, Sc, Sf = int(g['R']), int(g['Sc']), int(g['Sf'])
    rk = dict(ogen.DEFAULT_RENDERING_KWARGS, depth_resolution=Sc, depth_resolution_importance=Sf)
    jitter, u = cases.rng_inputs(1, R, Sc, Sf)
    t = lambda k: torch.from_numpy(g[k])
    ws = ogen.mapping(sd, t('z'), t('c_cond'), rk, truncation_psi=float(g['psi']), truncation_cutoff=int(g['cutoff']))
    out, st = ogen.synthesis(sd, ws, t('c'), t('v'), mesh.synthetic_uv_face_mask(), rk, jitter, u, neural_rendering_resolution=R, return_stages=True,
                             force_fp32=False, net_kw=dict(fp16_resolution=32, conv_clamp=256, cpu_rounding=True))
    check_fp16_backbone_outputs(g, st, out, 'oracle')


@pytest.mark.parametrize('suffix', ['sr4X', 'cb16384', 'cb16384_cm256', 'sr2X_r64'])
def test_oracle_reproduces_reference_golden_of_other_architectures(suffix):
    """The reference's other architectures — super-resolution modules (tat/superresolution.py:29-124) and backbone widths (`--cbase` / `--cmax`,
    train_next3d.py:199-200) — run by the reference's own constructors (oracle/pin_against_reference.py --sr-modules / --channel-widths): the oracle
    restatement reproduces the committed outputs (the pin reported max-abs 0.0; here: the same comparison from the fixture)."""
    import os
    from next3d_amd import mesh, spec
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', f'case_r32_s24_{suffix}.npz'))
    d = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'demo_inputs.npz'))
    cls = str(g['sr_class'])
    cb, cm = (int(g['channel_base']), int(g['channel_max'])) if 'channel_base' in g.files else (32768, 512)
    sd = spec.synthetic_state_dict(0, sr=cls, channel_base=cb, channel_max=cm)
    sd.update(mesh.mesh_buffers(d['faces'], d['uvs'], d['uvfaces']))
    R, Sc, Sf = int(g['R']), int(g['Sc']), int(g['Sf'])
    rk = dict(ogen.DEFAULT_RENDERING_KWARGS, depth_resolution=Sc, depth_resolution_importance=Sf, superresolution_module='training_avatar_texture.superresolution.' + cls)
    jitter, u = cases.rng_inputs(1, R, Sc, Sf)
    t = lambda k: torch.from_numpy(g[k])
    ws = ogen.mapping(sd, t('z'), t('c_cond'), rk, truncation_psi=float(g['psi']), truncation_cutoff=int(g['cutoff']))
    assert float((ws - t('ws')).abs().max()) <= 1e-6
    out = ogen.synthesis(sd, ws, t('c'), t('v'), mesh.synthetic_uv_face_mask(), rk, jitter, u, neural_rendering_resolution=R)
    step = int(g['image_step'])
    assert float((out['image_raw'] - t('image_raw')).abs().max()) <= 1e-5 and float((out['image_depth'] - t('image_depth')).abs().max()) <= 1e-5
    assert float((out['image'][..., ::step, ::step] - t('image_sub')).abs().max()) <= 1e-5


@pytest.mark.parametrize('depth', [8, 1])
def test_oracle_mapping_depths_against_reference(depth):
    """mapping_kwargs.num_layers = 8 (the MappingNetwork's own default: what the reference builds when the key is absent) and 1: the oracle's ws against the
    reference's own (tests/golden/mapping_depth.npz, oracle/pin_against_reference.py --mapping-depth)."""
    import os
    from next3d_amd import spec
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'mapping_depth.npz'))
    sd = spec.synthetic_state_dict(0, only=lambda n: n.startswith('backbone.mapping'), mapping_layers=depth)
    ws = ogen.mapping(sd, torch.from_numpy(g['z']), torch.from_numpy(g['c_cond']), ogen.DEFAULT_RENDERING_KWARGS, truncation_psi=float(g['psi']), truncation_cutoff=int(g['cutoff']))
    assert float((ws - torch.from_numpy(g[f'ws_depth{depth}'])).abs().max()) <= 1e-6


def test_fp16_blocks_teacher_forced_oracle_vs_reference():
    """tests/golden/fp16_blocks.npz (round 5): every float16 block of the four backbones evaluated ALONE by the reference on a stated input.  The
    oracle's float16 emulation with the reference's off-GPU bias_act rounding reproduces each block almost bit for bit; the same emulation with
    bias_act.cu's single rounding and the reference's own float32 route both FAIL the bound — one block deep the comparison tells the arithmetic
    apart (the end-to-end float16 tolerances cannot: FP16_BB_IMAGE_TOL)."""
    import _fp16_blocks as fb
    from next3d_amd import spec
    from oracle import networks as onet
    sd = spec.synthetic_state_dict(0)
    blocks = fb.blocks()
    assert len(blocks) == 15 and {b['net'] for b in blocks} == set(fb.PREFIX)
    for b in blocks:
        x, img = onet.synthesis_block_fp16(sd, b['prefix'], b['x'].float(), b['img'], b['block_ws'], conv_clamp=256, cpu_rounding=True, noise_mode='const')
        same, mean_ulp, ie = fb.compare(x.half(), img, b)
        assert same >= fb.MIN_EQUAL and mean_ulp <= fb.MAX_MEAN_ULP and ie <= fb.IMG_TOL_ULP, (b['net'], b['res'], same, mean_ulp, ie)
        assert b['fp32_route'][0] < 0.5 and b['fp32_route'][1] > 5 * fb.MAX_MEAN_ULP, (b['net'], b['res'], b['fp32_route'])       # the float32 route fails it
    b = blocks[0]
    x, img = onet.synthesis_block_fp16(sd, b['prefix'], b['x'].float(), b['img'], b['block_ws'], conv_clamp=256, cpu_rounding=False, noise_mode='const')
    same, mean_ulp, _ = fb.compare(x.half(), img, b)
    assert same < 0.5 and mean_ulp > fb.MAX_MEAN_ULP                      # bias_act.cu's single rounding is ANOTHER arithmetic than this off-GPU golden
