"""CPU (-m "not gpu") dry run of the HOST side of the path: the generator's whole launch sequence is driven against a recording
stand-in for libn3d.so that marshals every argument exactly as ctypes would (so a wrong type, count or struct is caught) and
re-checks the preconditions the C entry points enforce (N3D_CHECK in conv2d_bf16x3.hip / conv2d.hip) — but launches nothing.
It covers configurations the GPU tests do not run: batch 8 per GPU (BASELINE.json configs[3]), batch 3, a 128² neural render
(no resize in front of the super-resolution, superresolution.py:282), coarse-only sampling, the cached call patterns.
No arithmetic happens here (outputs are uninitialised memory); numerics are the GPU tests' business."""
import os
import subprocess
import sys
from collections import Counter

import pytest
import torch


import _dryrun


@pytest.fixture
def dry(monkeypatch):
    pts, calls = _dryrun.patches()
    for obj, attr, val in pts:
        monkeypatch.setattr(obj, attr, val)
    return calls


def test_generator_built_and_called_under_inference_mode(dry):
    """A model constructed / loaded under torch.inference_mode() holds inference tensors, which track no version counter (`t._version` raises): the
    per-call parameter-update check, the operator layer's prepared-weight cache and the filter cache all have to cope (the scripts themselves
    run under torch.no_grad(), gen_samples_next3d.py:163)."""
    from next3d_amd import demo
    with torch.inference_mode():
        G, _ = demo.build_generator(torch.device('cpu'))
        G.overlap_static = False
        assert next(G.parameters()).is_inference()
        z, c, c_cond, v = demo.demo_batch([0])
        for _ in range(2):
            ws = G.mapping(z, c_cond, truncation_psi=0.7, truncation_cutoff=14)
            out = G.synthesis(ws, c, v, neural_rendering_resolution=32, noise_mode='const')
        assert tuple(out['image'].shape) == (1, 3, 512, 512)
    z, c, c_cond, v = demo.demo_batch([1])
    ws = G.mapping(z, c_cond)                                       # ... and outside the mode afterwards
    assert tuple(G.synthesis(ws, c, v, noise_mode='const')['image'].shape) == (1, 3, 512, 512)


@pytest.mark.parametrize('cb,cm,widths', [(16384, 512, {32: 512, 64: 256, 128: 128, 256: 64}), (16384, 256, {32: 256, 64: 256, 128: 128, 256: 64})])
def test_other_channel_widths_dry_run(dry, cb, cm, widths):
    """`--cbase` / `--cmax` other than the ffhq-512 pickle's 32768 / 512 (train_next3d.py:199-200; every backbone gets them as synthesis_kwargs,
    tat/networks_stylegan2.py:614): the constructor builds the reference's shapes, every launch of a forward marshals (the recorder checks pointers, sizes
    and layouts), one launch per layer as for the default widths; widths the matrix-core kernels do not tile raise at construction."""
    from next3d_amd import demo, spec
    with pytest.raises(RuntimeError, match='multiples of 64'):
        demo.build_generator(torch.device('cpu'), channel_base=8192)
    G, sd = demo.build_generator(torch.device('cpu'), channel_base=cb, channel_max=cm)
    G.overlap_static = False
    for net in ('texture_backbone', 'backbone', 'mouth_backbone', 'neural_blending'):
        for r, c in widths.items():
            assert sd[f'{net}.synthesis.b{r}.conv1.weight'].shape[:2] == (c, c)
    assert sd['superresolution.block0.conv0.weight'].shape[:2] == (256, 32)           # the super-resolution module ignores channel_base / channel_max
    z, c, c_cond, v = demo.demo_batch([0, 1])
    ws = G.mapping(z, c_cond, truncation_psi=0.7, truncation_cutoff=14)
    for kw in (dict(force_fp32=True), {}):
        dry.clear()
        out = G.synthesis(ws, c, v, neural_rendering_resolution=64, noise_mode='const', **kw)
        cnt = Counter(dry)
        assert tuple(out['image'].shape) == (2, 3, 512, 512) and cnt['n3d_render_rays_ex'] == 1 and cnt['n3d_rasterize_views'] == 1
        assert sum(cnt.values()) - cnt['n3d_split8_from_nchw'] - cnt['n3d_conv2d_prep_weight'] - cnt['n3d_conv2d_prep_weight_bf16x3'] <= 160, cnt


@pytest.mark.parametrize('N,R,Sc,Sf', [(1, 32, 24, 24), (3, 64, 48, 0), (8, 64, 48, 48), (2, 128, 96, 96)])
def test_generator_launch_sequence_dry_run(dry, N, R, Sc, Sf):
    from next3d_amd import demo
    rk = dict(demo.RENDERING_KWARGS, depth_resolution=Sc, depth_resolution_importance=Sf)
    G, _ = demo.build_generator(torch.device('cpu'), rendering_kwargs=rk)
    G.overlap_static = False
    z, c, c_cond, v = demo.demo_batch(list(range(N)))
    kw = dict(neural_rendering_resolution=R, noise_mode='const', force_fp32=True)
    ws = G.mapping(z, c_cond, truncation_psi=0.7, truncation_cutoff=14)
    assert tuple(ws.shape) == (N, 28, 512)
    G.synthesis(ws, c, v, **kw)                                  # first call: also prepares the weights
    dry.clear()
    out = G.synthesis(ws, c, v, cache_backbone=True, cache_identity=True, **kw)
    full = Counter(dry)
    assert tuple(out['image'].shape) == (N, 3, 512, 512) and tuple(out['image_raw'].shape) == (N, 3, R, R)
    assert tuple(out['image_depth'].shape) == (N, 1, R, R)
    assert full['n3d_conv2d_prep_weight'] == 0 and full['n3d_conv2d_prep_weight_bf16x3'] == 0      # prepared once per model
    assert full['n3d_rasterize_views'] == 1 and full['n3d_render_rays_ex'] == 1 and full['n3d_texture_project_planes'] == 1
    assert full['n3d_resize_aa'] == 2 and full['n3d_resize_aa_strided'] == (0 if R == 128 else 2)       # mouth crop + paste (+ feature / rgb resize unless R == 128: rgb is a channel-slice view, no copy)
    n_full = sum(full.values())
    # the launch count does not depend on the batch: one launch per layer, whatever N — plus the split8 conversion passes of
    # the pre-split path, whose number follows the layers' eligibility (n3d_conv2d_split8_eligible: batch and size dependent)
    assert full['n3d_fc_multi'] == 2                             # every style affine / demodulation coefficient of the five networks: two launches
    assert n_full - full['n3d_split8_from_nchw'] == 146 + (0 if R == 128 else 2), full        # (round 5: + n3d_unpack_inputs, which replaced four torch copies)
    assert full['n3d_unpack_inputs'] == 1 and full['n3d_blend_planes_views'] == 1
    assert full['n3d_fir4_split8'] <= 14 and full['n3d_split8_from_nchw'] <= 24
    dry.clear()
    G.synthesis(ws, c, v, use_cached_backbone=True, **kw)        # camera orbit: renderer + super-resolution only
    orbit = Counter(dry)
    assert orbit['n3d_rasterize_views'] == 0 and orbit['n3d_render_rays_ex'] == 1 and sum(orbit.values()) < n_full // 4
    dry.clear()
    G.synthesis(ws, c, v, use_cached_identity=True, **kw)        # reenactment: no texture / static backbone
    reenact = Counter(dry)
    assert reenact['n3d_rasterize_views'] == 1 and sum(orbit.values()) < sum(reenact.values()) < n_full
    dry.clear()
    pts = torch.zeros(N, 1000, 3)
    smp = G.sample_mixed(pts, None, ws, v, noise_mode='const', use_cached_backbone=True)
    assert tuple(smp['rgb'].shape) == (N, 1000, 32) and tuple(smp['sigma'].shape) == (N, 1000, 1) and dry == ['n3d_sample_points']
    dry.clear()
    smp = G.sample_mixed(pts, None, ws, v)                        # noise_mode defaults to 'random' as in the reference (triplane_next3d.py:278 -> :311): runs, rebuilding the planes
    assert tuple(smp['sigma'].shape) == (N, 1000, 1) and dry[-1] == 'n3d_sample_points' and len(dry) > 100
    dry.clear()
    G.synthesis(ws, c, v, neural_rendering_resolution=R)         # noise_mode defaults to 'random' (the reference's default,
    rnd = Counter(dry)                                           # networks_stylegan2.py:311): noisy layers run sample by sample
    # the default call runs the float16 super-resolution blocks on the f16 kernels: 10 launches (cast to h8, one weight
    # modulation for all six layers, per block transposed conv + FIR + conv + toRGB) instead of the float32 route's 8 (+ 2 conversion passes)
    # (the last block's toRGB is evaluated in its conv1's epilogue — layers.FUSED_TORGB: one n3d_torgb_h8, one n3d_rgb_combine)
    # round 5: the BACKBONES' 32-colour toRGB layers fuse the same way where conv1 runs on the pre-split kernel (one n3d_rgb_combine instead of one 1x1
    # launch: the total does not move); their number depends on batch and size — not with random noise (those layers run sample by sample)
    nb = full['n3d_rgb_combine'] - 2
    assert 0 <= nb <= 12 and rnd['n3d_rgb_combine'] == 1
    sr16 = lambda cnt, fused=0: (cnt['n3d_cast_h8'], cnt['n3d_modulate_weights_f16_multi'], cnt['n3d_conv2d_f16'], cnt['n3d_fir4_h8'], cnt['n3d_torgb_h8'] + cnt['n3d_rgb_combine'] - fused)
    assert sr16(rnd) == (1, 1, 4, 2, 2) and sr16(full, nb) == (0, 0, 0, 0, 2)      # (float32 route: both blocks' toRGB fused into their conv1: two combine launches)
    n_rnd = sum(rnd.values()) - 10 + 8 - rnd['n3d_split8_from_nchw'] + full['n3d_split8_from_nchw']
    assert n_rnd > n_full if N > 1 else n_rnd == n_full
    dry.clear()
    G.synthesis(ws, c, v, neural_rendering_resolution=R, noise_mode='const')      # no force_fp32: the reference's default, fp16
    half = Counter(dry)                                                          # super-resolution blocks (sr_num_fp16_res = 4)
    assert sr16(half, nb) == (1, 1, 4, 2, 2)
    assert sum(half.values()) - half['n3d_split8_from_nchw'] == 146 + (0 if R == 128 else 2) - 8 + 10
    # the switches that once made the default call raise: strict-fp32 arithmetic, no pre-split hand-off -> the float16
    # blocks run on their own kernels whatever the float32 layers use; random super-resolution noise -> the float16 kernels with the noisy part of each
    # layer sample by sample (one noise image per launch: conv1 + the FIR behind conv0 run N times, the transposed convolutions once), never an error
    import warnings
    from next3d_amd import layers
    for attr, val in (('PRECISION', 'fp32'), ('PRESPLIT', False)):
        old_attr = getattr(layers, attr)
        try:
            setattr(layers, attr, val)
            dry.clear()
            out = G.synthesis(ws, c, v, neural_rendering_resolution=R, noise_mode='const')
            assert tuple(out['image'].shape) == (N, 3, 512, 512) and sr16(Counter(dry)) == (1, 1, 4, 2, 2)      # (no backbone toRGB fuses under either switch)
        finally:
            setattr(layers, attr, old_attr)
    G.rendering_kwargs['superresolution_noise_mode'] = 'random'
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        dry.clear()
        out = G.synthesis(ws, c, v, neural_rendering_resolution=R, noise_mode='const')
    G.rendering_kwargs['superresolution_noise_mode'] = 'none'
    assert tuple(out['image'].shape) == (N, 3, 512, 512) and sr16(Counter(dry), nb) == (1, 1, 2 + 2 * N, 2 * N, 2)
    with pytest.raises(RuntimeError):
        G.synthesis(ws, c, v, neural_rendering_resolution=R, noise_mode='fancy')
    # sr_antialias=False (never set by train_next3d.py): fine for an up-scaling resize (same taps), refused for a down-scaling one (no plain-bilinear kernel)
    G.rendering_kwargs['sr_antialias'] = False
    try:
        if R <= 128:
            G.synthesis(ws, c, v, neural_rendering_resolution=R, noise_mode='const')
        with pytest.raises(RuntimeError, match='sr_antialias'):
            G.synthesis(ws, c, v, neural_rendering_resolution=160, noise_mode='const')
    finally:
        G.rendering_kwargs['sr_antialias'] = True
        G.neural_rendering_resolution = R


@pytest.mark.skipif(not os.path.isdir('/root/reference/training_avatar_texture'), reason='needs the reference tree (build container only)')
def test_reference_networks_drive_this_operator_layer_dry_run():
    """B1 at the call-pattern level: the REFERENCE's own SynthesisNetwork (StyleGAN2 texture backbone, fused
    modulated convolutions = grouped conv2d_resample calls, up-sampling layers, toRGB + skip upsample), its StyleUNet and its
    super-resolution module (superresolution.py:264-290, conv_clamp 256, 512² output) run on
    this package's `torch_utils.ops.*` after install_dropin(), against the recording stand-in: every call the reference's network
    code makes is accepted by this operator layer and ends in well-formed libn3d.so launches."""
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r); sys.path.insert(0, '/root/reference')
import torch
from collections import Counter
from next3d_amd import mesh
from oracle import ref_shims
ref_shims.install(mesh.synthetic_uv_face_mask()[0, 0].numpy())
import next3d_amd
next3d_amd.install_dropin()
import _dryrun
pts, calls = _dryrun.patches()
for obj, attr, val in pts:
    setattr(obj, attr, val)
from training_avatar_texture import networks_stylegan2 as ns, networks_stylegan2_styleunet as nu
assert ns.conv2d_resample.__name__.startswith('next3d_amd.') and nu.upfirdn2d.__name__.startswith('next3d_amd.')
with torch.no_grad():
    net = ns.SynthesisNetwork(w_dim=512, img_resolution=64, img_channels=32, channel_base=32768, channel_max=512, num_fp16_res=0,
                              fused_modconv_default='inference_only').eval().requires_grad_(False)
    img = net(torch.randn(2, net.num_ws, 512), noise_mode='const')
    assert tuple(img.shape) == (2, 32, 64, 64), img.shape
    c1 = Counter(calls)
    # ONE launch per layer for the whole batch (the fused modulated convolution's groups = batch call: per-sample weights through
    # n3d_conv2d_desc.wt_batch_stride) + one re-tile launch per layer: 9 convolutions + 5 toRGB at 64 x 64
    assert c1['n3d_conv2d'] + c1['n3d_conv2d_bf16x3'] == 2 * 5 - 1 + 5 == c1['n3d_conv2d_prep_weight_grouped'], c1
    assert c1['n3d_upfirdn2d'] + c1['n3d_upfirdn2d_pitched'] >= 4, c1
    calls.clear()
    unet = nu.SynthesisNetwork(w_dim=512, img_resolution=64, img_channels=32, in_size=64, final_size=16, cond_channels=32, num_cond_res=64,
                               channel_base=32768, channel_max=512, num_fp16_res=0, fused_modconv_default='inference_only').eval().requires_grad_(False)
    out = unet(torch.randn(1, 32, 64, 64), torch.randn(1, unet.num_ws, 512), noise_mode='const')
    assert tuple(out.shape) == (1, 32, 64, 64), out.shape
    c2 = Counter(calls)
    assert c2['n3d_conv2d'] + c2['n3d_conv2d_bf16x3'] > 10 and c2['n3d_upfirdn2d'] + c2['n3d_upfirdn2d_pitched'] > 2, c2
    calls.clear()
    from training_avatar_texture import superresolution as rsr
    srn = rsr.SuperresolutionHybrid8XDC(channels=32, img_resolution=512, sr_num_fp16_res=4, sr_antialias=True, channel_base=32768,
                                        channel_max=512, fused_modconv_default='inference_only').eval().requires_grad_(False)
    big = srn(torch.randn(1, 3, 64, 64), torch.randn(1, 32, 64, 64), torch.randn(1, 14, 512), noise_mode='none')
    assert tuple(big.shape) == (1, 3, 512, 512), big.shape
    c3 = Counter(calls)
    assert c3['n3d_conv2d'] + c3['n3d_conv2d_bf16x3'] == 6 and c3['n3d_upfirdn2d'] + c3['n3d_upfirdn2d_pitched'] == 4, c3    # 2 x (conv0 up, conv1, toRGB); 2 up FIRs + 2 skip upsamples
print('ok', dict(c1), dict(c2), dict(c3))
""" % (repo, os.path.join(repo, 'tests'))
    r = subprocess.run([sys.executable, '-c', code], cwd='/tmp', capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().startswith('ok'), r.stdout[-2000:] + r.stderr[-4000:]


def test_operator_layer_dtype_handling_dry_run(dry):
    """B1 takes float32 or float16 (fp16 tensors are converted by n3d_cast around the fp32-accumulating kernels: the call
    sequence shows the conversions); anything else — or an input / weight dtype mismatch — raises RuntimeError before any
    pointer could reach a float32 kernel (fp16 pointers used to be passed to fp32 kernels unchecked)."""
    from next3d_amd.torch_utils.ops import conv2d_gradfix as cg, conv2d_resample, filtered_lrelu, fma, upfirdn2d
    x, w = torch.randn(2, 16, 12, 12), torch.randn(8, 16, 3, 3)
    f = upfirdn2d.setup_filter([1, 3, 3, 1])
    y = conv2d_resample.conv2d_resample(x.half().contiguous(memory_format=torch.channels_last), w.half(), f=f, up=2, padding=1, flip_weight=False)
    assert y.dtype == torch.float16 and tuple(y.shape) == (2, 8, 24, 24)
    # float16 operands outside the f16 kernels' shapes (O % 64 != 0): widened on the device, multiplied on the split-bf16 kernels (exact
    # for float16 operands; the float16 weights are re-tiled directly), stored as float16
    assert dry == ['n3d_cast', 'n3d_conv2d_prep_weight_grouped', 'n3d_conv2d_bf16x3', 'n3d_cast', 'n3d_cast', 'n3d_upfirdn2d_pitched', 'n3d_cast']
    dry.clear()
    # the fused modulated convolution of a reference fp16 block: groups = batch, per-sample float16 weights -> the f16 matrix-core kernels
    xg, wg = torch.randn(1, 3 * 32, 20, 40).half(), torch.randn(3 * 64, 32, 3, 3).half()
    y = cg.conv2d(xg, wg, padding=1, groups=3)
    assert y.dtype == torch.float16 and tuple(y.shape) == (1, 3 * 64, 20, 40)
    assert dry == ['n3d_conv2d_prep_weight_grouped', 'n3d_cast_h8_ex', 'n3d_conv2d_f16', 'n3d_cast_h8_ex']
    dry.clear()
    y = cg.conv_transpose2d(xg, torch.randn(3 * 32, 64, 3, 3).half(), stride=2, groups=3)
    assert y.dtype == torch.float16 and tuple(y.shape) == (1, 3 * 64, 41, 81)
    assert dry == ['n3d_conv2d_prep_weight_grouped', 'n3d_cast_h8_ex', 'n3d_conv2d_f16', 'n3d_cast_h8_ex']
    dry.clear()
    y = cg.conv2d(xg, torch.randn(3 * 3, 32, 1, 1).half(), groups=3)               # toRGB of such a block: [N*3, I, 1, 1]
    assert y.dtype == torch.float16 and tuple(y.shape) == (1, 9, 20, 40) and dry == ['n3d_cast_h8_ex', 'n3d_torgb_h8', 'n3d_cast']
    dry.clear()
    # float32, groups = batch: one launch, per-sample weights; a persistent weight tensor with groups == 1 is re-tiled once
    y = cg.conv2d(xg.float(), wg.float(), padding=1, groups=3)
    assert y.dtype == torch.float32 and tuple(y.shape) == (1, 3 * 64, 20, 40) and dry == ['n3d_conv2d_prep_weight_grouped', 'n3d_conv2d_bf16x3']
    dry.clear()
    wp = torch.randn(8, 16, 3, 3)
    cg.conv2d(x, wp, padding=1); cg.conv2d(x, wp, padding=1)
    assert dry.count('n3d_conv2d_prep_weight_grouped') == 1 and dry.count('n3d_conv2d_bf16x3') == 2
    wp.mul_(2.0)                                                                   # in-place update: re-tiled
    cg.conv2d(x, wp, padding=1)
    assert dry.count('n3d_conv2d_prep_weight_grouped') == 2
    dry.clear()
    y = conv2d_resample.conv2d_resample(x, w, f=f, down=2, padding=1)
    assert y.dtype == torch.float32 and tuple(y.shape) == (2, 8, 6, 6) and 'n3d_cast' not in dry
    y = filtered_lrelu.filtered_lrelu(x.half(), fu=f, fd=f, b=torch.randn(16).half(), up=2, down=2, padding=[3, 2, 3, 2])
    assert y.dtype == torch.float16 and tuple(y.shape) == (2, 16, 12, 12) and dry.count('n3d_filtered_lrelu') == 1
    for bad in (lambda: conv2d_resample.conv2d_resample(x.double(), w.double(), padding=1),
                lambda: cg.conv2d(x.half(), w, padding=1), lambda: cg.conv2d(x, w.half(), padding=1),
                lambda: cg.conv_transpose2d(x.half(), w.transpose(0, 1), stride=2),
                lambda: cg.conv_launch(x.to(torch.bfloat16), cg.prep_weight(w), 3, 0, 8),
                lambda: cg.conv_launch(x, cg.prep_weight(w).half(), 3, 0, 8),
                lambda: cg.prep_weight(w.to(torch.bfloat16)), lambda: upfirdn2d.upfirdn2d(x.double(), f),
                lambda: upfirdn2d.upfirdn2d(x, f.double()), lambda: fma.fma(x.half(), x, x),
                lambda: filtered_lrelu.filtered_lrelu(x.to(torch.bfloat16))):
        with pytest.raises(RuntimeError):
            bad()


@pytest.mark.skipif(not os.path.isdir('/root/reference/training_avatar_texture'), reason='needs the reference tree (build container only)')
def test_unreloaded_reference_generator_runs_on_op_layer_and_third_party_shims_dry_run():
    """Boundary B1 end to end: the REFERENCE's own TriPlaneGenerator — constructed, not reloaded — with
    install_dropin(third_party=True): its `torch_utils.ops.*`, `pytorch3d.{io.load_obj, structures.Meshes, renderer.mesh.rasterize_meshes}`
    and `cv2.{imread, floodFill}` all resolve to this package, and one full `synthesis` (its Python: rasterize -> Pytorch3dRasterizer.forward
    -> fill_mouth -> mouth crop / paste -> StyleUNets -> renderer -> super-resolution, float16 blocks included) runs against the
    recording stand-in of libn3d.so: every call is accepted and marshalled (no arithmetic here; the shims' kernels are checked
    bit for bit on the GPU, tests/test_path_kernels_gpu.py::test_third_party_shims_*)."""
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
from collections import Counter
from next3d_amd import mesh, spec, demo
from oracle import ref_shims
mask = mesh.synthetic_uv_face_mask()
ref_shims.install(mask[0, 0].numpy(), third_party=False)
import next3d_amd
next3d_amd.install_dropin(third_party=True)
import cv2, pytorch3d
assert cv2.__name__ == 'next3d_amd.shims.cv2' and pytorch3d.__name__ == 'next3d_amd.shims.pytorch3d'
cv2.imread = lambda path, *a: np.stack([(mask[0, 0].numpy() * 255).round().astype(np.uint8)] * 3, -1)     # data/ffhq/uv_face_eye_mask.png is not in the tree
cv2._device = lambda: torch.device('cpu')
import _dryrun
pts, calls = _dryrun.patches()
for obj, attr, val in pts:
    setattr(obj, attr, val)
_empty = torch.empty
torch.empty = lambda *a, **k: _empty(*a, **k).zero_()          # un-launched kernels leave their outputs untouched: keep gather indices valid
from oracle.pin_against_reference import RENDERING_KWARGS
G = ref_shims.build_reference_generator(dict(RENDERING_KWARGS, depth_resolution=12, depth_resolution_importance=12))
import training_avatar_texture.volumetric_rendering.renderer as vr
assert vr.rasterize_meshes.__module__.startswith('next3d_amd.shims') and vr.Meshes.__module__.startswith('next3d_amd.shims') and vr.cv2 is cv2
z, c, c_cond, v = demo.demo_batch([0, 1])
with torch.no_grad():
    ws = G.mapping(z, c_cond, truncation_psi=0.7, truncation_cutoff=14)
    calls.clear()
    out = G.synthesis(ws, c, v, neural_rendering_resolution=32, noise_mode='const')
cnt = Counter(calls)
assert tuple(out['image'].shape) == (2, 3, 512, 512) and tuple(out['image_raw'].shape) == (2, 3, 32, 32)
assert cnt['n3d_rasterize_meshes'] == 4 and cnt['n3d_flood_fill'] == 4 * 2, cnt          # 4 views; fill_mouth loops over the batch
assert cnt['n3d_conv2d'] + cnt['n3d_conv2d_bf16x3'] > 60 and cnt['n3d_upfirdn2d'] + cnt['n3d_upfirdn2d_pitched'] > 20, cnt
print('B1_DRY_RUN_OK', sum(cnt.values()))
""" % (repo, os.path.join(repo, 'tests'))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and 'B1_DRY_RUN_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_b1_route_driver_dry_run():
    """oracle/b1_route.py (the reference's code pattern on the operator layer + shims: what tests/test_b1_route_gpu.py and bench.py's
    `b1_route` extra run on the GPU box, where /root/reference does not exist) against the recording stand-in of libn3d.so: the whole
    forward — fused modulated convolutions as groups = batch calls, the float16 super-resolution blocks on real half tensors, the
    rasteriser / flood-fill shims — is accepted and marshalled; one launch per convolution layer whatever the batch."""
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np, torch
from collections import Counter
from next3d_amd import demo, mesh, spec
import next3d_amd.shims.cv2 as cv2
cv2._device = lambda: torch.device('cpu')
import _dryrun
pts, calls = _dryrun.patches()
for obj, attr, val in pts:
    setattr(obj, attr, val)
_empty = torch.empty
torch.empty = lambda *a, **k: _empty(*a, **k).zero_()          # un-launched kernels leave their outputs untouched: keep gather indices valid
from oracle import b1_route, cases
route = b1_route.Route('cpu')
P = spec.synthetic_state_dict(0)
d = demo.demo_arrays()
P.update(mesh.mesh_buffers(d['faces'], d['uvs'], d['uvfaces']))
rk = dict(demo.RENDERING_KWARGS, depth_resolution=12, depth_resolution_importance=12)
z, c, c_cond, v = demo.demo_batch([0, 1])
jitter, u = cases.rng_inputs(2, 32, 12, 12)
mask = torch.nn.functional.interpolate(mesh.synthetic_uv_face_mask().float(), [256, 256])
with torch.no_grad():
    ws = route.mapping(P, z, c_cond, rk, truncation_psi=0.7, truncation_cutoff=14)
    assert tuple(ws.shape) == (2, 28, 512)
    calls.clear()
    out = route.synthesis(P, ws, c, v, mask, rk, jitter, u, neural_rendering_resolution=32, force_fp32=False)
cnt = Counter(calls)
assert tuple(out['image'].shape) == (2, 3, 512, 512) and out['image'].dtype == torch.float32 and tuple(out['image_raw'].shape) == (2, 3, 32, 32)
assert cnt['n3d_rasterize_meshes'] == 4 and cnt['n3d_flood_fill'] == 4 * 2, cnt
convs = cnt['n3d_conv2d'] + cnt['n3d_conv2d_bf16x3'] + cnt['n3d_conv2d_f16'] + cnt['n3d_torgb_h8']
assert cnt['n3d_conv2d_f16'] == 4 and cnt['n3d_torgb_h8'] == 2 and cnt['n3d_cast_h8_ex'] == 2 * 4 + 2, cnt     # the two float16 blocks
assert convs == cnt['n3d_bias_act'] == 101, (convs, cnt)     # one launch per convolution layer of the five networks (each followed by its bias_act), whatever the batch
print('B1_ROUTE_DRY_RUN_OK', convs, sum(cnt.values()))
""" % (repo, os.path.join(repo, 'tests'))
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and 'B1_ROUTE_DRY_RUN_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_fp16_backbones_launch_sequence_dry_run(dry):
    """num_fp16_res = 4 / conv_clamp = 256 (legacy.load_network_pkl(force_fp16=True), legacy.py:49-59): the blocks of resolution >= 32
    of the four backbones run on the f16 kernels — texture / static backbone 4 blocks each, mouth StyleUNet 4, blending StyleUNet 3
    (b64-b256: its b32 does not exist) — and force_fp32=True gives the float32 launch sequence back."""
    from next3d_amd import demo, mesh
    from next3d_amd.generator import TriPlaneGenerator
    d = demo.demo_arrays()
    G = TriPlaneGenerator(512, 25, 512, 512, 3, (d['faces'], d['uvs'], d['uvfaces']), sr_num_fp16_res=4, mapping_kwargs=dict(num_layers=2),
                          rendering_kwargs=dict(demo.RENDERING_KWARGS, depth_resolution=12, depth_resolution_importance=12),
                          sr_kwargs=dict(channel_base=32768, channel_max=512), uv_face_mask=mesh.synthetic_uv_face_mask(), channel_base=32768,
                          channel_max=512, num_fp16_res=4, conv_clamp=256)
    assert G.backbone_fp16_resolution == 32 and G.backbone_conv_clamp == 256
    G.overlap_static = False
    z, c, c_cond, v = demo.demo_batch([0, 1])
    ws = G.mapping(z, c_cond, truncation_psi=0.7, truncation_cutoff=14)
    kw = dict(neural_rendering_resolution=32, noise_mode='const')
    G.synthesis(ws, c, v, **kw)
    dry.clear()
    out = G.synthesis(ws, c, v, **kw)
    cnt = Counter(dry)
    blocks = 4 + 4 + 4 + 3
    assert tuple(out['image'].shape) == (2, 3, 512, 512)
    # (the super-resolution's last toRGB is evaluated in its conv1's epilogue — layers.FUSED_TORGB: n3d_rgb_combine instead of n3d_torgb_h8)
    assert cnt['n3d_conv2d_f16'] == 2 * (blocks + 2) and cnt['n3d_torgb_h8'] + cnt['n3d_rgb_combine'] == blocks + 2 == cnt['n3d_fir4_h8'] and cnt['n3d_rgb_combine'] == 1, cnt
    assert cnt['n3d_modulate_weights_f16_multi'] == 2 * 3 + 2 + 1      # 12 layers = two launches per 4-block network, 9 = two for the blending net, one for the SR
    dry.clear()
    G.synthesis(ws, c, v, force_fp32=True, **kw)
    cnt32 = Counter(dry)
    assert cnt32['n3d_conv2d_f16'] == 0 and cnt32['n3d_cast_h8'] == 0
    dry.clear()
    G.synthesis(ws, c, v, neural_rendering_resolution=32, noise_mode='random')      # random noise (the reference's default): the float16 blocks stay on the f16 kernels,
    rnd = Counter(dry)                                                              # their noisy parts sample by sample (one noise image per launch), never an error
    # per block: the transposed convolution once for the batch + its FIR per sample, conv1 per sample; the super-resolution blocks have noise_mode 'none' (4 launches)
    assert rnd['n3d_conv2d_f16'] == blocks * (1 + 2) + 4 and rnd['n3d_fir4_h8'] == blocks * 2 + 2, rnd
