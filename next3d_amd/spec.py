"""Parameter / buffer inventory of the Next3D TriPlaneGenerator and seeded synthetic weights.

The names and shapes reproduce the reference module tree (training_avatar_texture/
triplane_next3d.py:63-109 and the constructors it calls) so that a reference ``state_dict`` /
``misc.copy_params_and_buffers(require_all=True)`` (gen_samples_next3d.py:154) maps one-to-one
onto this package's generator.  tests/golden/ref_state_dict_spec.txt holds the list produced
by the reference's own constructors; tests/test_spec.py diffs the two.

There is no network and no pretrained pickle (README.md:40 is a Drive link), so benchmarks and
parity tests run on *seeded synthetic* weights: `synthetic_state_dict(seed)`.
"""
import hashlib
from collections import OrderedDict

import numpy as np
import torch

W_DIM = 512
Z_DIM = 512
C_DIM = 25
PLANE_RES = 256
N_VERTS, N_UVS, N_FACES = 5023, 5118, 9976          # FLAME topology (data/demo/demo.obj)


def channels_dict(img_resolution, channel_base=32768, channel_max=512):
    log2 = int(np.log2(img_resolution))
    return {2 ** i: min(channel_base // 2 ** i, channel_max) for i in range(2, log2 + 1)}


def _fir():
    f = torch.tensor([1., 3., 3., 1.])
    f = torch.outer(f, f)
    return f / f.sum()


def _mapping(spec, p, num_ws, num_layers=2):
    """MappingNetwork (tat/networks_stylegan2.py:200-231): embed, then `num_layers` FC layers 1024 -> 512 -> ... -> 512 (num_layers: `mapping_kwargs`, train_next3d.py
    map_depth = 2 for every next3d configuration; the class's own default is 8)."""
    spec[f'{p}.w_avg'] = ((W_DIM,), 'w_avg')
    spec[f'{p}.embed.weight'] = ((W_DIM, C_DIM), 'randn')
    spec[f'{p}.embed.bias'] = ((W_DIM,), 'bias')
    for i in range(num_layers):
        spec[f'{p}.fc{i}.weight'] = ((W_DIM, Z_DIM + W_DIM if i == 0 else W_DIM), 'randn_lr')     # stored / lr_multiplier (0.01)
        spec[f'{p}.fc{i}.bias'] = ((W_DIM,), 'bias')


def _synth_layer(spec, p, ic, oc, res, k=3):
    spec[f'{p}.weight'] = ((oc, ic, k, k), 'randn')
    spec[f'{p}.noise_strength'] = ((), 'noise_strength')
    spec[f'{p}.bias'] = ((oc,), 'bias')
    spec[f'{p}.resample_filter'] = ((4, 4), 'fir')
    spec[f'{p}.noise_const'] = ((res, res), 'randn')
    spec[f'{p}.affine.weight'] = ((ic, W_DIM), 'randn')
    spec[f'{p}.affine.bias'] = ((ic,), 'affine_bias')


def _torgb(spec, p, ic, oc):
    spec[f'{p}.weight'] = ((oc, ic, 1, 1), 'randn')
    spec[f'{p}.bias'] = ((oc,), 'bias')
    spec[f'{p}.affine.weight'] = ((ic, W_DIM), 'randn')
    spec[f'{p}.affine.bias'] = ((ic,), 'affine_bias')


def _block(spec, p, ic, oc, res, img_channels):
    if ic == 0:
        spec[f'{p}.const'] = ((oc, res, res), 'randn')
    spec[f'{p}.resample_filter'] = ((4, 4), 'fir')
    if ic != 0:
        _synth_layer(spec, f'{p}.conv0', ic, oc, res)
    _synth_layer(spec, f'{p}.conv1', oc, oc, res)
    _torgb(spec, f'{p}.torgb', oc, img_channels)


def _conv2d_layer(spec, p, ic, oc, k, bias=True):
    spec[f'{p}.weight'] = ((oc, ic, k, k), 'randn')
    if bias:
        spec[f'{p}.bias'] = ((oc,), 'bias')
    spec[f'{p}.resample_filter'] = ((4, 4), 'fir')


def _synthesis(spec, p, img_channels, img_resolution=PLANE_RES, channel_base=32768, channel_max=512):
    cd = channels_dict(img_resolution, channel_base, channel_max)
    for res in sorted(cd):
        _block(spec, f'{p}.b{res}', cd[res // 2] if res > 4 else 0, cd[res], res, img_channels)
    return cd


def _styleunet(spec, p, img_channels, cond_channels, in_size, final_size, channel_base=32768, channel_max=512):
    cd = _synthesis(spec, p, img_channels, PLANE_RES, channel_base, channel_max)
    enc_res = [2 ** i for i in range(int(np.log2(in_size)), int(np.log2(final_size)) - 1, -1)]
    for i, res in enumerate(enc_res[:-1]):
        e = f'{p}.encoder.{i}'
        spec[f'{e}.resample_filter'] = ((4, 4), 'fir')
        _conv2d_layer(spec, f'{e}.fromrgb', cond_channels, cd[res], 1, bias=False)
        _conv2d_layer(spec, f'{e}.conv1', cd[res], cd[res], 3)
        _conv2d_layer(spec, f'{e}.conv2', cd[res], cd[res // 2], 3)
    for i, res in enumerate(enc_res[::-1]):
        nc = cd[res]
        _conv2d_layer(spec, f'{p}.fusion.{i}', nc * 2 if res > final_size else nc, nc, 3)


# The reference's super-resolution modules (tat/superresolution.py): class name -> (img_resolution, input_resolution, resize rule of forward —
# 'ne': resize when the render differs from input_resolution (:48,127,282), 'lt': only when it is smaller (4X, :82) —,
# [(block kind, in channels, out channels, block resolution)]: 'up' = SynthesisBlock, 'noup' = SynthesisBlockNoUp (:158), module-level resample_filter buffer?)
SR_MODULES = {
    'SuperresolutionHybrid8XDC': (512, 128, 'ne', [('up', 32, 256, 256), ('up', 256, 128, 512)], False),     # :264-290 (next3d_ffhq_512)
    'SuperresolutionHybrid8X': (512, 128, 'ne', [('up', 32, 128, 256), ('up', 128, 64, 512)], True),          # :29-58
    'SuperresolutionHybrid4X': (256, 128, 'lt', [('noup', 32, 128, 128), ('up', 128, 64, 256)], True),        # :62-91
    'SuperresolutionHybrid2X': (128, 64, 'ne', [('noup', 32, 128, 64), ('up', 128, 64, 128)], True),          # :95-124
}
DEFAULT_SR = 'SuperresolutionHybrid8XDC'


def sr_module(name):
    """`rendering_kwargs['superresolution_module']` (a dotted class path) -> (class name, SR_MODULES entry); RuntimeError for anything else
    (SuperresolutionHybridDeepfp32 raises in the reference too: triplane_next3d.py:66 passes it `sr_antialias`, which it hands on to SynthesisLayer -> TypeError)."""
    cls = str(name or DEFAULT_SR).rsplit('.', 1)[-1]
    if cls not in SR_MODULES:
        raise RuntimeError(f'superresolution_module {name!r}: implemented are ' + ', '.join(SR_MODULES))
    return cls, SR_MODULES[cls]


def check_channels(channel_base=32768, channel_max=512):
    """The four 256 x 256 backbones' channels_dict for `channel_base` / `channel_max` (train_next3d.py:199-200 --cbase / --cmax; ffhq-512: 32768 / 512).
    The split-bf16 / f16 matrix-core kernels tile 64 output channels per workgroup: every width must be a multiple of 64, at most 512."""
    cd = channels_dict(PLANE_RES, int(channel_base), int(channel_max))
    if any(c % 64 or not 64 <= c <= 512 for c in cd.values()):
        raise RuntimeError(f'channel_base={channel_base} / channel_max={channel_max} gives block widths {cd}: this build runs widths that are multiples of 64 in 64..512 '
                           '(e.g. channel_base 32768 or 16384, channel_max 512 or 256)')
    return cd


def build_spec(sr=DEFAULT_SR, channel_base=32768, channel_max=512, mapping_layers=2):
    """name -> (shape, kind) for every parameter and buffer of TriPlaneGenerator (`sr`: the super-resolution class, SR_MODULES; channel_base /
    channel_max: the `synthesis_kwargs` every backbone receives, triplane_next3d.py:63-65,109 — the super-resolution modules ignore theirs)."""
    cb, cm = int(channel_base), int(channel_max)
    spec = OrderedDict()
    # texture_backbone: StyleGAN2 256², 32 ch (triplane_next3d.py:63)
    _synthesis(spec, 'texture_backbone.synthesis', 32, PLANE_RES, cb, cm)
    _mapping(spec, 'texture_backbone.mapping', 14, mapping_layers)
    # mouth_backbone: StyleUNet 64² -> 256², final 4 (:64)
    _styleunet(spec, 'mouth_backbone.synthesis', 32, 32, 64, 4, cb, cm)
    _mapping(spec, 'mouth_backbone.mapping', 14, mapping_layers)
    # backbone: StyleGAN2 256², 96 ch, mapping broadcasts to 28 ws (:65)
    _synthesis(spec, 'backbone.synthesis', 96, PLANE_RES, cb, cm)
    _mapping(spec, 'backbone.mapping', 28, mapping_layers)
    # superresolution (superresolution.py:29-124, :264-277): two blocks, every layer a SynthesisLayer of the block's resolution, toRGB to 3 colours
    _, (_, _, _, sr_blocks, sr_filter) = sr_module(sr)
    for bi, (_, ic, oc, res) in enumerate(sr_blocks):
        spec[f'superresolution.block{bi}.resample_filter'] = ((4, 4), 'fir')
        _synth_layer(spec, f'superresolution.block{bi}.conv0', ic, oc, res)
        _synth_layer(spec, f'superresolution.block{bi}.conv1', oc, oc, res)
        _torgb(spec, f'superresolution.block{bi}.torgb', oc, 3)
    if sr_filter:
        spec['superresolution.resample_filter'] = ((4, 4), 'fir')
    # decoder: OSGDecoder (triplane_next3d.py:353-357)
    spec['decoder.net.0.weight'] = ((64, 32), 'randn')
    spec['decoder.net.0.bias'] = ((64,), 'bias')
    spec['decoder.net.2.weight'] = ((33, 64), 'randn')
    spec['decoder.net.2.bias'] = ((33,), 'bias')
    # mesh buffers (:86-103)
    n_dense = 2 * (PLANE_RES - 1 - 2 - 2) * (PLANE_RES - 1 - 5 - 5)
    spec['dense_faces'] = ((1, n_dense, 3), 'mesh')
    spec['faces'] = ((1, N_FACES, 3), 'mesh')
    spec['raw_uvcoords'] = ((1, N_UVS, 2), 'mesh')
    spec['uvcoords'] = ((1, N_UVS, 3), 'mesh')
    spec['uvfaces'] = ((1, N_FACES, 3), 'mesh')
    spec['face_uvcoords'] = ((1, N_FACES, 3, 3), 'mesh')
    # neural_blending: StyleUNet 256² -> 256², final 32 (:109)
    _styleunet(spec, 'neural_blending.synthesis', 32, 32, 256, 32, cb, cm)
    _mapping(spec, 'neural_blending.mapping', 14, mapping_layers)
    return spec


_INT_BUFFERS = ('dense_faces', 'faces', 'uvfaces')


def _seed_for(name, seed):
    h = hashlib.sha256(f'{seed}:{name}'.encode()).digest()
    return int.from_bytes(h[:7], 'little')


_WIDE_STYLE, _WIDE_TORGB = 3.0, 1.0 / 3.0


def synthetic_state_dict(seed=0, only=None, profile='unit', sr=DEFAULT_SR, channel_base=32768, channel_max=512, mapping_layers=2):
    """Seeded synthetic weights (CPU fp32).  Distributions follow the reference initialisers
    (randn weights, affine bias 1) except that biases, noise_strength and w_avg — zero at init in
    the reference — get small seeded non-zero values so those code paths are exercised
    (SURVEY.md §8c).  Each tensor has its own generator, so any subset is reproducible.
    Mesh buffers are NOT produced here (see next3d_amd.mesh.mesh_buffers).

    profile='wide' (VERDICT r4 item 4a): statistics closer to a TRAINED StyleGAN2 than unit-variance randn — heavy-tailed convolution /
    fully-connected weights (a normal draw times a log-normal factor, sigma 0.6, rescaled to unit variance: kurtosis ~ 12 instead of 3),
    style magnitudes x 3 (affine weights and biases; toRGB is not demodulated, so image magnitudes follow), noise strengths ~ 0.3,
    biases ~ 0.3 — a second, independent test of the split-bf16 arithmetic's error (the unit profile's 8.8e-5 is one draw)."""
    if profile not in ('unit', 'wide'):
        raise ValueError(profile)
    wide = profile == 'wide'
    out = OrderedDict()
    for name, (shape, kind) in build_spec(sr, channel_base, channel_max, mapping_layers).items():
        if kind == 'mesh' or (only is not None and not only(name)):
            continue
        if kind == 'fir':
            out[name] = _fir()
            continue
        g = torch.Generator().manual_seed(_seed_for(name if not wide else f'wide:{name}', seed))
        r = torch.randn(shape, generator=g, dtype=torch.float32)
        if wide and kind in ('randn', 'randn_lr'):
            tail = torch.exp(0.6 * torch.randn(shape, generator=g, dtype=torch.float32)) / float(np.exp(0.6 ** 2))       # E[(r e^{0.6 g})^2] = e^{0.72}
            r = r * tail
            if name.endswith('.affine.weight'):
                r = _WIDE_STYLE * r
            elif '.torgb.weight' in name:
                r = _WIDE_TORGB * r
        if kind == 'randn':
            t = r
        elif kind == 'randn_lr':
            t = r / 0.01
        elif kind == 'bias':
            t = (0.3 if wide else 0.1) * r
        elif kind == 'affine_bias':
            t = _WIDE_STYLE * (1.0 + 0.3 * r) if wide else 1.0 + 0.1 * r
        elif kind == 'noise_strength':
            t = (0.3 if wide else 0.1) * r
        elif kind == 'w_avg':
            t = 0.25 * r
        else:
            raise KeyError(kind)
        out[name] = t
    return out
