"""-m gpu parity of the float16-block kernels (conv2d_f16.hip, sr_f16.hip: the reference's default super-resolution route,
training/networks_stylegan2.py:417-452 with use_fp16 and not force_fp32) against torch-CPU restatements of the reference's
float16 branch: float16 operands, float32 accumulation, one float16 rounding per operator.  Accumulation ORDER is the only
freedom, so outputs may differ by one float16 ulp where a float32 sum lands next to a rounding boundary: the tests allow 1 ulp
on a small fraction of the elements and nothing beyond."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from next3d_amd import _lib, layers
from oracle import ops as O

pytestmark = pytest.mark.gpu


def _g(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def _q(t):
    return t.half().float()


def _ulp16(ref):
    """float16 ulp at the magnitude of `ref` (float32 tensor of float16 values)."""
    a = ref.abs().clamp_min(2.0 ** -14)
    return torch.exp2(torch.floor(torch.log2(a)) - 10)


def _check_f16(name, got, ref, max_frac=0.02, pre=None, pre_gain=1.0):
    """`got`, `ref`: float32 tensors holding float16 values; differences of at most ONE float16 ulp, on at most max_frac of the elements.
    Outputs that are themselves a cancellation (|y| far below the summed magnitudes) carry the float32 summation-order noise of the
    large terms, many of THEIR ulps: an absolute floor of 2^-19 of the tensor's largest magnitude covers them.
    `pre`: the float32 value BEFORE the operator's first float16 rounding (the convolution / FIR result in front of bias_act): a
    one-ulp flip of that intermediate passes through x + b (where it may dwarf a cancelling sum) times `pre_gain`."""
    got, ref = got.detach().cpu().float(), ref.detach().cpu().float()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    d = (got - ref).abs()
    slack = 2.0 ** -19 * float(ref.abs().max())
    for t in (pre if isinstance(pre, (list, tuple)) else ([] if pre is None else [pre])):     # every intermediate float16 rounding in front of x + b
        slack = slack + pre_gain * _ulp16(_q(t.detach().cpu().float()))
    ulps = (d - slack).clamp_min(0) / _ulp16(ref)
    frac = float((d > 0).float().mean())
    print(f'{name}: max ulp {float(ulps.max()):.2f}, differing fraction {frac:.2e}, absmax {float(ref.abs().max()):.3g}')
    assert float(ulps.max()) <= 1.0 + 1e-6, (name, float(ulps.max()))
    assert frac <= max_frac, (name, frac)


class _Layer:
    """Stand-in for layers.PreparedConv (what modulate_weights_f16 / synthesis_layer_f16 read)."""

    def __init__(self, w, bias=None, noise_const=None, noise_strength=None):
        self.weight = w
        self.out_channels, self.in_channels, self.ksize = w.shape[0], w.shape[1], w.shape[2]
        self.bias, self.noise_const, self.noise_strength = bias, noise_const, noise_strength


def _ref_modulated_weights(w, s, demod):
    """modulated_conv2d's fused float16 branch up to `w.to(float16)` (training/networks_stylegan2.py:53-66, :88)."""
    o, i, kh, kw = w.shape
    n = s.shape[0]
    if demod:
        w = w * (1 / np.sqrt(i * kh * kw) / w.norm(float('inf'), dim=[1, 2, 3], keepdim=True))
        s = s / s.norm(float('inf'), dim=1, keepdim=True)
    wm = w.unsqueeze(0) * s.reshape(n, 1, -1, 1, 1)
    if demod:
        wm = wm * (wm.square().sum(dim=[2, 3, 4]) + 1e-8).rsqrt().reshape(n, -1, 1, 1, 1)
    return wm.half()                                                       # [N,O,I,k,k]


def _untile3(w16, n, o, i):
    """[N][9][I/16][2][O][8] -> [N,O,I,3,3]."""
    t = w16.reshape(n, 9, i // 16, 2, o, 8).permute(0, 4, 2, 3, 5, 1).reshape(n, o, i, 9)
    return t.reshape(n, o, i, 3, 3)


def _tile3(w16):
    """[N,O,I,3,3] float16 -> the flat [N][9][I/16][2][O][8] operand tiles of n3d_conv2d_f16."""
    n, o, i = w16.shape[:3]
    return w16.reshape(n, o, i // 16, 2, 8, 9).permute(0, 5, 2, 3, 1, 4).contiguous().reshape(-1)


@pytest.mark.parametrize('O,I,k,demod', [(64, 32, 3, True), (256, 256, 3, True), (128, 128, 3, True), (3, 256, 1, False), (3, 128, 1, False)])
def test_modulate_weights_f16(dev, O, I, k, demod):
    w, s = _g((O, I, k, k), 1), _g((3, I), 2) + 1.0
    ref = _ref_modulated_weights(w, s, demod)
    L = _Layer(w.to(dev))
    out = layers.modulate_weights_f16(L, s.to(dev), demodulate=demod).cpu()
    got = _untile3(out, 3, O, I) if k == 3 else out.reshape(3, O, I, 1, 1)
    # the reduction order of the demodulation sum differs -> a float32 ulp on d -> rarely a float16 ulp on a weight
    _check_f16(f'modulated weights O{O} I{I} k{k}', got.float(), ref.float(), max_frac=0.01)


def _ref_layer_epilogue(y32, bias, noise, gain, clamp, alpha=0.2, act='lrelu'):
    """conv output (float32 accumulation) -> float16 -> [+noise -> float16] -> bias_act in float32 -> float16 (bias_act.cu, half tensors)."""
    v = _q(y32)
    if noise is not None:
        v = _q(v + noise)
    t = v + _q(bias).reshape(1, -1, 1, 1)
    if act == 'lrelu':
        t = torch.where(t > 0, t, t * alpha)
    t = t * gain
    if clamp is not None:
        t = t.clamp(-clamp, clamp)
    return _q(t)


# the last two shapes (I >= 256, >= 256 workgroups) run on the <2,4> tile form (conv2d_f16.hip: F16Tile, 64 channels x 32 x 32 pixels), the second
# with ragged right / bottom tiles
@pytest.mark.parametrize('N,I,O,H,W,noise', [(2, 32, 64, 16, 32, False), (1, 128, 128, 40, 70, True), (2, 64, 192, 33, 64, True),
                                               (1, 256, 64, 64, 64, False), (4, 32, 256, 128, 256, True), (4, 256, 128, 128, 256, True),
                                               (4, 256, 64, 200, 330, False)])
def test_conv2d_f16_stride1(dev, N, I, O, H, W, noise):
    x, w, s = _q(_g((N, I, H, W), 3)), _g((O, I, 3, 3), 4), _g((N, I), 5) + 1.0
    bias, nz = _g((O,), 6, 0.1), _g((H, W), 7)
    nstr = torch.tensor(0.3)
    w16 = _ref_modulated_weights(w, s, True)                                # the REFERENCE's float16 weights
    pre = torch.cat([F.conv2d(x[n:n + 1], w16[n].float(), padding=1) for n in range(N)], 0)
    ref = _ref_layer_epilogue(pre, bias, (nz * nstr) if noise else None, float(np.sqrt(2)), 256.0)
    L = _Layer(w.to(dev), bias.to(dev), nz.to(dev) if noise else None, nstr.to(dev))
    wt = _tile3(w16).to(dev)          # identical weights on both sides (one float16 ulp on a weight is 2^-11 of a product: many ulps of a small output)
    assert float((layers.modulate_weights_f16(L, s.to(dev)).float() - wt.float()).abs().max()) <= float(_ulp16(w16.float()).max())
    epi = _lib.make_epilogue(noise=L.noise_const, noise_strength=L.noise_strength if noise else None, bias=L.bias, act='lrelu',
                             gain=float(np.sqrt(2)), clamp=256.0)
    y = layers.conv2d_f16(_lib.H8.from_nchw(x.to(dev)), wt, O, 0, epi)
    _check_f16(f'conv2d_f16 stride 1 {N}x{I}->{O} {H}x{W}', y.to_float(), ref, pre=[pre, _q(pre) + nz * nstr] if noise else pre, pre_gain=float(np.sqrt(2)))


@pytest.mark.parametrize('N,I,O,H,W', [(2, 32, 64, 16, 16), (1, 64, 128, 33, 40), (1, 128, 64, 64, 64), (2, 16, 64, 4, 7), (1, 32, 64, 72, 160)])
def test_conv2d_f16_transposed(dev, N, I, O, H, W):
    x, w, s = _q(_g((N, I, H, W), 8)), _g((O, I, 3, 3), 9), _g((N, I), 10) + 1.0
    w16 = _ref_modulated_weights(w, s, True)
    # conv2d_resample.py:116-127: the layer weight [O,I,k,k], transposed to [I,O,k,k], goes to conv_transpose2d UNFLIPPED
    ref = _q(torch.cat([F.conv_transpose2d(x[n:n + 1], w16[n].float().transpose(0, 1), stride=2) for n in range(N)], 0))
    wt = _tile3(w16).to(dev)
    y = layers.conv2d_f16(_lib.H8.from_nchw(x.to(dev)), wt, O, 2)
    assert y.shape == (N, O, 2 * H + 1, 2 * W + 1)
    _check_f16(f'conv2d_f16 transposed {N}x{I}->{O} {H}x{W}', y.to_float(), ref)


@pytest.mark.parametrize('N,C,H,W,noise', [(2, 64, 33, 33, False), (1, 16, 131, 70, True), (1, 8, 21, 257, True), (1, 8, 62, 123, False)])
def test_fir4_h8(dev, N, C, H, W, noise):
    fir = O.setup_filter((1, 3, 3, 1))
    z = _q(_g((N, C, H, W), 11, 4.0))
    bias, nz, nstr = _g((C,), 12, 0.2), _g((H - 1, W - 1), 13), torch.tensor(0.5)
    y32 = O.upfirdn2d(z, fir, padding=[1, 1, 1, 1], gain=4)                # float32 arithmetic on the float16 values, rounded by the epilogue helper
    ref = _ref_layer_epilogue(y32, bias, (nz * nstr) if noise else None, float(np.sqrt(2)), 256.0)
    b_d, nz_d, ns_d, f_d = bias.to(dev), nz.to(dev), nstr.to(dev), fir.to(dev)
    assert layers.fir_factor(f_d) is not None and torch.equal(layers.fir_factor(f_d).cpu(), torch.tensor([1., 3., 3., 1.]) / 8)
    for sep in ('1', '0'):                    # the separable form and the generic 16-tap kernel
        old = layers.uf.FIR_SEP
        layers.uf.FIR_SEP = sep == '1'
        try:
            epi = _lib.make_epilogue(noise=nz_d if noise else None, noise_strength=ns_d if noise else None, bias=b_d, act='lrelu', gain=float(np.sqrt(2)), clamp=256.0)
            yh = layers.fir4_h8(_lib.H8.from_nchw(z.to(dev)), f_d, epi)
        finally:
            layers.uf.FIR_SEP = old
        _check_f16(f'fir4_h8 sep={sep} {N}x{C} {H}x{W}', yh.to_float(), ref, pre=[y32, _q(y32) + nz * nstr] if noise else y32, pre_gain=float(np.sqrt(2)))


def test_modulate_weights_f16_multi_equals_single(dev):
    """The one-launch form writes exactly what the per-layer launches write."""
    Ls = [_Layer(_g((64, 32, 3, 3), 30).to(dev)), _Layer(_g((128, 128, 3, 3), 31).to(dev)), _Layer(_g((3, 128, 1, 1), 32).to(dev))]
    base = (_g((3, 32 + 128 + 128), 33) + 1.0).to(dev)
    views = [base[:, 0:32], base[:, 32:160], base[:, 160:288]]
    multi = layers.modulate_weights_f16_multi([(Ls[0], views[0], True), (Ls[1], views[1], True), (Ls[2], views[2], False)], base, 3)
    for L, v, d, m in zip(Ls, views, (True, True, False), multi):
        assert torch.equal(layers.modulate_weights_f16(L, v, demodulate=d), m)


@pytest.mark.parametrize('N,C,H,W,with_img', [(2, 256, 32, 48, True), (1, 128, 64, 64, True), (1, 64, 17, 19, False)])
def test_torgb_h8(dev, N, C, H, W, with_img):
    fir = O.setup_filter((1, 3, 3, 1))
    x, w, s, bias = _q(_g((N, C, H, W), 14)), _g((3, C, 1, 1), 15), (_g((N, C), 16) + 1.0) / np.sqrt(C), _g((3,), 17, 0.1)
    img_lo = _g((N, 3, H // 2, W // 2), 18) if with_img else None
    w16 = _ref_modulated_weights(w, s, False)
    y = _q(torch.cat([F.conv2d(x[n:n + 1], w16[n].float()) for n in range(N)], 0))
    y = _q((y + _q(bias).reshape(1, -1, 1, 1)).clamp(-256, 256))
    ref = (O.upsample2d(img_lo, fir) + y) if with_img else y
    L = _Layer(w.to(dev), bias.to(dev))
    img = layers.torgb_layer_f16(L, _lib.H8.from_nchw(x.to(dev)), s.to(dev), fir.to(dev), conv_clamp=256, img_lo=img_lo.to(dev) if with_img else None)
    err = float((img.cpu() - ref).abs().max())
    print(f'torgb_h8 {N}x{C} {H}x{W}: max-abs {err:.3e} (absmax {float(ref.abs().max()):.3g})')
    # toRGB itself: at most one float16 ulp; the float32 skip-image upsample adds float32 rounding only
    assert err <= float(_ulp16(y).max()) + 1e-5


@pytest.mark.parametrize('N,I,O,H,W,C,noise', [(4, 32, 128, 128, 128, 3, True), (2, 16, 64, 50, 70, 4, False), (1, 64, 192, 96, 160, 1, True)])
def test_fused_torgb_of_a_float16_last_layer(dev, N, I, O, H, W, C, noise):
    """n3d_conv2d_f16 with rgb_* + n3d_rgb_combine(round_f16): a float16 block's LAST conv1 evaluating its ToRGBLayer in the epilogue (the h8
    feature map is never written) against the two layers run separately (n3d_conv2d_f16 -> n3d_torgb_h8): the same float16 operands, float32
    sums in another order -> equal up to one float16 ulp of the colour where a sum lands on a rounding boundary."""
    fir = layers.uf.setup_filter([1, 3, 3, 1]).to(dev)
    x, w, s = _q(_g((N, I, H, W), 30)), _g((O, I, 3, 3), 31), _g((N, I), 32) + 1.0
    bias, nz, nstr = _g((O,), 33, 0.1), _g((H, W), 34), torch.tensor(0.3)
    wr, sr, br = _g((C, O, 1, 1), 35), (_g((N, O), 36) + 1.0) / np.sqrt(O), _g((C,), 37, 0.1)
    even = H % 2 == 0 and W % 2 == 0
    img_lo = _g((N, C, H // 2, W // 2), 38).to(dev) if even else None
    L1 = _Layer(w.to(dev), bias.to(dev), nz.to(dev) if noise else None, nstr.to(dev))
    LT = _Layer(wr.to(dev), br.to(dev))
    w16 = layers.modulate_weights_f16(L1, s.to(dev))
    wt16 = layers.modulate_weights_f16(LT, sr.to(dev), demodulate=False)
    epi = _lib.make_epilogue(noise=L1.noise_const, noise_strength=L1.noise_strength if noise else None, bias=L1.bias, act='lrelu', gain=float(np.sqrt(2)), clamp=256.0)
    xh = _lib.H8.from_nchw(x.to(dev))
    feat = layers.conv2d_f16(xh, w16, O, 0, epi)
    sep = layers.torgb_layer_f16(LT, feat, None, fir, conv_clamp=256, img_lo=img_lo, w16=wt16)
    part = layers.conv2d_f16(xh, w16, O, 0, epi, rgb=(wt16, C))
    assert tuple(part.shape) == (N, O // 64, C, H, W)
    got = layers.torgb_combine_f16(LT, part, fir, conv_clamp=256, img_lo=img_lo)
    d = (got - sep).abs()
    y = sep - (layers.uf.upsample2d(img_lo, fir) if even else 0)
    print(f'fused float16 toRGB {N}x{I}->{O}->{C} {H}x{W}: max-abs {float(d.max()):.3e}, differing {float((d > 0).float().mean()):.2e} of the values')
    # one ulp on the rounded sum can become two behind the second rounding (bias_act), and an ulp doubles across a binade boundary
    # (the first rounding happens on the sum BEFORE the bias is added: its ulp is that of |y - bias|, bounded here by |y| + max |bias|)
    assert bool((d <= 3 * _ulp16(y.abs().cpu() + float(br.abs().max())).to(dev) + 1e-5).all()) and float((d > 1e-6).float().mean()) < 0.02


def test_cast_h8_round_trip(dev):
    x = _g((2, 24, 9, 13), 19, 100.0)
    h = _lib.H8.from_nchw(x.to(dev))
    assert torch.equal(h.to_float().cpu(), _q(x))
    y = torch.empty(2, 24, 9, 13, device=dev)
    _lib.check(_lib.lib().n3d_cast_h8(_lib.ptr(h.data), _lib.ptr(y), 2, 24, 9 * 13, 0, 0, _lib.stream()))
    assert torch.equal(y.cpu(), _q(x))


@pytest.mark.parametrize('up,rgb', [(1, False), (2, False), (1, True)])
def test_float16_layer_with_random_noise_runs_sample_by_sample(dev, up, rgb):
    """noise_mode='random' (the reference's default, training/networks_stylegan2.py:318-319: randn([N,1,res,res]) * noise_strength added inside
    modulated_conv2d) on a float16 layer: the kernels take one noise image per launch, so layers.synthesis_layer_f16 draws [N,res,res] with torch.randn
    from the device generator and runs the noisy part of the layer sample by sample.  Checked against the 'const' path of the SAME layer called once per
    sample with that sample's draw as its noise image (same device seed -> same draws): bit-identical; and different from the noise-free layer."""
    N, I, O, H, W, C = 3, 32, 128, 32, 32, 3
    fir = layers.uf.setup_filter([1, 3, 3, 1]).to(dev)
    res = (2 * H, 2 * W) if up == 2 else (H, W)
    x, w, s = _q(_g((N, I, H, W), 70)), _g((O, I, 3, 3), 71), _g((N, I), 72) + 1.0
    L1 = _Layer(w.to(dev), _g((O,), 73, 0.1).to(dev), torch.zeros(res, device=dev), torch.tensor(0.5, device=dev))
    w16 = layers.modulate_weights_f16(L1, s.to(dev))
    xh = _lib.H8.from_nchw(x.to(dev))
    kw = dict(up=up, conv_clamp=256, w16=w16)
    if rgb:
        LT = _Layer(_g((C, O, 1, 1), 74).to(dev), _g((C,), 75, 0.1).to(dev))
        wt16 = layers.modulate_weights_f16(LT, ((_g((N, O), 76) + 1.0) / np.sqrt(O)).to(dev), demodulate=False)
        kw['rgb'] = (wt16, C)
    val = (lambda r: r) if rgb else (lambda r: r.data)
    torch.manual_seed(1234)
    got = val(layers.synthesis_layer_f16(L1, xh, None, fir, noise_mode='random', **kw))
    torch.manual_seed(1234)
    draws = torch.randn([N, *res], dtype=torch.float32, device=dev)
    per = w16.numel() // N
    for i in range(N):
        Li = _Layer(L1.weight, L1.bias, draws[i].contiguous(), L1.noise_strength)
        kwi = dict(kw, w16=w16[i * per:(i + 1) * per])
        if rgb:
            kwi['rgb'] = (wt16.reshape(N, -1)[i], C)
        want = val(layers.synthesis_layer_f16(Li, xh.sample(i), None, fir, noise_mode='const', **kwi))
        assert torch.equal(got[i:i + 1], want), (up, rgb, i)
    quiet = val(layers.synthesis_layer_f16(L1, xh, None, fir, noise_mode='none', **kw))
    assert float((got.float() - quiet.float()).abs().max()) > 1e-2                   # the noise is really added
