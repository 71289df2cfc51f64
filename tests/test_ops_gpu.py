"""-m gpu parity tests for the operator layer: libn3d.so (through the C ABI) vs the CPU oracle on seeded inputs.
Tolerances are written per test; fp32 end to end, so they sit at the fp32-roundoff level (north_star: 1e-3 on RGB)."""
import numpy as np
import pytest
import torch

from oracle import ops as O

pytestmark = pytest.mark.gpu


def _gen(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def _close(a, b, atol, rtol=1e-5):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    assert bool((err <= tol).all()), f'max err {float(err.max()):.3e} (ref absmax {float(b.abs().max()):.3e})'


@pytest.mark.parametrize('act', ['linear', 'relu', 'lrelu', 'tanh', 'sigmoid', 'elu', 'selu', 'softplus', 'swish'])
@pytest.mark.parametrize('shape,dim', [((3, 5, 7, 9), 1), ((4, 512), 1), ((2, 6, 4, 4), 0), ((1, 3, 17, 5), 3)])
def test_bias_act(dev, act, shape, dim):
    from next3d_amd.torch_utils.ops import bias_act
    x, b = _gen(shape, 1, 3.0), _gen((shape[dim],), 2)
    for kw in (dict(), dict(gain=0.7, clamp=1.5, alpha=0.3)):
        y = bias_act.bias_act(x.to(dev), b.to(dev), dim=dim, act=act, **kw)
        _close(y, O.bias_act(x, b, dim=dim, act=act, **kw), atol=2e-6)
    y = bias_act.bias_act(x.to(dev), None, act=act)
    _close(y, O.bias_act(x, None, act=act), atol=2e-6)


def test_bias_act_f16_and_errors(dev):
    from next3d_amd.torch_utils.ops import bias_act
    x, b = _gen((2, 8, 5, 5), 3).half(), _gen((8,), 4).half()
    y = bias_act.bias_act(x.to(dev), b.to(dev), act='lrelu', clamp=256)
    _close(y.float(), O.bias_act(x.float(), b.float(), act='lrelu', clamp=256), atol=2e-3, rtol=2e-3)
    with pytest.raises(RuntimeError):
        bias_act.bias_act(x.to(dev), _gen((7,), 1).half().to(dev))
    with pytest.raises(RuntimeError):
        bias_act.bias_act(x, b)                       # CPU tensors: no fallback
    assert bias_act.bias_act(torch.empty(0, 4, device=dev), None).numel() == 0


def test_bias_act_propagates_nan_without_clamp(dev):
    """without a clamp a NaN activation stays NaN on BOTH kernels (the 2-D row form serves contiguous NCHW tensors; fminf / fmaxf
    against +-inf would turn it into -inf), as the reference's bias_act does (bias_act.cu: the clamp is applied only when clamp >= 0)."""
    from next3d_amd.torch_utils.ops import bias_act
    for shape in ((2, 8, 16, 16), (2, 8, 5, 5)):                      # row kernel (inner extent 256) / generic kernel (25)
        x = torch.zeros(shape)
        x[1, 3, 2, 1] = float('nan')
        b = _gen((shape[1],), 5)
        for act in ('linear', 'lrelu'):
            y = bias_act.bias_act(x.to(dev), b.to(dev), act=act).cpu()
            assert torch.isnan(y[1, 3, 2, 1]) and int(torch.isnan(y).sum()) == 1, (shape, act)


def test_operator_layer_matches_reference_ref_ops(dev):
    """The B1 operator layer on libn3d.so against the REFERENCE's own _bias_act_ref / _upfirdn2d_ref / _filtered_lrelu_ref
    outputs (tests/golden/ref_ops.npz): every activation, separable / asymmetric filters, per-axis up x down, negative
    padding, flips.  fp32 end to end: tolerance at fp32 round-off (transcendentals: device libm vs host libm)."""
    import _ref_ops
    from next3d_amd.torch_utils.ops import bias_act, filtered_lrelu, upfirdn2d
    mod = {'bias_act': bias_act.bias_act, 'upfirdn2d': upfirdn2d.upfirdn2d, 'filtered_lrelu': filtered_lrelu.filtered_lrelu}
    for i, op, kw, t, y_ref in _ref_ops.load():
        y = _ref_ops.run(mod, op, kw, t, upfirdn2d.setup_filter, to=lambda v: v.to(dev))
        _close(y, y_ref, atol=1e-5 if op != 'bias_act' else 4e-6)


UF_CASES = [
    # (shape, up, down, padding, gain)  — the three hot shapes of SURVEY §8(a7) + odd ones
    ((2, 5, 17, 17), 1, 1, [1, 1, 1, 1], 4.0),
    ((2, 3, 16, 16), 2, 1, [2, 1, 2, 1], 4.0),
    ((1, 4, 32, 32), 1, 2, [1, 1, 1, 1], 1.0),
    ((1, 4, 32, 32), 1, 1, [2, 2, 2, 2], 1.0),
    ((2, 2, 13, 70), 2, 3, [3, 0, -1, 2], 0.5),
    ((1, 1, 4, 4), 1, 1, [2, 1, 1, 2], 1.0),
    ((1, 2, 130, 67), 1, 1, [1, 1, 1, 1], 4.0),
]


@pytest.mark.parametrize('shape,up,down,padding,gain', UF_CASES)
@pytest.mark.parametrize('flip', [False, True])
def test_upfirdn2d(dev, shape, up, down, padding, gain, flip):
    from next3d_amd.torch_utils.ops import upfirdn2d
    x = _gen(shape, 5)
    f = O.setup_filter((1, 3, 3, 1)) + 0.01 * torch.arange(16.).reshape(4, 4)     # asymmetric taps
    y = upfirdn2d.upfirdn2d(x.to(dev), f.to(dev), up=up, down=down, padding=padding, flip_filter=flip, gain=gain)
    _close(y, O.upfirdn2d(x, f, up=up, down=down, padding=padding, flip_filter=flip, gain=gain), atol=1e-5)


def test_upfirdn2d_helpers_and_separable(dev):
    from next3d_amd.torch_utils.ops import upfirdn2d
    x = _gen((2, 3, 24, 20), 6)
    f = O.setup_filter((1, 3, 3, 1))
    _close(upfirdn2d.upsample2d(x.to(dev), f.to(dev)), O.upsample2d(x, f), atol=1e-5)
    _close(upfirdn2d.downsample2d(x.to(dev), f.to(dev)), O.downsample2d(x, f), atol=1e-5)
    f1 = upfirdn2d.setup_filter([1, 2, 4, 6, 9, 6, 4, 2, 1][:8])                      # 8 taps -> separable 1-D
    assert f1.ndim == 1
    _close(upfirdn2d.upfirdn2d(x.to(dev), f1.to(dev), up=2, padding=[4, 3, 4, 3], gain=4),
           O.upfirdn2d(x, f1, up=2, padding=[4, 3, 4, 3], gain=4), atol=1e-5)
    _close(upfirdn2d.upfirdn2d(x.to(dev), None), x, atol=0)
    with pytest.raises(RuntimeError):
        upfirdn2d.upfirdn2d(_gen((1, 1, 2, 2), 1).to(dev), f.to(dev))               # output would be empty


def test_fc(dev):
    from next3d_amd import layers
    for (n, i, o) in [(4, 512, 512), (3, 1024, 512), (2, 25, 512), (9, 512, 96), (1, 33, 7)]:
        x, w, b = _gen((n, i), 7), _gen((o, i), 8), _gen((o,), 9)
        y = layers.fc(x.to(dev), w.to(dev), b.to(dev), wgain=1 / np.sqrt(i))
        _close(y, O.fully_connected(x, w, b), atol=2e-5)
        y = layers.fc(x.to(dev), (w / 0.01).to(dev), b.to(dev), wgain=0.01 / np.sqrt(i), bgain=0.01, act='lrelu')
        _close(y, O.fully_connected(x, w / 0.01, b, activation='lrelu', lr_multiplier=0.01), atol=2e-5)
    s, w2 = _gen((4, 256), 10), _gen((128, 256), 11).square()
    d = layers.fc(s.to(dev), w2.to(dev), pre_square=True, post_rsqrt=True)
    _close(d, (s.square() @ w2.t() + 1e-8).rsqrt(), atol=1e-6, rtol=1e-5)


CONV_CASES = [
    # (N, I, O, H, W, ksize, mode)
    (2, 16, 128, 32, 32, 3, 0),
    (1, 24, 160, 20, 37, 3, 0),      # ragged: channel tails, partial tiles
    (2, 512, 512, 4, 4, 3, 0),       # deep-K small grid (split-K)
    (1, 64, 128, 16, 16, 3, 0),
    (2, 32, 256, 33, 33, 3, 1),      # stride 2 after the FIR pre-filter (odd input)
    (1, 16, 128, 17, 9, 3, 1),
    (2, 32, 128, 16, 16, 3, 2),      # transposed
    (1, 512, 512, 4, 4, 3, 2),
    (1, 8, 128, 5, 40, 3, 2),
    (2, 128, 32, 64, 64, 1, 0),      # toRGB-like
    (1, 128, 3, 32, 48, 1, 0),
    (2, 32, 512, 8, 8, 1, 0),        # fromrgb-like
    (1, 40, 96, 16, 16, 1, 0),
]


def _conv_ref(x, w, mode):
    import torch.nn.functional as F
    if mode == 0:
        return F.conv2d(x, w, padding=w.shape[2] // 2)
    if mode == 1:
        return F.conv2d(x, w, stride=2)
    return F.conv_transpose2d(x, w.transpose(0, 1), stride=2)


@pytest.mark.parametrize('N,I,OC,H,W,k,mode', CONV_CASES)
def test_conv2d_plain(dev, N, I, OC, H, W, k, mode):
    from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
    x, w = _gen((N, I, H, W), 12), _gen((OC, I, k, k), 13) / np.sqrt(I * k * k)
    ref = _conv_ref(x, w, mode)
    wt = cg.prep_weight(w.to(dev))
    for ksplit in (1, None, 3):
        y = cg.conv_launch(x.to(dev), wt, k, mode, OC, ksplit=ksplit)
        _close(y, ref, atol=2e-5, rtol=1e-4)


def test_conv2d_epilogue_and_style(dev):
    from next3d_amd import _lib
    from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
    N, I, OC, H, W = 2, 32, 128, 24, 40
    x, w = _gen((N, I, H, W), 14), _gen((OC, I, 3, 3), 15) / np.sqrt(I * 9)
    s, d, b = _gen((N, I), 16), _gen((N, OC), 17).abs() + 0.5, _gen((OC,), 18)
    noise, ns, res = _gen((H, W), 19), torch.tensor(0.3), _gen((N, OC, H, W), 20)
    import torch.nn.functional as F
    ref = F.conv2d(x * s[:, :, None, None], w, padding=1) * d[:, :, None, None] * 0.9 + noise * ns
    ref = O.bias_act(ref, b, act='lrelu', gain=1.3, clamp=2.0) + res
    t = lambda a: a.to(dev)
    for ksplit in (1, 4):
        epi = _lib.make_epilogue(row_scale=t(d), noise=t(noise), noise_strength=t(ns), bias=t(b), residual=t(res),
                                 const_scale=0.9, act='lrelu', gain=1.3, clamp=2.0)
        y = cg.conv_launch(t(x), cg.prep_weight(t(w)), 3, 0, OC, style=t(s), epilogue=epi, ksplit=ksplit)
        _close(y, ref, atol=3e-5, rtol=1e-4)


def test_conv2d_resample_branches(dev):
    """The branch table of conv2d_resample (reference conv2d_resample.py:96-136) incl. the grouped (fused-modconv) form."""
    from next3d_amd.torch_utils.ops import conv2d_resample as cr
    f = O.setup_filter((1, 3, 3, 1))
    x = _gen((2, 16, 20, 20), 21)
    w3, w1 = _gen((32, 16, 3, 3), 22) / 12, _gen((32, 16, 1, 1), 23) / 4
    t = lambda a: a.to(dev)
    for kw in (dict(w=w3, padding=1), dict(w=w3, up=2, padding=1, flip_weight=False), dict(w=w3, down=2, padding=1),
               dict(w=w1), dict(w=w1, up=2), dict(w=w1, down=2), dict(w=w3, padding=1, flip_weight=False)):
        w = kw.pop('w')
        _close(cr.conv2d_resample(t(x), t(w), f=t(f), **kw), O.conv2d_resample(x, w, f=f, **kw), atol=3e-5, rtol=1e-4)
    xg = x.reshape(1, 32, 20, 20)
    wg = _gen((2 * 24, 16, 3, 3), 24) / 12
    for kw in (dict(padding=1), dict(up=2, padding=1, flip_weight=False)):
        _close(cr.conv2d_resample(t(xg), t(wg), f=t(f), groups=2, **kw), O.conv2d_resample(xg, wg, f=f, groups=2, **kw),
               atol=3e-5, rtol=1e-4)


@pytest.mark.parametrize('up,res,ic,oc', [(1, 16, 64, 128), (2, 16, 64, 128), (1, 4, 512, 512), (2, 64, 32, 256)])
def test_synthesis_layer_and_torgb(dev, up, res, ic, oc):
    from next3d_amd import layers
    from oracle import networks as ON
    P = {'L.weight': _gen((oc, ic, 3, 3), 30), 'L.bias': _gen((oc,), 31) * 0.1, 'L.affine.weight': _gen((ic, 512), 32),
         'L.affine.bias': 1 + 0.1 * _gen((ic,), 33), 'L.noise_const': _gen((res, res), 34), 'L.noise_strength': torch.tensor(0.2),
         'T.weight': _gen((32, oc, 1, 1), 35), 'T.bias': _gen((32,), 36) * 0.1, 'T.affine.weight': _gen((oc, 512), 37),
         'T.affine.bias': 1 + 0.1 * _gen((oc,), 38)}
    Pd = {k: v.to(dev) for k, v in P.items()}
    fir = O.setup_filter((1, 3, 3, 1))
    N = 2
    x, w = _gen((N, ic, res // up, res // up), 39), _gen((N, 512), 40)
    for clamp in (None, 1.0):
        ref = ON.synthesis_layer(P, 'L', x, w, up=up, conv_clamp=clamp)
        y = layers.synthesis_layer(layers.PreparedConv(Pd, 'L', True), x.to(dev), w.to(dev), fir.to(dev), up=up, conv_clamp=clamp)
        _close(y, ref, atol=1e-4, rtol=1e-4)
    img = _gen((N, 32, res, res), 41)
    ref = ON.torgb_layer(P, 'T', ref, w) + img
    yt = layers.torgb_layer(layers.PreparedConv(Pd, 'T', True, demodulate=False), y, w.to(dev), residual=img.to(dev))
    _close(yt, ref, atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize('N,I,OC,H,W', [(2, 32, 128, 32, 32), (1, 64, 100, 19, 45), (2, 512, 512, 32, 32), (1, 128, 64, 64, 96), (1, 16, 3, 8, 32),
                                       (4, 512, 512, 16, 16), (2, 128, 70, 8, 8), (4, 64, 64, 4, 4), (1, 32, 40, 37, 20), (1, 16, 8, 70, 5),
                                       (1, 32, 64, 9, 31)])
def test_conv2d_bf16x3(dev, N, I, OC, H, W):
    """Split-bf16 conv vs the fp32 oracle: operand truncation at 2^-16 relative -> tolerance 1e-4 of the output scale
    (A = identity-like checks are implicit: weights and inputs are asymmetric random)."""
    import torch.nn.functional as F
    from next3d_amd import _lib
    from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
    x, w = _gen((N, I, H, W), 50), _gen((OC, I, 3, 3), 51) / np.sqrt(I * 9)
    s, d, b = _gen((N, I), 52), _gen((N, OC), 53).abs() + 0.5, _gen((OC,), 54)
    ref_plain = F.conv2d(x, w, padding=1)
    ref_full = O.bias_act(F.conv2d(x * s[:, :, None, None], w, padding=1) * d[:, :, None, None], b, act='lrelu')
    t = lambda a: a.to(dev)
    wt16 = cg.prep_weight_bf16x3(t(w))
    for ksplit in (1, None, 2):
        y = cg.conv_launch(t(x), wt16, 3, 0, OC, ksplit=ksplit, bf16x3=True)
        err = float((y.cpu() - ref_plain).abs().max())
        assert err <= 1e-4 * max(1.0, float(ref_plain.abs().max())), (ksplit, err)
    ts, td, tb = t(s), t(d), t(b)
    y = cg.conv_launch(t(x), wt16, 3, 0, OC, style=ts, epilogue=_lib.make_epilogue(row_scale=td, bias=tb, act='lrelu'), bf16x3=True)
    err = float((y.cpu() - ref_full).abs().max())
    assert err <= 1e-4 * max(1.0, float(ref_full.abs().max())), err


@pytest.mark.parametrize('N,I,OC,H,W', [(4, 512, 512, 4, 4), (4, 512, 512, 8, 8), (4, 512, 512, 16, 16), (4, 1024, 512, 8, 8), (1, 512, 512, 32, 32),
                                       (3, 128, 64, 4, 4), (2, 256, 96, 16, 16), (1, 128, 32, 8, 8), (2, 128, 64, 8, 16), (8, 128, 256, 16, 16), (5, 128, 32, 4, 8)])
def test_conv2d_bf16x3_few_pixel_kernel(dev, N, I, OC, H, W):
    """conv2d_sk_bf16x3_kernel (few-pixel layers: the whole K inside one workgroup, 8 waves splitting the input channels, patch staged per
    wave, LDS reduction, epilogue in the same launch) against the float32 oracle: 32- and 64-pixel tiles, two samples per tile (4 x 4
    images, odd batch), channel-tile counts that do / do not divide by the 8 XCDs, non-square images, a batch-strided input view and
    output view, the full layer epilogue with a residual, run-to-run bitwise equality."""
    import torch.nn.functional as F
    from next3d_amd import _lib
    from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
    assert _lib.lib().n3d_conv2d_sk_eligible(N, I, OC, H, W) == 1
    x, w = _gen((N, I, H, W), 160), _gen((OC, I, 3, 3), 161) / np.sqrt(I * 9)
    s, d, b = _gen((N, I), 162), _gen((N, OC), 163).abs() + 0.5, _gen((OC,), 164)
    noise, ns, res = _gen((H, W), 165), torch.tensor(0.4), _gen((N, OC, H, W), 166)
    ref_plain = F.conv2d(x, w, padding=1)
    ref_full = O.bias_act(F.conv2d(x * s[:, :, None, None], w, padding=1) * d[:, :, None, None] * 0.8 + noise * ns, b, act='lrelu', gain=1.2, clamp=1.5) + res
    t = lambda a: a.to(dev)
    wt16 = cg.prep_weight_bf16x3(t(w))
    y = cg.conv_launch(t(x), wt16, 3, 0, OC, bf16x3=True)
    err = float((y.cpu() - ref_plain).abs().max())
    assert err <= 1e-4 * max(1.0, float(ref_plain.abs().max())), err
    epi = _lib.make_epilogue(row_scale=t(d), noise=t(noise), noise_strength=t(ns), bias=t(b), const_scale=0.8, act='lrelu', gain=1.2, clamp=1.5, residual=t(res))
    xv = torch.zeros(N, I + 16, H, W, device=dev)[:, 16:]                 # batch stride > I * H * W (the U-Net's concatenation buffers)
    xv.copy_(t(x))
    out = torch.zeros(N, OC + 8, H, W, device=dev)[:, :OC]
    y = cg.conv_launch(xv, wt16, 3, 0, OC, style=t(s), epilogue=epi, bf16x3=True, out=out)
    err = float((y.cpu() - ref_full).abs().max())
    assert err <= 1e-4 * max(1.0, float(ref_full.abs().max())), err
    assert torch.equal(y, cg.conv_launch(t(x), wt16, 3, 0, OC, style=t(s), epilogue=epi, bf16x3=True))       # other views, same bits; run to run
    # batch stride 0: one image for the whole batch (SynthesisBlock's learned constant, `const.unsqueeze(0).expand(n, ...)`, networks_stylegan2.py:463-465)
    xe = t(x)[:1].expand(N, -1, -1, -1)
    ye = cg.conv_launch(xe, wt16, 3, 0, OC, style=t(s), bf16x3=True)
    assert torch.equal(ye, cg.conv_launch(xe.contiguous(), wt16, 3, 0, OC, style=t(s), bf16x3=True))


@pytest.mark.parametrize('N,I,OC,H,W', [(4, 512, 512, 17, 17), (4, 512, 512, 9, 9), (1, 512, 512, 17, 17), (3, 128, 96, 9, 9), (2, 256, 64, 33, 33), (1, 512, 512, 9, 9)])
def test_conv2d_sk_stride2_few_pixel_layers(dev, N, I, OC, H, W):
    """conv2d_sk_bf16x3_kernel<1, 2> (round 5): the few-pixel STRIDE-2 layers — the mouth encoder's 17 x 17 -> 8 x 8 and 9 x 9 -> 4 x 4 convolutions behind
    their FIR (Conv2dLayer down = 2, conv2d_resample.py:108-111), which ran on the fp32-MFMA kernel with split-K 16 + a reduce launch — in one launch on the
    split-bf16 matrix-core path, against float32 ATen: plain, with the full epilogue + residual into a strided output view, odd batch / two samples per
    tile, channel counts that do not divide by the XCDs, run-to-run bitwise equality; the library's own eligibility answer decides the route."""
    import torch.nn.functional as F
    from next3d_amd import _lib
    from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
    assert _lib.lib().n3d_conv2d_sk_s2_eligible(N, I, OC, H, W) == 1 and cg.sk_s2_eligible(N, I, OC, H, W)
    assert _lib.lib().n3d_conv2d_sk_s2_eligible(N, I, OC, H + 1, W) == 0 and _lib.lib().n3d_conv2d_sk_s2_eligible(N, I + 16, OC, H, W) == 0
    OH, OW = (H - 3) // 2 + 1, (W - 3) // 2 + 1
    x, w = _gen((N, I, H, W), 170), _gen((OC, I, 3, 3), 171) / np.sqrt(I * 9)
    b, res = _gen((OC,), 172), _gen((N, OC, OH, OW), 173)
    ref_plain = F.conv2d(x, w, stride=2)
    ref_full = O.bias_act(ref_plain * 0.7, b, act='lrelu', gain=1.3, clamp=2.0) + res
    t = lambda a: a.to(dev)
    wt16 = cg.prep_weight_bf16x3(t(w))
    y = cg.conv_launch(t(x), wt16, 3, 1, OC, bf16x3=True)
    assert tuple(y.shape) == (N, OC, OH, OW)
    err = float((y.cpu() - ref_plain).abs().max())
    assert err <= 1e-4 * max(1.0, float(ref_plain.abs().max())), err
    epi = _lib.make_epilogue(bias=t(b), const_scale=0.7, act='lrelu', gain=1.3, clamp=2.0, residual=t(res))
    out = torch.zeros(N, OC + 8, OH, OW, device=dev)[:, 8:]              # a channel-slice view (the U-Net's concatenation buffers)
    y2 = cg.conv_launch(t(x), wt16, 3, 1, OC, epilogue=epi, bf16x3=True, out=out)
    err = float((y2.cpu() - ref_full).abs().max())
    assert err <= 1e-4 * max(1.0, float(ref_full.abs().max())), err
    assert torch.equal(y2, cg.conv_launch(t(x), wt16, 3, 1, OC, epilogue=epi, bf16x3=True))
    # against the fp32-MFMA route these layers took before (split-K + reduce launch): same result to the split-bf16 accuracy
    y32 = cg.conv_launch(t(x), cg.prep_weight(t(w)), 3, 1, OC, epilogue=epi)
    assert float((y2 - y32).abs().max()) <= 1e-4 * max(1.0, float(ref_full.abs().max()))


@pytest.mark.parametrize('N,I,OC,H,W', [(2, 64, 128, 256, 256), (3, 48, 256, 144, 160), (3, 32, 128, 250, 200), (1, 128, 128, 512, 512),
                                       (4, 512, 256, 128, 128)])
def test_conv2d_bf16x3_persistent(dev, N, I, OC, H, W):
    """Layers with >= 2 tiles per CU take the persistent kernel (conv2d_p_bf16x3.hip: K loop pipelined across tiles, epilogue
    under the other wave role's MFMA block).  Cases: even tile counts; an odd number of 16-channel chunks (the LDS buffer parity
    flips from tile to tile) with a tile count that does not divide by 8 XCDs x 32 workgroups; ragged image edges; the 512²
    super-resolution shape; the deepest eligible layer (512 input channels: 32 chunks, two style values per filling thread).  Checked with the full layer epilogue (style, demodulation, noise, bias, leaky ReLU, clamp), with the
    bare linear epilogue, and against the plain kernel (forced by a split-K of 2)."""
    import torch.nn.functional as F
    from next3d_amd import _lib
    from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
    assert _lib.lib().n3d_conv2d_bf16x3_blocks(N, OC, H, W, 0) >= 512
    x, w = _gen((N, I, H, W), 80), _gen((OC, I, 3, 3), 81) / np.sqrt(I * 9)
    s, d, b = _gen((N, I), 82), _gen((N, OC), 83).abs() + 0.5, _gen((OC,), 84)
    noise, ns = _gen((H, W), 85), torch.tensor(0.4)
    ref_plain = F.conv2d(x, w, padding=1)
    ref_full = O.bias_act(F.conv2d(x * s[:, :, None, None], w, padding=1) * d[:, :, None, None] * 0.8 + noise * ns, b, act='lrelu', gain=1.2, clamp=1.5)
    t = lambda a: a.to(dev)
    wt16 = cg.prep_weight_bf16x3(t(w))
    y = cg.conv_launch(t(x), wt16, 3, 0, OC, ksplit=1, bf16x3=True)
    err = float((y.cpu() - ref_plain).abs().max())
    assert err <= 1e-4 * max(1.0, float(ref_plain.abs().max())), err
    y2 = cg.conv_launch(t(x), wt16, 3, 0, OC, ksplit=2, bf16x3=True)                 # split-K -> the plain kernel + reduction pass
    assert float((y - y2).abs().max()) <= 2e-5 * max(1.0, float(ref_plain.abs().max()))
    epi = _lib.make_epilogue(row_scale=t(d), noise=t(noise), noise_strength=t(ns), bias=t(b), const_scale=0.8, act='lrelu', gain=1.2, clamp=1.5)
    y = cg.conv_launch(t(x), wt16, 3, 0, OC, style=t(s), epilogue=epi, ksplit=1, bf16x3=True)
    err = float((y.cpu() - ref_full).abs().max())
    assert err <= 1e-4 * max(1.0, float(ref_full.abs().max())), err
    assert torch.equal(y, cg.conv_launch(t(x), wt16, 3, 0, OC, style=t(s), epilogue=epi, ksplit=1, bf16x3=True))       # run-to-run bitwise


@pytest.mark.parametrize('N,I,OC,H,W', [(2, 32, 128, 32, 32), (1, 64, 100, 7, 45), (1, 512, 256, 64, 64), (1, 16, 64, 4, 33),
                                       (4, 512, 512, 16, 16), (2, 128, 70, 8, 8), (1, 32, 64, 9, 13), (4, 64, 256, 66, 34),
                                       (1, 16, 64, 40, 3), (1, 16, 8, 1, 1)])
def test_conv2d_up_bf16x3(dev, N, I, OC, H, W):
    """Split-bf16 transposed stride-2 conv vs F.conv_transpose2d."""
    import torch.nn.functional as F
    from next3d_amd import _lib
    from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
    x, w = _gen((N, I, H, W), 60), _gen((OC, I, 3, 3), 61) / np.sqrt(I * 9)
    s, d = _gen((N, I), 62), _gen((N, OC), 63).abs() + 0.5
    ref = F.conv_transpose2d(x * s[:, :, None, None], w.transpose(0, 1), stride=2) * d[:, :, None, None]
    t = lambda a: a.to(dev)
    wt16, ts, td = cg.prep_weight_bf16x3(t(w)), t(s), t(d)
    for ksplit in (1, None, 2):
        y = cg.conv_launch(t(x), wt16, 3, 2, OC, style=ts, epilogue=_lib.make_epilogue(row_scale=td), ksplit=ksplit, bf16x3=True)
        err = float((y.cpu() - ref).abs().max())
        assert err <= 1e-4 * max(1.0, float(ref.abs().max())), (ksplit, err)


@pytest.mark.parametrize('shape', [(2, 3, 17, 17), (1, 2, 129, 257), (1, 2, 65, 65), (2, 1, 33, 129)])
@pytest.mark.parametrize('flip', [False, True])
def test_upfirdn2d_pitched_rows(dev, shape, flip):
    """The FIR after a transposed conv reads an odd-width input through rows padded to 16 bytes (aligned float4 path):
    same result as the dense input, with and without the fused epilogue, and NaNs in the pitch padding must not leak."""
    from next3d_amd import _lib
    from next3d_amd.torch_utils.ops import upfirdn2d
    n, c, h, w = shape
    x = _gen(shape, 70)
    f = O.setup_filter((1, 3, 3, 1)) + 0.01 * torch.arange(16.).reshape(4, 4)
    buf = torch.full((n, c, h, (w + 3) // 4 * 4), float('nan'), device=dev)
    xv = buf[..., :w]
    xv.copy_(x.to(dev))
    ref = O.upfirdn2d(x, f, padding=[1, 1, 1, 1], flip_filter=flip, gain=4)
    y = upfirdn2d.upfirdn2d(xv, f.to(dev), padding=[1, 1, 1, 1], flip_filter=flip, gain=4)
    assert y.shape == ref.shape and y.is_contiguous()
    _close(y, ref, atol=1e-5)
    noise, b, res = _gen(ref.shape[2:], 71), _gen((c,), 72), _gen(ref.shape, 73)
    ns = torch.tensor([0.3])
    epi = _lib.make_epilogue(noise=noise.to(dev), noise_strength=ns.to(dev), bias=b.to(dev), act='lrelu', gain=np.sqrt(2), clamp=1.5,
                             residual=res.to(dev))
    y = upfirdn2d.upfirdn2d(xv, f.to(dev), padding=[1, 1, 1, 1], flip_filter=flip, gain=4, _epilogue=epi)
    ref2 = O.bias_act(ref + noise * ns, b, act='lrelu', gain=np.sqrt(2), clamp=1.5) + res
    _close(y, ref2, atol=1e-5)


@pytest.mark.parametrize('bf16x3', [False, True])
@pytest.mark.parametrize('mode,N,I,OC,H,W', [(2, 2, 32, 70, 16, 32), (2, 1, 64, 64, 33, 45), (0, 1, 32, 40, 9, 37), (1, 1, 16, 8, 21, 35)])
def test_conv2d_row_pitch(dev, bf16x3, mode, N, I, OC, H, W):
    """conv_launch(row_pitch=True): the padded-row output view holds exactly the dense result (all kernels, split-K too)."""
    from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
    if bf16x3 and not cg.bf16x3_eligible(I, H, W, 3, mode):
        pytest.skip('shape not on the bf16x3 path')
    x, w = _gen((N, I, H, W), 80).to(dev), (_gen((OC, I, 3, 3), 81) / np.sqrt(I * 9)).to(dev)
    wt = cg.prep_weight_bf16x3(w) if bf16x3 else cg.prep_weight(w)
    for ksplit in (1, 2):
        dense = cg.conv_launch(x, wt, 3, mode, OC, ksplit=ksplit, bf16x3=bf16x3)
        pitched = cg.conv_launch(x, wt, 3, mode, OC, ksplit=ksplit, bf16x3=bf16x3, row_pitch=True)
        assert pitched.shape == dense.shape and pitched.stride(2) % 4 == 0 and pitched.stride(1) == pitched.shape[2] * pitched.stride(2)
        assert torch.equal(pitched, dense), (mode, ksplit)


@pytest.mark.parametrize('N,I,OC,H,W', [(2, 64, 128, 65, 65), (1, 128, 70, 129, 131), (1, 16, 64, 35, 67), (1, 32, 8, 9, 9), (2, 32, 64, 4, 70),
                                       (1, 256, 256, 66, 64)])
def test_conv2d_stride2_bf16x3(dev, N, I, OC, H, W):
    """Split-bf16 polyphase stride-2 conv vs F.conv2d(stride=2), plain / split-K / fused epilogue with residual."""
    import torch.nn.functional as F
    from next3d_amd import _lib
    from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
    x, w = _gen((N, I, H, W), 100), _gen((OC, I, 3, 3), 101) / np.sqrt(I * 9)
    b = _gen((OC,), 102)
    ref = F.conv2d(x, w, stride=2)
    res = _gen(tuple(ref.shape), 103)
    t = lambda a: a.to(dev)
    wt16 = cg.prep_weight_bf16x3(t(w))
    for ksplit in (1, None, 2):
        y = cg.conv_launch(t(x), wt16, 3, 1, OC, ksplit=ksplit, bf16x3=True)
        assert y.shape == ref.shape
        err = float((y.cpu() - ref).abs().max())
        assert err <= 1e-4 * max(1.0, float(ref.abs().max())), (ksplit, err)
    ref2 = O.bias_act(ref * 0.5, b, act='lrelu') + res
    tb, tres = t(b), t(res)
    y = cg.conv_launch(t(x), wt16, 3, 1, OC, epilogue=_lib.make_epilogue(const_scale=0.5, bias=tb, act='lrelu', residual=tres), bf16x3=True)
    err = float((y.cpu() - ref2).abs().max())
    assert err <= 1e-4 * max(1.0, float(ref2.abs().max())), err
    # pitched input rows (what the FIR in front of the layer hands over): NaNs in the padding must not leak
    buf = torch.full((N, I, H, (W + 3) // 4 * 4 + 4), float('nan'), device=dev)
    xv = buf[..., :W]
    xv.copy_(t(x))
    yp = cg.conv_launch(xv, wt16, 3, 1, OC, ksplit=1, bf16x3=True)
    assert torch.equal(yp, cg.conv_launch(t(x), wt16, 3, 1, OC, ksplit=1, bf16x3=True))


@pytest.mark.parametrize('shape', [(2, 3, 16, 16), (1, 2, 128, 256), (1, 2, 64, 64), (2, 1, 33, 128)])
def test_upfirdn2d_pad2_pitched_output(dev, shape):
    """The FIR in front of a stride-2 conv (pad 2, output (H+1) x (W+1)) written with rows padded to 16 bytes: the
    [..., :OW] view equals the dense result (aligned float4 kernel when the input rows are aligned)."""
    from next3d_amd.torch_utils.ops import upfirdn2d
    x = _gen(shape, 75)
    f = O.setup_filter((1, 3, 3, 1)) + 0.01 * torch.arange(16.).reshape(4, 4)
    ref = O.upfirdn2d(x, f, padding=[2, 2, 2, 2])
    y = upfirdn2d.upfirdn2d(x.to(dev), f.to(dev), padding=[2, 2, 2, 2], _row_pitch=True)
    assert y.shape == ref.shape and y.stride(2) % 4 == 0 and y.stride(2) >= y.shape[3]
    _close(y, ref, atol=1e-5)


@pytest.mark.parametrize('bf16x3', [False, True])
@pytest.mark.parametrize('N,I,OC,H,W', [(2, 128, 96, 32, 32), (1, 32, 512, 16, 16), (2, 512, 3, 8, 8), (1, 64, 40, 6, 10), (1, 256, 130, 20, 12),
                                       (4, 512, 96, 4, 4), (1, 128, 32, 64, 48), (1, 128, 96, 72, 80), (2, 64, 70, 66, 68)])
def test_conv1x1_and_fused_skip_upsample(dev, bf16x3, N, I, OC, H, W):
    """1x1 conv (toRGB / fromRGB): fp32-MFMA and split-bf16 kernels vs F.conv2d, with style, bias, clamp, a dense residual,
    and the fused skip path  img = upsample2d(img_lowres) + toRGB(x)  (n3d_epilogue.residual_up_filter)."""
    import torch.nn.functional as F
    from next3d_amd import _lib
    from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
    x, w = _gen((N, I, H, W), 90), _gen((OC, I, 1, 1), 91) / np.sqrt(I)
    st, b = _gen((N, I), 92), _gen((OC,), 93)
    res, low = _gen((N, OC, H, W), 94), _gen((N, OC, H // 2, W // 2), 95)
    f = O.setup_filter((1, 3, 3, 1)) + 0.01 * torch.arange(16.).reshape(4, 4)       # asymmetric taps: catches index flips
    t = lambda a: a.to(dev)
    wt = cg.prep_weight_bf16x3(t(w)) if bf16x3 else cg.prep_weight(t(w))
    tol = lambda ref: (1e-4 if bf16x3 else 2e-5) * max(1.0, float(ref.abs().max()))
    core = F.conv2d(x * st[:, :, None, None], w)
    y = cg.conv_launch(t(x), wt, 1, 0, OC, style=t(st), bf16x3=bf16x3)
    assert float((y.cpu() - core).abs().max()) <= tol(core)
    ref = O.bias_act(core, b, clamp=0.7) + res
    tb, tres, tlow, tf = t(b), t(res), t(low), t(f)
    y = cg.conv_launch(t(x), wt, 1, 0, OC, style=t(st), epilogue=_lib.make_epilogue(bias=tb, clamp=0.7, residual=tres), bf16x3=bf16x3)
    assert float((y.cpu() - ref).abs().max()) <= tol(ref)
    ref = O.bias_act(core, b, clamp=0.7) + O.upsample2d(low, f)
    y = cg.conv_launch(t(x), wt, 1, 0, OC, style=t(st), bf16x3=bf16x3,
                       epilogue=_lib.make_epilogue(bias=tb, clamp=0.7, residual=tlow, residual_up_filter=tf))
    assert float((y.cpu() - ref).abs().max()) <= tol(ref)
    if not bf16x3:          # the split-K epilogue kernel takes the same fused path
        y = cg.conv_launch(t(x), wt, 1, 0, OC, style=t(st), ksplit=2,
                           epilogue=_lib.make_epilogue(bias=tb, clamp=0.7, residual=tlow, residual_up_filter=tf))
        assert float((y.cpu() - ref).abs().max()) <= tol(ref)


def test_filtered_lrelu_fma_and_small_kernels(dev):
    from next3d_amd import _lib
    from next3d_amd.torch_utils.ops import filtered_lrelu, fma, upfirdn2d
    x, b = _gen((2, 6, 18, 18), 70), _gen((6,), 71)
    for fu_t, fd_t, up, down, pad in (([1, 3, 3, 1], [1, 3, 3, 1], 2, 2, [2, 1, 2, 1]), ([1, 2, 1], None, 1, 1, 1),
                                      (list(range(1, 13)), list(range(1, 13)), 2, 2, 10)):
        fu = upfirdn2d.setup_filter(fu_t)
        fd = upfirdn2d.setup_filter(fd_t) if fd_t is not None else None
        ref = O.filtered_lrelu(x, fu, fd, b, up=up, down=down, padding=pad, gain=1.7, slope=0.1, clamp=0.8)
        y = filtered_lrelu.filtered_lrelu(x.to(dev), fu.to(dev), None if fd is None else fd.to(dev), b.to(dev), up=up, down=down,
                                          padding=pad, gain=1.7, slope=0.1, clamp=0.8)
        _close(y, ref, atol=2e-5, rtol=1e-4)
    a, bb, cc = _gen((2, 5, 7, 9), 72), _gen((2, 5, 1, 1), 73), _gen((7, 9), 74)
    _close(fma.fma(a.to(dev), bb.to(dev), cc.to(dev)), torch.addcmul(cc, a, bb), atol=1e-6)
    img = _gen((2, 3, 8, 8), 75) * 2
    out = torch.empty(img.shape, dtype=torch.uint8, device=dev)
    _lib.check(_lib.lib().n3d_to_uint8(_lib.ptr(img.to(dev)), _lib.ptr(out), img.numel(), _lib.stream()))
    assert torch.equal(out.cpu(), (img * 127.5 + 128).clamp(0, 255).to(torch.uint8))


def test_reference_fp16_block_call_sequence_through_operator_layer(dev):
    """the reference's DEFAULT super-resolution mode is fp16 (superresolution.py:36,210-217 — no script passes
    force_fp32).  This drives the call sequence of one fp16 SynthesisBlock (networks_stylegan2.py:548-586 -> modulated_conv2d
    :56-91 fp16 branch -> grouped conv2d_resample, bias_act with conv_clamp=256, toRGB, fp32 skip image) through the B1
    operator layer with fp16 / channels_last tensors, and compares with the CPU oracle run on fp32 arithmetic with the
    SAME fp16 rounding points (storage after every conv / FIR / bias_act).  Tolerance: the two sides differ only by the
    accumulation order inside a convolution, which can move a result across an fp16 rounding boundary -> a few fp16 ulps
    (2^-11 relative) of the tensor's magnitude."""
    from next3d_amd.torch_utils.ops import bias_act, conv2d_resample, upfirdn2d
    q = lambda t: t.half().float()
    N, I, O_, R, clamp = 2, 32, 64, 16, 256
    x0, wl = _gen((N, I, R, R), 50), _gen((N, 512), 51)
    P = {'c0.w': _gen((O_, I, 3, 3), 52), 'c0.b': _gen((O_,), 53) * 0.1, 'c0.aw': _gen((I, 512), 54), 'c0.ab': torch.ones(I),
         'c0.noise': _gen((2 * R, 2 * R), 55), 'c0.ns': torch.tensor(0.1),
         'c1.w': _gen((O_, O_, 3, 3), 56), 'c1.b': _gen((O_,), 57) * 0.1, 'c1.aw': _gen((O_, 512), 58), 'c1.ab': torch.ones(O_),
         'c1.noise': _gen((2 * R, 2 * R), 59), 'c1.ns': torch.tensor(0.1),
         'rgb.w': _gen((3, O_, 1, 1), 60), 'rgb.b': _gen((3,), 61) * 0.1, 'rgb.aw': _gen((O_, 512), 62), 'rgb.ab': torch.ones(O_)}
    img0 = _gen((N, 3, R, R), 63)
    f = O.setup_filter((1, 3, 3, 1))

    def modconv(x, wt, styles, f_, up, demod, ops, dt):
        """modulated_conv2d, fused branch, fp16 pre-normalisation included (:56-59)."""
        o, i, kh, kw = wt.shape
        if dt == torch.float16 and demod:
            wt = wt * (1 / np.sqrt(i * kh * kw) / wt.norm(float('inf'), dim=[1, 2, 3], keepdim=True))
            styles = styles / styles.norm(float('inf'), dim=1, keepdim=True)
        w = wt.unsqueeze(0) * styles.reshape(N, 1, -1, 1, 1)
        if demod:
            w = w * (w.square().sum(dim=[2, 3, 4]) + 1e-8).rsqrt().reshape(N, -1, 1, 1, 1)
        return ops['resample'](x.reshape(1, -1, *x.shape[2:]), w.reshape(-1, i, kh, kw), f_, up, kh // 2).reshape(N, -1, x.shape[2] * up, x.shape[3] * up)

    def block(x, img, ops, dt, to):
        affine = lambda k: to(wl) @ (to(P[k + '.aw']).t() / np.sqrt(512)) + to(P[k + '.ab'])
        x = ops['cast'](x)
        x = modconv(x, to(P['c0.w']), affine('c0'), to(f), 2, True, ops, dt)
        x = ops['add'](x, to(P['c0.noise'] * P['c0.ns']))
        x = ops['bias_act'](x, to(P['c0.b']), 'lrelu', clamp)
        x = modconv(x, to(P['c1.w']), affine('c1'), to(f), 1, True, ops, dt)
        x = ops['add'](x, to(P['c1.noise'] * P['c1.ns']))
        x = ops['bias_act'](x, to(P['c1.b']), 'lrelu', clamp)
        y = modconv(x, to(P['rgb.w']), affine('rgb') / np.sqrt(O_), None, 1, False, ops, dt)
        y = ops['bias_act'](y, to(P['rgb.b']), 'linear', clamp)
        img = ops['up'](img, to(f)) + y.float()
        return x, img

    gpu = {'cast': lambda t: t.to(dtype=torch.float16, memory_format=torch.channels_last),
           'resample': lambda x, w, f_, up, pad: conv2d_resample.conv2d_resample(x=x, w=w.to(x.dtype), f=f_, up=up, padding=pad, groups=N,
                                                                               flip_weight=(up == 1)),
           'add': lambda x, nz: x.add_(nz.to(x.dtype)),
           'bias_act': lambda x, b, act, cl: bias_act.bias_act(x, b.to(x.dtype), act=act, clamp=cl),
           'up': lambda img, f_: upfirdn2d.upsample2d(img, f_)}
    cpu = {'cast': q,
           'resample': lambda x, w, f_, up, pad: O.conv2d_resample(x, q(w), f=f_, up=up, padding=pad, groups=N, flip_weight=(up == 1), quant=q),
           'add': lambda x, nz: q(x + q(nz)),
           'bias_act': lambda x, b, act, cl: q(O.bias_act(x, q(b), act=act, clamp=cl)),
           'up': lambda img, f_: O.upsample2d(img, f_)}
    xg, ig = block(x0.to(dev), img0.to(dev), gpu, torch.float16, lambda t: t.to(dev))
    xc, ic = block(x0, img0, cpu, torch.float16, lambda t: t)
    assert xg.dtype == torch.float16 and ig.dtype == torch.float32
    ulp = 2.0 ** -11
    ex, ei = float((xg.float().cpu() - xc).abs().max()), float((ig.cpu() - ic).abs().max())
    print('fp16 block: x absmax', float(xc.abs().max()), 'err', ex, '| img absmax', float(ic.abs().max()), 'err', ei)
    assert ex <= 8 * ulp * float(xc.abs().max()) and ei <= 8 * ulp * float(ic.abs().max())


def test_operator_layer_fp16_and_dtype_errors(dev):
    """fp16 tensors run (fp32 accumulation, fp16 storage); unsupported or mixed dtypes raise instead of reaching a float32
    kernel with a non-float32 pointer."""
    from next3d_amd.torch_utils.ops import conv2d_gradfix as cg, conv2d_resample, filtered_lrelu, fma, upfirdn2d
    x, w = _gen((2, 16, 12, 12), 70), _gen((8, 16, 3, 3), 71) / 12
    f = O.setup_filter((1, 3, 3, 1))
    q = lambda t: t.half().float()
    y = conv2d_resample.conv2d_resample(x.half().to(dev), w.half().to(dev), f=f.to(dev), padding=1)
    assert y.dtype == torch.float16
    _close(y.float(), q(O.conv2d_resample(q(x), q(w), f=f, padding=1)), atol=4e-3, rtol=2e-3)
    y = upfirdn2d.upfirdn2d(x.half().to(dev), f.to(dev), up=2, padding=[2, 1, 2, 1], gain=4)
    assert y.dtype == torch.float16
    _close(y.float(), q(O.upfirdn2d(q(x), f, up=2, padding=[2, 1, 2, 1], gain=4)), atol=2e-3, rtol=2e-3)
    y = filtered_lrelu.filtered_lrelu(x.half().to(dev), fu=f.to(dev), fd=f.to(dev), b=_gen((16,), 72).half().to(dev), up=2, down=2, padding=[3, 2, 3, 2])
    assert y.dtype == torch.float16
    _close(y.float(), q(O.filtered_lrelu(q(x), fu=f, fd=f, b=q(_gen((16,), 72)), up=2, down=2, padding=[3, 2, 3, 2])), atol=4e-3, rtol=2e-3)
    a, b, c = _gen((2, 4, 5, 5), 73), _gen((2, 4, 1, 1), 74), _gen((1, 1, 5, 5), 75)
    y = fma.fma(a.half().to(dev), b.half().to(dev), c.half().to(dev))
    assert y.dtype == torch.float16
    _close(y.float(), q(q(a) * q(b) + q(c)), atol=2e-3, rtol=2e-3)
    for bad in (lambda: conv2d_resample.conv2d_resample(x.double().to(dev), w.double().to(dev), padding=1),
                lambda: cg.conv2d(x.half().to(dev), w.to(dev), padding=1),
                lambda: cg.conv2d(x.to(dev), w.half().to(dev), padding=1),
                lambda: cg.conv_launch(x.to(torch.bfloat16).to(dev), cg.prep_weight(w.to(dev)), 3, 0, 8),
                lambda: cg.prep_weight(w.to(torch.bfloat16).to(dev)),
                lambda: upfirdn2d.upfirdn2d(x.double().to(dev), f.to(dev)),
                lambda: upfirdn2d.upfirdn2d(x.to(dev), f.double().to(dev)),
                lambda: fma.fma(a.half().to(dev), b.to(dev), c.to(dev)),
                lambda: filtered_lrelu.filtered_lrelu(x.to(torch.bfloat16).to(dev))):
        with pytest.raises(RuntimeError):
            bad()


@pytest.mark.parametrize('up', [1, 2])
def test_synthesis_layer_random_noise_mode(dev, monkeypatch, up):
    """noise_mode='random' (the reference's default, networks_stylegan2.py:318-319: a fresh N(0,1) image per sample): with the
    draw replaced by the learned noise image it must reproduce noise_mode='const' bit for bit; with real draws the samples of
    a batch get different noise."""
    from next3d_amd import layers
    res, ic, oc, N = 16, 32, 64, 2
    P = {'L.weight': _gen((oc, ic, 3, 3), 80), 'L.bias': _gen((oc,), 81) * 0.1, 'L.affine.weight': _gen((ic, 512), 82),
         'L.affine.bias': torch.ones(ic), 'L.noise_const': _gen((res, res), 83), 'L.noise_strength': torch.tensor(0.5)}
    L = layers.PreparedConv({k: v.to(dev) for k, v in P.items()}, 'L', True)
    fir = O.setup_filter((1, 3, 3, 1)).to(dev)
    x = _gen((1, ic, res // up, res // up), 84).to(dev).expand(N, -1, -1, -1)          # identical samples, identical latents
    w = _gen((1, 512), 85).to(dev).expand(N, -1).contiguous()
    y_const = layers.synthesis_layer(L, x, w, fir, up=up, noise_mode='const')
    y_rand = layers.synthesis_layer(L, x, w, fir, up=up, noise_mode='random')
    assert y_rand.shape == y_const.shape and not torch.equal(y_rand[0], y_rand[1]) and not torch.equal(y_rand, y_const)
    monkeypatch.setattr(torch, 'randn', lambda shape, **k: L.noise_const.unsqueeze(0).expand(*shape).contiguous())
    y_fixed = layers.synthesis_layer(L, x, w, fir, up=up, noise_mode='random')
    assert torch.equal(y_fixed, y_const)


def _to_split8(x):
    """float32 [N,C,H,W] -> _lib.Split8 with the kernels' own split (hi = bf16(x), lo = bf16(x - hi)), built with torch."""
    from next3d_amd import _lib
    n, c, h, w = x.shape
    hi = x.bfloat16()
    lo = (x - hi.float()).bfloat16()
    s = _lib.Split8(n, c, h, w, x.device)
    t = torch.stack([hi, lo], 1).reshape(n, 2, c // 8, 8, h, w).permute(0, 1, 2, 4, 5, 3).contiguous()
    s.data.copy_(t.reshape(-1))
    return s


@pytest.mark.parametrize('N,I,OC,H,W', [(4, 32, 128, 64, 64), (2, 32, 256, 128, 128), (1, 64, 96, 37, 53), (3, 32, 160, 16, 32), (1, 256, 512, 32, 32),
                                       (1, 32, 32, 9, 300)])
def test_conv1x1_split8_output_equals_conversion_pass(dev, monkeypatch, N, I, OC, H, W):
    """The 1x1 kernel's split8 epilogue (y_layout = N3D_LAYOUT_SPLIT8: the encoders' fromrgb in front of conv1) writes exactly
    the bits n3d_split8_from_nchw makes of its float32 result — plain, with bias / lrelu / clamp, with the skip residual —
    and `layers.conv2d_layer(sole_consumer=)` followed by the 3x3 layer is bit-identical with and without it."""
    from next3d_amd import _lib, layers as L
    from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
    x = _gen((N, I, H, W), 140).to(dev)
    w = (_gen((OC, I, 1, 1), 141) / np.sqrt(I)).to(dev)
    wt16 = cg.prep_weight_bf16x3(w)
    b, res = _gen((OC,), 142).to(dev), _gen((N, OC, H, W), 143).to(dev)
    for kw in (dict(), dict(bias=b, act='lrelu', gain=1.3, clamp=1.5), dict(bias=b, residual=res, const_scale=0.7)):
        ref = cg.split8_from_nchw(cg.conv_launch(x, wt16, 1, 0, OC, epilogue=_lib.make_epilogue(**kw), bf16x3=True))
        y = cg.conv_launch(x, wt16, 1, 0, OC, epilogue=_lib.make_epilogue(**kw), bf16x3=True, out_split8=True)
        assert isinstance(y, _lib.Split8) and y.shape == (N, OC, H, W)
        if H * W <= 4096 and I % 128 == 0:     # the float32 result of this shape comes from the split-K 1x1 kernel (another summation order)
            _close(y.to_float(), ref.to_float(), atol=2e-5 * float(ref.to_float().abs().max()), rtol=0)
        else:
            assert torch.equal(y.data.view(torch.int16), ref.data.view(torch.int16)), kw.keys()
    with pytest.raises(RuntimeError):
        cg.conv_launch(x, cg.prep_weight(w), 1, 0, OC, out_split8=True)                   # only the split-bf16 1x1 kernel writes it
    if OC % 64 == 0 and cg.split8_eligible(N, OC, OC, H, W):
        P = {'a.weight': w, 'a.bias': b, 'c.weight': (_gen((OC, OC, 3, 3), 144)).to(dev), 'c.bias': b}
        A, C = L.PreparedConv(P, 'a', modulated=False), L.PreparedConv(P, 'c', modulated=False)
        fir = O.setup_filter((1, 3, 3, 1)).to(dev)
        outs = []
        for direct in (True, False):
            monkeypatch.setattr(L, 'DIRECT_SPLIT8', direct)
            t = L.conv2d_layer(A, x, fir, residual=res, sole_consumer=C)
            assert isinstance(t, _lib.Split8) == direct
            outs.append(L.conv2d_layer(C, t, fir, activation='lrelu'))
        if H * W <= 4096 and I % 128 == 0:
            _close(outs[0], outs[1], atol=2e-5 * float(outs[1].abs().max()), rtol=0)
        else:
            assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize('N,I,OC,H,W', [(2, 256, 3, 128, 128), (1, 128, 3, 96, 200), (2, 512, 3, 80, 80), (4, 512, 3, 64, 64), (2, 512, 96, 32, 32),
                                       (1, 256, 96, 128, 128), (1, 64, 40, 37, 53), (3, 128, 128, 16, 16), (1, 32, 8, 9, 300)])
def test_conv1x1_side_output_equals_conversion_pass(dev, N, I, OC, H, W):
    """n3d_conv2d_desc.side_split8 (toRGB also writing x * the next block's styles as split8): the main output is unchanged
    bit for bit and the side tensor is exactly n3d_split8_from_nchw(x, side_style) — on the pixel-tiled kernels (weights
    resident / streamed, 1..4 channel tiles) and the split-K kernel of the <= 64 x 64 layers, with the fused skip upsample."""
    from next3d_amd import _lib
    from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
    x = _gen((N, I, H, W), 150).to(dev)
    wt16 = cg.prep_weight_bf16x3((_gen((OC, I, 1, 1), 151) / np.sqrt(I)).to(dev))
    st, st2, b = _gen((N, I), 152).to(dev), _gen((N, I + 5), 153).to(dev)[:, 2:I + 2], _gen((OC,), 154).to(dev)      # st2: strided rows
    low, f = _gen((N, OC, H // 2, W // 2), 155).to(dev), O.setup_filter((1, 3, 3, 1)).to(dev)
    kws = [dict(bias=b, clamp=0.8)]
    if H % 2 == 0 and W % 2 == 0:
        kws.append(dict(bias=b, clamp=0.8, residual=low, residual_up_filter=f))
    for kw in kws:
        ref = cg.conv_launch(x, wt16, 1, 0, OC, style=st, epilogue=_lib.make_epilogue(**kw), bf16x3=True)
        y, side = cg.conv_launch(x, wt16, 1, 0, OC, style=st, epilogue=_lib.make_epilogue(**kw), bf16x3=True, side_style=st2)
        assert torch.equal(y, ref)
        want = cg.split8_from_nchw(x, st2)
        assert side.shape == (N, I, H, W) and torch.equal(side.data.view(torch.int16), want.data.view(torch.int16))
    with pytest.raises(RuntimeError):
        cg.conv_launch(x, wt16, 1, 0, OC, style=st, bf16x3=True, side_style=st2.double())
    if OC <= 128:
        xv = torch.empty(N, I + 32, H, W, device=dev)[:, 32:]          # a channel-slice view (batch stride > I*H*W), as in the U-Net's buffers
        xv.copy_(x)
        y, side = cg.conv_launch(xv, wt16, 1, 0, OC, style=st, bf16x3=True, side_style=st2)
        assert torch.equal(side.data.view(torch.int16), cg.split8_from_nchw(x, st2).data.view(torch.int16))


@pytest.mark.parametrize('N,I,OC,H,W', [(4, 256, 256, 128, 128), (4, 64, 512, 48, 80), (2, 128, 100, 256, 256), (8, 32, 64, 128, 160)])
def test_presplit_conv_matches_plain_bf16x3_kernel(dev, N, I, OC, H, W):
    """conv2d_ps1 / ps2_bf16x3_kernel (split8 input staged by LDS-DMA) against the register-staged split-bf16 kernels on the same
    operands: same products, same accumulation order per output -> expected bit-identical; ragged sizes exercise the halo
    coming from the buffer descriptor's range check and the masked stores."""
    from next3d_amd import _lib
    from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
    assert cg.split8_eligible(N, I, OC, H, W)
    x = _gen((N, I, H, W), 100).to(dev)
    w = (_gen((OC, I, 3, 3), 101) / np.sqrt(9 * I)).to(dev)
    wt16 = cg.prep_weight_bf16x3(w)
    dco, bias, noise = (1 + 0.1 * _gen((N, OC), 102)).to(dev), _gen((OC,), 103).to(dev), _gen((H, W), 104).to(dev)
    ns = torch.tensor(0.3, device=dev)
    for kw in (dict(), dict(row_scale=dco, noise=noise, noise_strength=ns, bias=bias, act='lrelu', gain=1.4, clamp=2.0)):
        ref = cg.conv_launch(x, wt16, 3, 0, OC, epilogue=_lib.make_epilogue(**kw), bf16x3=True)
        y = cg.conv_launch(_to_split8(x), wt16, 3, 0, OC, epilogue=_lib.make_epilogue(**kw), bf16x3=True)
        print('presplit vs plain: max abs diff', float((y - ref).abs().max()), 'bit-identical', bool(torch.equal(y, ref)))
        _close(y, ref, atol=1e-5, rtol=1e-5)
    _close(y, O.bias_act(torch.nn.functional.conv2d(x.cpu(), w.cpu(), padding=1) * dco.cpu()[:, :, None, None] + noise.cpu() * 0.3, bias.cpu(),
                         act='lrelu', gain=1.4, clamp=2.0), atol=2e-4, rtol=1e-4)
    with pytest.raises(RuntimeError):
        cg.conv_launch(_to_split8(x), wt16, 3, 0, OC, style=torch.ones(N, I, device=dev), bf16x3=True)      # the split8 input is modulated already


@pytest.mark.parametrize('N,I,OC,H,W,C', [(4, 32, 128, 256, 256, 3), (2, 64, 128, 200, 330, 3), (2, 16, 64, 200, 330, 4), (3, 32, 100, 200, 200, 1),
                                          (4, 32, 128, 256, 256, 32), (2, 64, 256, 128, 128, 32), (2, 16, 64, 200, 330, 32), (3, 32, 100, 200, 202, 17), (4, 512, 512, 64, 64, 32)])
def test_fused_torgb_epilogue_matches_separate_layers(dev, N, I, OC, H, W, C):
    """n3d_conv2d_desc.rgb_* + n3d_rgb_combine (a network's LAST 3x3 layer evaluating its toRGB in the epilogue; the feature map is never
    written) against the two layers run separately — the same pre-split kernel writing x, then the 1x1 toRGB kernel with the skip-image
    upsample in its epilogue — and against float32 ATen; big grids (one LDS buffer, two workgroups per CU) and small ones (two buffers),
    ragged tiles, O not a multiple of 64, 1 / 3 / 4 colours on the VALU and 17 / 32 colours (round 5: the backbones' toRGB layers) as an epilogue contraction on
    the matrix cores (split-bf16: 3 products per MAC, the 1x1 kernel's arithmetic); and with the split8 side output for the layer's second reader (bit-identical to
    n3d_split8_from_nchw of the separately written feature map)."""
    from next3d_amd import _lib, layers
    from next3d_amd.torch_utils.ops import conv2d_gradfix as cg, upfirdn2d as uf
    assert cg.split8_eligible(N, I, OC, H, W) and _lib.lib().n3d_conv2d_split8_ksplit(N, I, OC, H, W) == 1
    x = _gen((N, I, H, W), 130).to(dev)
    w = (_gen((OC, I, 3, 3), 131) / np.sqrt(9 * I)).to(dev)
    wt16 = cg.prep_weight_bf16x3(w)
    dco, bias, noise = (1 + 0.1 * _gen((N, OC), 132)).to(dev), _gen((OC,), 133).to(dev), _gen((H, W), 134).to(dev)
    ns = torch.tensor(0.3, device=dev)
    wrgb, srgb, brgb = (_gen((C, OC), 135) / np.sqrt(OC)).to(dev), (1 + 0.2 * _gen((N, OC), 136)).to(dev), _gen((C,), 137).to(dev)
    fir = uf.setup_filter([1, 3, 3, 1]).to(dev)
    even = H % 2 == 0 and W % 2 == 0
    img_lo = _gen((N, C, H // 2, W // 2), 138).to(dev) if even else None
    kw = dict(row_scale=dco, noise=noise, noise_strength=ns, bias=bias, act='lrelu', gain=1.4, clamp=2.0)
    xs = _to_split8(x)
    feat = cg.conv_launch(xs, wt16, 3, 0, OC, epilogue=_lib.make_epilogue(**kw), bf16x3=True)
    part = cg.conv_launch(xs, wt16, 3, 0, OC, epilogue=_lib.make_epilogue(**kw), bf16x3=True, rgb=(wrgb, srgb))
    assert tuple(part.shape) == (N, (OC + 63) // 64, C, H, W)
    tkw = dict(bias=brgb, clamp=1.5, residual=img_lo, residual_up_filter=fir if even else None)
    got = cg.rgb_combine(part, _lib.make_epilogue(**tkw))
    want = torch.einsum('nohw,co,no->nchw', feat.double(), wrgb.double(), srgb.double()).float() + brgb[None, :, None, None]
    want = want.clamp(-1.5, 1.5)
    if even:
        want = want + uf.upsample2d(img_lo, fir)
    print('fused toRGB vs float64 sum over the kernel\'s own feature map: max abs diff', float((got - want).abs().max()))
    if C <= 4:
        _close(got, want, atol=2e-6, rtol=2e-6)                            # float32 FMA chain over <= 128 channels
    else:
        _close(got, want, atol=1e-4, rtol=1e-5)                            # split-bf16 products: 2^-17 of the summed |products| (as the separate 1x1 kernel below)
    if OC % 16 == 0:                                                       # and against the separate 1x1 split-bf16 toRGB kernel (3 bf16 products per MAC)
        sep = cg.conv_launch(feat, cg.prep_weight_bf16x3(wrgb.reshape(C, OC, 1, 1)), 1, 0, C, style=srgb, epilogue=_lib.make_epilogue(**tkw), bf16x3=True)
        _close(got, sep, atol=1e-4, rtol=1e-5)                             # (that kernel's error: 2^-17 of the summed |products|; measured 2.4e-5)
    if OC % 8 == 0:      # ... and with the layer's second reader served from the same epilogue: its output times that reader's styles, as split8
        st2 = (1 + 0.3 * _gen((N, OC), 139)).to(dev)
        part2, side = cg.conv_launch(xs, wt16, 3, 0, OC, epilogue=_lib.make_epilogue(**kw), bf16x3=True, rgb=(wrgb, srgb), side_style=st2)
        want_side = cg.split8_from_nchw(feat, st2)
        assert side.shape == (N, OC, H, W) and torch.equal(side.data.view(torch.int16), want_side.data.view(torch.int16))
        _close(cg.rgb_combine(part2, _lib.make_epilogue(**tkw)), got, atol=1e-6, rtol=1e-6)      # (another kernel instance: same sums, FMA order equal)
    with pytest.raises(RuntimeError):
        cg.conv_launch(x, wt16, 3, 0, OC, style=torch.ones(N, I, device=dev), bf16x3=True, rgb=(wrgb, srgb))     # NCHW input: not the pre-split kernel
    with pytest.raises(RuntimeError):
        cg.conv_launch(xs, wt16, 3, 0, OC, bf16x3=True, rgb=(torch.ones(33, OC, device=dev), srgb))               # more than 32 colours


@pytest.mark.parametrize('N,C,H,W,pad', [(2, 32, 64, 64, 2), (1, 16, 37, 101, 2), (3, 8, 16, 20, 1), (1, 64, 128, 128, 2)])
@pytest.mark.parametrize('sep', ['1', '0'])
def test_fir4_split8_from_nchw_matches_float_fir(dev, monkeypatch, N, C, H, W, pad, sep):
    """n3d_fir4_split8_nchw / n3d_fir4_split8_nchw_sep (the FIR in front of a stride-2 convolution, float32 NCHW in, split8 out;
    sep '1': the separable kernel, '0': the 16-tap one) against the float32 FIR of the same library: hi + lo reproduces it to the
    16 bits the pair carries — dense input, a row-pitched batch-strided view, and with an epilogue + the next layer's styles."""
    from next3d_amd import _lib
    from next3d_amd.torch_utils.ops import upfirdn2d as uf
    monkeypatch.setattr(uf, 'FIR_SEP', 'all' if sep == '1' else False)
    x = _gen((N, C, H, W), 120).to(dev)
    f = uf.setup_filter([1, 3, 3, 1]).to(dev)
    ref = uf.upfirdn2d(x, f, padding=[pad] * 4)
    y = uf._fir4_split8_nchw(x, f, pad)
    assert tuple(y.shape) == tuple(ref.shape)
    print('fir4 nchw -> split8: max |hi + lo - fir|', float((y.to_float() - ref).abs().max()))
    _close(y.to_float(), ref, atol=1e-6, rtol=2.0 ** -15)                 # hi + lo carries 16 significant bits
    _close(ref.cpu(), O.upfirdn2d(x.cpu(), f.cpu(), padding=[pad] * 4), atol=1e-6, rtol=1e-6)
    xv = torch.zeros(N, C + 8, H, W + 3, device=dev)[:, 8:, :, :W]        # pitched rows, batch stride > C*H*pitch
    xv.copy_(x)
    _close(uf._fir4_split8_nchw(xv, f, pad).to_float(), ref, atol=1e-6, rtol=2.0 ** -15)
    bias, style = _gen((C,), 121).to(dev), (1 + 0.2 * _gen((N, C), 122)).to(dev)
    act = dict(bias=bias, act='lrelu', gain=float(np.sqrt(2)), clamp=1.5)
    ref2 = uf.upfirdn2d(x, f, padding=[pad] * 4, gain=2, _epilogue=_lib.make_epilogue(**act)) * style[:, :, None, None]
    _close(uf._fir4_split8_nchw(x, f, pad, gain=2, epilogue=_lib.make_epilogue(**act), out_scale=style).to_float(), ref2, atol=1e-6, rtol=2.0 ** -15)


@pytest.mark.parametrize('N,I,OC,H,W,ks', [(4, 128, 256, 257, 257, 1), (2, 64, 100, 65, 129, 1), (4, 512, 512, 65, 65, 4), (1, 32, 64, 33, 47, 2),
                                          (3, 16, 64, 40, 36, 1)])
def test_presplit_stride2_conv_matches_plain_kernel(dev, N, I, OC, H, W, ks):
    """conv2d_s2_ps_bf16x3_kernel (split8 input, (chunk, phase) stages by LDS-DMA with the de-interleave in the source
    addresses, three LDS buffers) against the register-staged stride-2 kernel on the same operands and against ATen; odd and
    even input sizes, ragged tiles, split-K."""
    from next3d_amd import _lib
    from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
    x = _gen((N, I, H, W), 110).to(dev)
    w = (_gen((OC, I, 3, 3), 111) / np.sqrt(9 * I)).to(dev)
    wt16 = cg.prep_weight_bf16x3(w)
    bias = _gen((OC,), 113).to(dev)
    for kw in (dict(), dict(const_scale=0.7, bias=bias, act='lrelu', gain=1.4, clamp=2.0)):
        ref = cg.conv_launch(x, wt16, 3, 1, OC, epilogue=_lib.make_epilogue(**kw), bf16x3=True, ksplit=ks)
        y = cg.conv_launch(_to_split8(x), wt16, 3, 1, OC, epilogue=_lib.make_epilogue(**kw), bf16x3=True, ksplit=ks)
        print('stride-2 presplit vs plain: max abs diff', float((y - ref).abs().max()), 'bit-identical', bool(torch.equal(y, ref)))
        _close(y, ref, atol=1e-5, rtol=1e-5)
    _close(y, O.bias_act(torch.nn.functional.conv2d(x.cpu(), w.cpu(), stride=2) * 0.7, bias.cpu(), act='lrelu', gain=1.4, clamp=2.0), atol=2e-4, rtol=1e-4)
    y2 = cg.conv_launch(cg.split8_from_nchw(x), wt16, 3, 1, OC, epilogue=_lib.make_epilogue(**kw), bf16x3=True, ksplit=ks)
    assert torch.equal(y2, y)                                             # the library's own conversion pass = the torch-built operand image


@pytest.mark.parametrize('sep', ['1', '0'])
@pytest.mark.parametrize('N,C,H,W', [(2, 32, 16, 32), (1, 64, 64, 64), (3, 16, 40, 24), (1, 8, 7, 100)])
def test_fir4_split8_matches_float_fir(dev, monkeypatch, N, C, H, W, sep):
    """n3d_fir4_split8 (c8 input -> FIR + layer epilogue + next layer's style + hi/lo split, split8 output) against the float32
    FIR path on the same values: hi + lo must reproduce style * fir_out to 2^-16 relative (what two bf16 halves carry)."""
    from next3d_amd import _lib
    from next3d_amd.torch_utils.ops import upfirdn2d as uf
    monkeypatch.setattr(uf, 'FIR_SEP', 'all' if sep == '1' else False)      # '1': the separable form (n3d_fir4_split8_sep), '0': the 16-tap kernel
    f = O.setup_filter((1, 3, 3, 1)).to(dev)
    zh, zw = 2 * H + 1, 2 * W + 1
    z = _gen((N, C, zh, zw), 110).to(dev)
    zc = _lib.C8(N, C, zh, zw, dev)
    zc.data.copy_(z.reshape(N, C // 8, 8, zh, zw).permute(0, 1, 3, 4, 2))
    assert torch.equal(zc.to_nchw(), z)
    bias, noise, ns = _gen((C,), 111).to(dev), _gen((2 * H, 2 * W), 112).to(dev), torch.tensor(0.2, device=dev)
    style = (1 + 0.2 * _gen((N, C), 113)).to(dev)
    act = dict(noise=noise, noise_strength=ns, bias=bias, act='lrelu', gain=float(np.sqrt(2)), clamp=3.0)
    ref = uf.upfirdn2d(z, f, padding=[1, 1, 1, 1], gain=4, _epilogue=_lib.make_epilogue(**act)) * style[:, :, None, None]
    s = uf._fir4_split8(zc, f, 4, _lib.make_epilogue(**act), style)
    assert s.shape == (N, C, 2 * H, 2 * W)
    _close(s.to_float(), ref, atol=1e-6, rtol=2.0 ** -15)
    t = s.data.reshape(N, 2, C // 8, 2 * H, 2 * W, 8).float()
    hi_expected = ref.bfloat16().float().reshape(N, C // 8, 8, 2 * H, 2 * W).permute(0, 1, 3, 4, 2)
    assert float((t[:, 0] - hi_expected).abs().max()) <= float(ref.abs().max()) * 2.0 ** -7      # hi is the bf16 rounding of the value


@pytest.mark.parametrize('N,I,OC,H,W', [(4, 256, 128, 64, 64), (2, 64, 64, 33, 40), (4, 32, 256, 128, 128), (2, 32, 64, 72, 96)])
def test_transposed_conv_channel_interleaved_output(dev, monkeypatch, N, I, OC, H, W):
    """The transposed split-bf16 kernel writing the c8 layout (what the FIR of the pre-split path reads) returns the same
    numbers as its NCHW output, bit for bit."""
    from next3d_amd import _lib
    from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
    x, style, dco = _gen((N, I, H, W), 120).to(dev), (1 + 0.1 * _gen((N, I), 121)).to(dev), (1 + 0.1 * _gen((N, OC), 122)).to(dev)
    wt16 = cg.prep_weight_bf16x3((_gen((OC, I, 3, 3), 123) / np.sqrt(9 * I)).to(dev))
    ref = cg.conv_launch(x, wt16, 3, 2, OC, style=style, epilogue=_lib.make_epilogue(row_scale=dco), bf16x3=True, ksplit=1)
    c8 = cg.conv_launch(x, wt16, 3, 2, OC, style=style, epilogue=_lib.make_epilogue(row_scale=dco), bf16x3=True, out_c8=True)
    assert c8.shape == tuple(ref.shape) and torch.equal(c8.to_nchw(), ref)
    with pytest.raises(RuntimeError):
        cg.conv_launch(x, wt16, 3, 2, OC, style=style, epilogue=_lib.make_epilogue(row_scale=dco, bias=dco[0]), bf16x3=True, out_c8=True)
    # the same layer with the input converted once (style multiplied in, operands split) and staged by LDS-DMA: same products
    xs = cg.split8_from_nchw(x, style)
    assert float((xs.to_float() - x * style[:, :, None, None]).abs().max()) <= 2.0 ** -15 * float((x * style[:, :, None, None]).abs().max())
    ps = cg.conv_launch(xs, wt16, 3, 2, OC, epilogue=_lib.make_epilogue(row_scale=dco), bf16x3=True, out_c8=True)
    print('transposed presplit vs register-staged: max abs diff', float((ps.to_nchw() - ref).abs().max()), 'bit-identical', bool(torch.equal(ps.to_nchw(), ref)))
    _close(ps.to_nchw(), ref, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize('N,I,OC,H,W,mode', [(4, 512, 512, 4, 4, 0), (4, 512, 512, 8, 8, 0), (1, 512, 512, 16, 16, 0), (4, 1024, 512, 8, 8, 0), (1, 512, 512, 4, 4, 0),
                                            (3, 256, 96, 8, 8, 0), (4, 512, 512, 4, 4, 2), (1, 512, 512, 8, 8, 2), (4, 512, 512, 9, 9, 1), (1, 512, 512, 17, 17, 1)])
def test_few_pixel_split_k_seam(dev, N, I, OC, H, W, mode):
    """Round 6: the few-pixel 3x3 kernels slice K over WORKGROUPS and reduce inside the launch (conv2d_sk_bf16x3.hip: write-through slabs, one arrival
    counter per output tile, the last arriver adds the slabs in slice order and applies the epilogue).  Against the same kernel with the whole K inside one
    workgroup (round 5's launch, SK_SEAM = False) and against float32 ATen; the arrival counters are zero again after every launch; 40 launches in a row and
    launches on three streams at once BESIDE chip-filling convolutions (uneven load, warm L1s: what an inter-workgroup hand-off has to survive) return the same bits."""
    import torch.nn.functional as F
    from next3d_amd import _lib
    from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
    x, w = _gen((N, I, H, W), 190), _gen((OC, I, 3, 3), 191) / np.sqrt(I * 9)
    s, d, b = _gen((N, I), 192), _gen((N, OC), 193).abs() + 0.5, _gen((OC,), 194)
    t = lambda a: a.to(dev)
    wt16 = cg.prep_weight_bf16x3(t(w))
    xs = x * s[:, :, None, None]
    ref = {0: lambda: F.conv2d(xs, w, padding=1), 1: lambda: F.conv2d(xs, w, stride=2), 2: lambda: F.conv_transpose2d(xs, w.transpose(0, 1), stride=2)}[mode]()
    ref = O.bias_act(ref * d[:, :, None, None], b, act='lrelu')
    fl, need = cg.sk_workspace(N, I, OC, H, W, mode)
    assert fl > 0 and 0 < need <= _lib.TICKET_COUNT, 'this shape is expected to run with K slices'
    run = lambda: cg.conv_launch(t(x), wt16, 3, mode, OC, style=t(s), epilogue=_lib.make_epilogue(row_scale=t(d), bias=t(b), act='lrelu'), bf16x3=True)
    old = cg.SK_SEAM
    try:
        cg.SK_SEAM = False
        y0 = run()
        cg.SK_SEAM = True
        y1 = run()
    finally:
        cg.SK_SEAM = old
    scale = max(1.0, float(ref.abs().max()))
    assert tuple(y1.shape) == tuple(ref.shape)
    assert float((y1.cpu() - ref).abs().max()) <= 1e-4 * scale
    assert float((y1 - y0).abs().max()) <= 2e-5 * scale                      # another summation order, the same products
    assert int(_lib.tickets().abs().sum()) == 0
    for _ in range(40):
        assert torch.equal(run(), y1)
    assert int(_lib.tickets().abs().sum()) == 0
    # three streams, each beside a chip-filling 3x3 layer of another stream
    big_x, big_w = torch.randn(2, 128, 128, 128, device=dev), cg.prep_weight_bf16x3(torch.randn(128, 128, 3, 3, device=dev) / 34)
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    outs = [[] for _ in streams]
    torch.cuda.synchronize()
    for it in range(12):
        for k, st in enumerate(streams):
            with torch.cuda.stream(st):
                if (it + k) % 3 != 2:
                    cg.conv_launch(big_x, big_w, 3, 0, 128, bf16x3=True)
                outs[k].append(run())
    torch.cuda.synchronize()
    assert all(torch.equal(o, y1) for lst in outs for o in lst)
    for st in streams:
        with torch.cuda.stream(st):
            assert int(_lib.tickets().abs().sum()) == 0
