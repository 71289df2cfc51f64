"""-m gpu parity: whole StyleGAN2 / StyleUNet / SR networks on libn3d.so vs the CPU oracle (seeded synthetic weights)."""
import pytest
import torch

from next3d_amd import spec
from oracle import networks as ON

pytestmark = pytest.mark.gpu


def _g(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


@pytest.fixture(params=['fp32', 'bf16x3'])
def precision(request):
    from next3d_amd import layers
    old = layers.PRECISION
    layers.set_precision(request.param)
    yield request.param
    layers.set_precision(old)


TOL = {'fp32': 2e-5, 'bf16x3': 3e-4}     # relative to the output scale; bf16x3 truncates operands at 2^-16


def _report(name, a, b, tol):
    a, b = a.detach().cpu(), b.detach().cpu()
    err = float((a - b).abs().max())
    scale = float(b.abs().max())
    print(f'{name}: max-abs err {err:.3e} (ref absmax {scale:.3e})')
    assert err <= tol * max(1.0, scale), (name, err, scale)


def test_texture_backbone(dev, precision):
    from next3d_amd import networks
    P = spec.synthetic_state_dict(0, only=lambda n: n.startswith('texture_backbone.synthesis'))
    Pd = {k: v.to(dev) for k, v in P.items()}
    ws = _g((2, 14, 512), 1)
    ref = ON.synthesis_network(P, 'texture_backbone.synthesis', ws)
    net = networks.SynthesisNet(Pd, 'texture_backbone.synthesis')
    _report('texture_backbone', net(ws.to(dev)), ref, TOL[precision])


def test_mouth_styleunet(dev, precision):
    from next3d_amd import networks
    P = spec.synthetic_state_dict(0, only=lambda n: n.startswith('mouth_backbone.synthesis'))
    Pd = {k: v.to(dev) for k, v in P.items()}
    ws, x = _g((2, 14, 512), 2), _g((2, 32, 64, 64), 3)
    ref = ON.styleunet_synthesis(P, 'mouth_backbone.synthesis', x, ws, in_size=64, final_size=4, num_cond_res=64)
    net = networks.StyleUNet(Pd, 'mouth_backbone.synthesis', in_size=64, final_size=4, num_cond_res=64)
    _report('mouth_backbone', net(x.to(dev), ws.to(dev)), ref, TOL[precision])


def test_neural_blending_styleunet(dev, precision):
    from next3d_amd import networks
    P = spec.synthetic_state_dict(0, only=lambda n: n.startswith('neural_blending.synthesis'))
    Pd = {k: v.to(dev) for k, v in P.items()}
    ws, x = _g((1, 14, 512), 4), _g((1, 32, 256, 256), 5)
    ref = ON.styleunet_synthesis(P, 'neural_blending.synthesis', x, ws, in_size=256, final_size=32, num_cond_res=256)
    net = networks.StyleUNet(Pd, 'neural_blending.synthesis', in_size=256, final_size=32, num_cond_res=256)
    _report('neural_blending', net(x.to(dev), ws.to(dev)), ref, TOL[precision])


@pytest.mark.parametrize('N,R', [(2, 64), (1, 128)])
def test_superresolution_fp32_and_fp16_modes(dev, N, R):
    """SuperresolutionHybrid8XDC (superresolution.py:264-290): the fp32 path (force_fp32, what the goldens pin) and the
    reference's DEFAULT on a GPU — fp16 blocks (sr_num_fp16_res > 0; no inference script passes force_fp32).
    The fp16 mode keeps float32 / split-bf16 arithmetic and rounds to float16 where the reference stores float16; the oracle
    emulates the reference's fp16 branch tensor by tensor (pre-normalised weights rounded to float16 included,
    oracle/networks.py:synthesis_block_fp16).  Tolerance of the fp16 comparison: the two sides agree on every rounding point
    of the activations but not on the float16 rounding of the modulated weights (2^-11 relative per weight, averaged over
    K = 9 x 256 terms) nor on accumulation order, each of which can move an activation across a float16 rounding boundary:
    8 float16 ulps (2^-11 relative) of the image's magnitude; fp32 mode: the networks' 3e-4."""
    from next3d_amd import generator, layers, networks
    layers.set_precision('bf16x3')
    P = spec.synthetic_state_dict(0, only=lambda n: n.startswith('superresolution'))
    Pd = {k: v.to(dev) for k, v in P.items()}
    ws, x = _g((N, 14, 512), 6), _g((N, 32, R, R), 7)
    rgb = x[:, :3].contiguous()
    sr = networks.SuperRes8XDC(Pd, 'superresolution', conv_clamp=256)
    ref32 = ON.superresolution(P, 'superresolution', rgb, x, ws, force_fp32=True)
    _report('superresolution fp32', sr(rgb.to(dev), x.to(dev), ws.to(dev), generator._resize_aa), ref32, 3e-4)
    ref16 = ON.superresolution(P, 'superresolution', rgb, x, ws, force_fp32=False)
    y16 = sr(rgb.to(dev), x.to(dev), ws.to(dev), generator._resize_aa, fp16=True)
    print('fp16 vs fp32 reference paths differ by', float((ref16 - ref32).abs().max()))
    _report('superresolution fp16', y16, ref16, 8 * 2.0 ** -11)
    assert float((y16.cpu() - ref32).abs().max()) > 0            # the mode does something
