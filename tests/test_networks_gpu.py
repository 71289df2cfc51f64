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
