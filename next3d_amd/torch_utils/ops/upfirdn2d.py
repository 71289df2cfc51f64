"""upfirdn2d family — same Python signatures as reference torch_utils/ops/upfirdn2d.py:72-389, forward only, HIP only."""
import numpy as np
import torch

from ... import _lib


def _parse_scaling(scaling):
    if isinstance(scaling, int):
        scaling = [scaling, scaling]
    assert isinstance(scaling, (list, tuple)) and all(isinstance(x, int) for x in scaling)
    sx, sy = scaling
    assert sx >= 1 and sy >= 1
    return sx, sy


def _parse_padding(padding):
    if isinstance(padding, int):
        padding = [padding, padding]
    assert isinstance(padding, (list, tuple)) and all(isinstance(x, int) for x in padding)
    if len(padding) == 2:
        px, py = padding
        padding = [px, px, py, py]
    px0, px1, py0, py1 = padding
    return px0, px1, py0, py1


def _get_filter_size(f):
    if f is None:
        return 1, 1
    assert isinstance(f, torch.Tensor) and f.ndim in [1, 2]
    return int(f.shape[-1]), int(f.shape[0])


def setup_filter(f, device=torch.device('cpu'), normalize=True, flip_filter=False, gain=1, separable=None):
    """FIR taps as a float32 tensor: 1-D (separable, >= 8 taps) or 2-D outer product."""
    if f is None:
        f = 1
    f = torch.as_tensor(f, dtype=torch.float32)
    assert f.ndim in [0, 1, 2] and f.numel() > 0
    if f.ndim == 0:
        f = f[np.newaxis]
    if separable is None:
        separable = (f.ndim == 1 and f.numel() >= 8)
    if f.ndim == 1 and not separable:
        f = f.ger(f)
    assert f.ndim == (1 if separable else 2)
    if normalize:
        f = f / f.sum()
    if flip_filter:
        f = f.flip(list(range(f.ndim)))
    f = f * (gain ** (f.ndim / 2))
    return f.to(device=device)


def _launch(x, f2d, up, down, padding, flip_filter, gain, epilogue=None, row_pitch=False):
    upx, upy = up
    downx, downy = down
    px0, px1, py0, py1 = padding
    n, c, h, w = x.shape
    fh, fw = f2d.shape
    ow = (w * upx + px0 + px1 - fw + downx) // downx
    oh = (h * upy + py0 + py1 - fh + downy) // downy
    if ow < 1 or oh < 1:
        raise RuntimeError('upfirdn2d: output would be empty')
    if row_pitch:           # rows padded to 16 bytes (odd widths): the [..., :ow] view of the wider buffer
        y = torch.empty([n, c, oh, (ow + 3) // 4 * 4], dtype=x.dtype, device=x.device)[..., :ow]
    else:
        y = torch.empty([n, c, oh, ow], dtype=x.dtype, device=x.device)
    _lib.check(_lib.lib().n3d_upfirdn2d_pitched(_lib.ptr(x), _lib.ptr(f2d), _lib.ptr(y), n, c, h, w, x.stride(2), y.stride(2), fh, fw, upx, upy,
                                                downx, downy, px0, px1, py0, py1, 1 if flip_filter else 0, float(gain),
                                                x.stride(0), y.stride(0), epilogue, _lib.stream()))
    return y


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, impl='cuda', _epilogue=None, _row_pitch=False):
    """Pad, upsample, filter, downsample a batch of 2-D images (see the reference docstring, upfirdn2d.py:120-160)."""
    assert isinstance(x, torch.Tensor) and impl in ['ref', 'cuda']
    if impl == 'ref':
        raise RuntimeError("impl='ref' is not part of the product: the CPU restatement is oracle/ops.py (tests only)")
    _lib.require_device(x, f)
    assert x.ndim == 4
    if x.dtype != torch.float32:
        raise RuntimeError('upfirdn2d: this build computes in float32 only')
    up, down, padding = _parse_scaling(up), _parse_scaling(down), _parse_padding(padding)
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
    assert f.dtype == torch.float32 and f.ndim in [1, 2]
    # dense planes, or rows with a pitch (a [..., :W] view of a wider buffer: conv_launch(..., row_pitch=True))
    x = x if (x.stride(3) == 1 and x.stride(2) >= x.shape[3] and x.stride(1) == x.shape[2] * x.stride(2)) else x.contiguous()
    if f.ndim == 2:
        return _launch(x, f.contiguous(), up, down, padding, flip_filter, gain, _epilogue, _row_pitch)
    # separable: horizontal pass then vertical pass, gain split as sqrt per pass (upfirdn2d.py:240-244)
    px0, px1, py0, py1 = padding
    g = float(gain) ** 0.5
    x = _launch(x, f.unsqueeze(0).contiguous(), (up[0], 1), (down[0], 1), (px0, px1, 0, 0), flip_filter, g)
    return _launch(x, f.unsqueeze(1).contiguous(), (1, up[1]), (1, down[1]), (0, 0, py0, py1), flip_filter, g, _epilogue)


def filter2d(x, f, padding=0, flip_filter=False, gain=1, impl='cuda'):
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [px0 + fw // 2, px1 + (fw - 1) // 2, py0 + fh // 2, py1 + (fh - 1) // 2]
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    upx, upy = _parse_scaling(up)
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [px0 + (fw + upx - 1) // 2, px1 + (fw - upx) // 2, py0 + (fh + upy - 1) // 2, py1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy, impl=impl)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    downx, downy = _parse_scaling(down)
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    p = [px0 + (fw - downx + 1) // 2, px1 + (fw - downx) // 2, py0 + (fh - downy + 1) // 2, py1 + (fh - downy) // 2]
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)
