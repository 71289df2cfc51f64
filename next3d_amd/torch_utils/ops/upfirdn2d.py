"""upfirdn2d family — the Python signatures of the reference's torch_utils/ops/upfirdn2d.py (:72 setup_filter, :120 upfirdn2d,
:279 filter2d, :315 upsample2d, :354 downsample2d), forward only, executed by libn3d.so (n3d_upfirdn2d_pitched).

float32 tensors run directly; float16 tensors (the reference's fp16 blocks, superresolution.py:210-217) are converted on the
device (n3d_cast), filtered with float32 accumulation and stored back as float16 — the same rounding points as the
reference's fp16 kernel (upfirdn2d.cu accumulates in float for half inputs, :36 `scalar_t`/`float` split).
"""
import torch

from ... import _lib


# ---------------------------------------------------------------------------------------------- argument normalisation
def _ints(value, count, what):
    """int or sequence of ints -> tuple of `count` ints (a shorter sequence of length count/2 is repeated per axis)."""
    seq = [value] * count if isinstance(value, int) else list(value)
    if not all(isinstance(v, int) for v in seq):
        raise AssertionError(f'{what} must be an int or a sequence of ints')
    if len(seq) * 2 == count:
        seq = [v for v in seq for _ in range(2)]
    if len(seq) != count:
        raise AssertionError(f'{what}: expected {count} values, got {len(seq)}')
    return tuple(seq)


def _parse_scaling(scaling):
    """-> (x factor, y factor), both >= 1."""
    sx, sy = _ints(scaling, 2, 'scaling')
    assert sx >= 1 and sy >= 1
    return sx, sy


def _parse_padding(padding):
    """-> (x before, x after, y before, y after); negative values crop."""
    return _ints(padding, 4, 'padding')


def _get_filter_size(f):
    """-> (taps along x, taps along y); no filter counts as a single tap."""
    if f is None:
        return 1, 1
    assert isinstance(f, torch.Tensor) and f.ndim in [1, 2]
    return int(f.shape[-1]), int(f.shape[0])


def setup_filter(f, device=torch.device('cpu'), normalize=True, flip_filter=False, gain=1, separable=None):
    """FIR taps for the functions below: a float32 [taps] (separable) or [fh, fw] tensor.  1-D input becomes an outer
    product unless it has 8 or more taps (or `separable` says otherwise); `gain` is split evenly over the two passes of a
    separable filter."""
    taps = torch.as_tensor(1 if f is None else f, dtype=torch.float32)
    if taps.ndim == 0:
        taps = taps.reshape(1)
    assert taps.ndim in [1, 2] and taps.numel() > 0
    if separable is None:
        separable = taps.ndim == 1 and taps.numel() >= 8
    if taps.ndim == 1 and not separable:
        taps = taps[:, None] * taps[None, :]
    assert taps.ndim == (1 if separable else 2)
    if normalize:
        taps = taps / taps.sum()
    if flip_filter:
        taps = taps.flip(list(range(taps.ndim)))
    return (taps * gain ** (taps.ndim / 2)).to(device=device)


# ---------------------------------------------------------------------------------------------- launch
def _launch(x, f2d, up, down, padding, flip_filter, gain, epilogue=None, row_pitch=False):
    """One n3d_upfirdn2d_pitched launch on float32 data; `up`, `down` = (x, y) pairs, `padding` = 4 ints, f2d = [fh, fw]."""
    (upx, upy), (downx, downy), (px0, px1, py0, py1) = up, down, padding
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


_FIR1D = {}                # id(filter tensor) -> (weak reference to it, its version counter, factor or None)
FIR_SEP = True             # separable evaluation of separable 4x4 filters (module constant; tools/fir_bench.py flips it in-process)


def fir_factor(fir):
    """4 device taps `a` with fir == outer(a, a), or None — decided once per filter TENSOR, inference tensors included (one host read; the networks call this
    at model preparation, networks.py, so that no forward — and no HIP-graph capture — ever does it).  The entry is tied to the
    tensor object by a weak reference and dropped with it: a later tensor at the same address never sees a stale factor.
    The model's filter is setup_filter([1,3,3,1]) = outer(v, v) with v = [1,3,3,1] / 8 (:96-116): the separable kernels
    (n3d_fir4_split8_sep, n3d_fir4_h8) evaluate the same float32 sum with half the multiply-adds."""
    import weakref
    from .conv2d_gradfix import _tensor_version
    key = id(fir)
    # inference tensors keep no version counter (a model built or loaded under torch.inference_mode()): their entry is tied to the tensor object and its storage
    # address instead — the filter is a registered buffer that nothing rewrites in place, and re-deciding it per call would cost a blocking device-to-host read
    # on every filtered layer (and is illegal inside a HIP-graph capture: ADVICE r5)
    version = _tensor_version(fir)
    if version is None:
        version = ('inference', fir.data_ptr())
    hit = _FIR1D.get(key)
    if hit is not None and hit[0]() is fir and hit[1] == version:
        return hit[2]
    f = fir.detach().to('cpu', torch.float64)
    a = None
    if tuple(f.shape) == (4, 4) and float(f.sum()) > 0:
        v = f.sum(1) / f.sum().sqrt()
        if torch.equal(torch.outer(v, v).to(torch.float32), f.to(torch.float32)):
            a = v.to(torch.float32).to(fir.device).contiguous()
    _FIR1D[key] = (weakref.ref(fir), version, a)
    weakref.finalize(fir, lambda k=key: _FIR1D.pop(k, None) if (_FIR1D.get(k) and _FIR1D[k][0]() is None) else None)
    return a


def _fir4_split8(x, f2d, gain, epilogue, out_scale):
    """The up-sampling layer's FIR (4x4 taps, padding 1 on every side) with its epilogue, writing the split8 layout for the 3x3
    convolution that follows (n3d_fir4_split8): x = the transposed convolution's output as a `_lib.C8` [N,C,H,W] ->
    `_lib.Split8` [N,C,H-1,W-1], values multiplied by out_scale [N,C] (the next layer's styles)."""
    assert isinstance(x, _lib.C8) and tuple(f2d.shape) == (4, 4) and out_scale.stride(1) == 1
    n, c, h, w = x.shape
    y = _lib.Split8(n, c, h - 1, w - 1, x.device)
    # (64 x 64 outputs: the 16-tap kernel's staged tile is the faster one, 21 vs 24 us — profiles/r03_fir_bench.txt; FIR_SEP = 'all'
    # forces the separable kernel there too: tests / tools)
    f1d = fir_factor(f2d) if (FIR_SEP and (epilogue is None or epilogue.act in (1, 3)) and (h > 100 or FIR_SEP == 'all')) else None
    if f1d is not None:
        _lib.check(_lib.lib().n3d_fir4_split8_sep(_lib.ptr(x.data), _lib.ptr(f1d), _lib.ptr(y.data), n, c, h, w, w, 0, 0, float(gain),
                                                  epilogue, _lib.ptr(out_scale), out_scale.stride(0), _lib.stream()))
        return y
    _lib.check(_lib.lib().n3d_fir4_split8(_lib.ptr(x.data), _lib.ptr(f2d), _lib.ptr(y.data), n, c, h, w, w, 0, 0, float(gain),
                                          epilogue, _lib.ptr(out_scale), out_scale.stride(0), _lib.stream()))
    return y


def _fir4_split8_nchw(x, f2d, pad, gain=1.0, epilogue=None, out_scale=None):
    """The 4x4 FIR with `pad` (1 / 2) zero pixels on every side from a float32 NCHW tensor straight into the split8 layout
    (n3d_fir4_split8_nchw): the filter in front of a stride-2 convolution (Conv2dLayer(down=2)), whose output is only ever read
    by that convolution."""
    assert tuple(f2d.shape) == (4, 4) and x.dtype == torch.float32 and x.ndim == 4
    if not _planes_ok(x):
        x = x.contiguous()
    n, c, h, w = x.shape
    y = _lib.Split8(n, c, h + 2 * pad - 3, w + 2 * pad - 3, x.device)
    f1d = fir_factor(f2d) if (FIR_SEP and (epilogue is None or epilogue.act in (1, 3))) else None
    if f1d is not None and h * x.stride(2) * 32 < 2 ** 31:
        _lib.check(_lib.lib().n3d_fir4_split8_nchw_sep(_lib.ptr(x), _lib.ptr(f1d), _lib.ptr(y.data), n, c, h, w, x.stride(2), x.stride(0), pad, 0,
                                                       float(gain), epilogue, _lib.ptr(out_scale), out_scale.stride(0) if out_scale is not None else 0,
                                                       _lib.stream()))
        return y
    _lib.check(_lib.lib().n3d_fir4_split8_nchw(_lib.ptr(x), _lib.ptr(f2d), _lib.ptr(y.data), n, c, h, w, x.stride(2), x.stride(0), pad, 0, float(gain),
                                               epilogue, _lib.ptr(out_scale), out_scale.stride(0) if out_scale is not None else 0, _lib.stream()))
    return y


def _planes_ok(x):
    """dense planes, or rows with a pitch (a [..., :W] view of a wider buffer: conv_launch(..., row_pitch=True))"""
    return x.stride(3) == 1 and x.stride(2) >= x.shape[3] and x.stride(1) == x.shape[2] * x.stride(2)


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, impl='cuda', _epilogue=None, _row_pitch=False):
    """Zero-stuff by `up`, pad / crop, correlate with the (flipped unless `flip_filter`) taps, keep every `down`-th sample."""
    assert isinstance(x, torch.Tensor) and impl in ['ref', 'cuda']
    if impl == 'ref':
        raise RuntimeError("impl='ref' is not part of the product: the CPU restatement is oracle/ops.py (tests only)")
    _lib.require_device(x, f)
    assert x.ndim == 4
    if x.dtype not in (torch.float32, torch.float16):
        raise RuntimeError(f'upfirdn2d: float32 or float16 input expected, got {x.dtype}')
    up, down, padding = _parse_scaling(up), _parse_scaling(down), _parse_padding(padding)
    if f is None:
        f = torch.ones([1, 1], dtype=torch.float32, device=x.device)
    if f.dtype != torch.float32 or f.ndim not in [1, 2]:
        raise RuntimeError('upfirdn2d: the filter must be a 1-D or 2-D float32 tensor (setup_filter)')
    out_dtype = x.dtype
    if x.dtype == torch.float16:                      # fp16 storage, fp32 arithmetic
        x = _lib.cast(x, torch.float32)
    elif not _planes_ok(x):
        x = x.contiguous()
    if f.ndim == 2:
        y = _launch(x, f.contiguous(), up, down, padding, flip_filter, gain, _epilogue, _row_pitch and out_dtype == torch.float32)
    else:   # separable: a row pass with the taps along x, then a column pass; each takes sqrt(gain)
        g = float(gain) ** 0.5
        rows = _launch(x, f[None, :].contiguous(), (up[0], 1), (down[0], 1), padding[:2] + (0, 0), flip_filter, g)
        y = _launch(rows, f[:, None].contiguous(), (1, up[1]), (1, down[1]), (0, 0) + padding[2:], flip_filter, g, _epilogue)
    return y if out_dtype == torch.float32 else _lib.cast(y, out_dtype)


# ---------------------------------------------------------------------------------------------- convenience wrappers
def _margins(taps, factor, upsampling):
    """Padding (before, after) that keeps an image aligned when it is filtered with `taps` taps around a x`factor` resampling:
    the filter's centre of mass sits on the sample grid of the LOW-rate side."""
    if upsampling:
        return (taps + factor - 1) // 2, (taps - factor) // 2
    return (taps - factor + 1) // 2, (taps - factor) // 2


def _resample(x, f, up, down, padding, flip_filter, gain, impl):
    ux, uy = _parse_scaling(up)
    dx, dy = _parse_scaling(down)
    px0, px1, py0, py1 = _parse_padding(padding)
    fw, fh = _get_filter_size(f)
    if (ux, uy) != (1, 1):
        mx, my = _margins(fw, ux, True), _margins(fh, uy, True)
    elif (dx, dy) != (1, 1):
        mx, my = _margins(fw, dx, False), _margins(fh, dy, False)
    else:                                   # plain filtering: 'same' size (odd taps centred, even taps lean to the front)
        mx, my = (fw // 2, (fw - 1) // 2), (fh // 2, (fh - 1) // 2)
    pad = [px0 + mx[0], px1 + mx[1], py0 + my[0], py1 + my[1]]
    return upfirdn2d(x, f, up=up, down=down, padding=pad, flip_filter=flip_filter, gain=gain * ux * uy, impl=impl)


def filter2d(x, f, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Filter without changing the size (plus `padding`)."""
    return _resample(x, f, 1, 1, padding, flip_filter, gain, impl)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Enlarge by an integer factor; the signal magnitude is kept (gain x factor^2 compensates the inserted zeros)."""
    return _resample(x, f, up, 1, padding, flip_filter, gain, impl)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Shrink by an integer factor."""
    return _resample(x, f, 1, down, padding, flip_filter, gain, impl)
