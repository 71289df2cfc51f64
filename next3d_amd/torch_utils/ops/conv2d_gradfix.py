"""conv2d / conv_transpose2d front-ends with the reference's names (torch_utils/ops/conv2d_gradfix.py:37-45),
executed by the fp32-MFMA implicit-GEMM kernel of libn3d.so (forward only; `enabled` / `no_weight_gradients`
exist for import compatibility and have no effect at inference).

dtypes: float32, or float16 for BOTH input and weight (what the reference's fp16 blocks pass, networks_stylegan2.py:84-88
`w.to(x.dtype)`): fp16 tensors are converted on the device (n3d_cast), multiplied with float32 accumulation and the result is
stored as float16 — fp16 x fp16 products are exact in fp32, so this is the arithmetic of an fp16 convolution with fp32
accumulation (cuDNN's default for half).  Anything else raises RuntimeError (the analogue of ATen's dtype check) — a
non-float32 pointer never reaches a float32 kernel."""
import contextlib

import os

import torch

from ... import _lib

enabled = False
weight_gradients_disabled = False


@contextlib.contextmanager
def no_weight_gradients(disable=True):
    yield


def _as_f32(t, what):
    """float32 view of a float32 / float16 DEVICE tensor (converted by n3d_cast); other dtypes raise."""
    if t.dtype not in (torch.float32, torch.float16):
        raise RuntimeError(f'{what}: float32 or float16 expected, got {t.dtype}')
    return _lib.cast(t, torch.float32)


def prep_weight(w, want_sq=False):
    """[O,I,k,k] -> K-major [k*k, I, O] (+ optional per-(o,i) sum of squares for demodulation)."""
    _lib.require_device(w)
    w = _as_f32(w, 'conv2d weight')
    o, i, kh, kw = w.shape
    if kh != kw or kh not in (1, 3):
        raise RuntimeError(f'conv2d: kernel {kh}x{kw} unsupported (1x1 or 3x3)')
    w = w.contiguous()
    wt = torch.empty([kh * kw, i, (o + 3) // 4 * 4], dtype=torch.float32, device=w.device)     # rows padded to 16 bytes
    wsq = torch.empty([o, i], dtype=torch.float32, device=w.device) if want_sq else None
    _lib.check(_lib.lib().n3d_conv2d_prep_weight(_lib.ptr(w), _lib.ptr(wt), _lib.ptr(wsq), o, i, kh, _lib.stream()))
    return (wt, wsq) if want_sq else wt


def prep_weight_bf16x3(w):
    """[O,I,k,k] fp32 (k = 3 or 1) -> split-bf16 K-major tiles for the bf16x3 kernels (see include/n3d.h)."""
    _lib.require_device(w)
    w = _as_f32(w, 'conv2d weight')
    o, i, kh, kw = w.shape
    if kh != kw or kh not in (1, 3) or i % 16 != 0:
        raise RuntimeError('prep_weight_bf16x3: needs a 3x3 or 1x1 kernel and I % 16 == 0')
    op64 = (o + 63) // 64 * 64
    wt16 = torch.empty([kh * kh, i // 16, 2, 2, op64, 8], dtype=torch.bfloat16, device=w.device)
    w = w.contiguous()
    _lib.check(_lib.lib().n3d_conv2d_prep_weight_bf16x3(_lib.ptr(w), _lib.ptr(wt16), o, i, kh, _lib.stream()))
    return wt16


def bf16x3_eligible(i, h, w, ksize, mode):
    """Layers the split-bf16 kernels cover: 3x3, I % 16 == 0, stride-1 or transposed stride-2, from 4x4 up (images
    narrower than a 32-pixel MFMA tile are flattened row-major over the tile's columns)."""
    if ksize == 1:          # 1x1 (toRGB / fromRGB / fusion): activations streamed straight into the MFMA fragments
        return mode == 0 and i % 32 == 0 and i <= 1024
    if mode == 1:           # stride 2: polyphase kernel, 16 x 32 output tiles (smaller outputs stay on the fp32 split-K path)
        return ksize == 3 and i % 16 == 0 and (w - 3) // 2 + 1 >= 32 and (h - 3) // 2 + 1 >= 16
    return ksize == 3 and i % 16 == 0 and mode in (0, 2) and w >= 4 and h >= 4


def split8_eligible(n, i, o, h, w):
    """True when the 3x3 stride-1 layer [n,i,h,w] -> o channels is taken by the pre-split kernel (its producer may then write
    the split8 layout): the library's own rule (n3d_conv2d_split8_eligible)."""
    return bool(_lib.lib().n3d_conv2d_split8_eligible(n, i, o, h, w))


def split8_from_nchw(x, scale=None):
    """float32 [N,C,H,W] (dense planes, any batch stride) -> `_lib.Split8`, every channel multiplied by scale [N,C] first (the
    consuming layer's styles) — n3d_split8_from_nchw.  For tensors with two consumers (a block's output feeds toRGB and the
    next block's up-sampling convolution, with different styles), whose producer therefore cannot write split8 itself."""
    _lib.require_device(x, scale)
    n, c, h, w = x.shape
    if x.dtype != torch.float32 or (scale is not None and (scale.dtype != torch.float32 or scale.stride(1) != 1)):
        raise RuntimeError('split8_from_nchw: float32 tensors expected')
    if x.stride()[1:] != (h * w, w, 1):
        x = x.contiguous()
    y = _lib.Split8(n, c, h, w, x.device)
    _lib.check(_lib.lib().n3d_split8_from_nchw(_lib.ptr(x), _lib.ptr(scale), _lib.ptr(y.data), n, c, h * w, x.stride(0),
                                               scale.stride(0) if scale is not None else 0, _lib.stream()))
    return y


KSPLIT_MAX = 64      # cap on the split-K factor of the split-bf16 3x3 kernels (module constant; tools sweep it in-process)


def out_shape(h, w, mode):
    if mode == 0:
        return h, w
    if mode == 1:
        return (h - 3) // 2 + 1, (w - 3) // 2 + 1
    return 2 * h + 1, 2 * w + 1


def pick_ksplit(n, i, o, gh, gw, ksize, mode=0):
    """Split the input channels over extra workgroups when the output grid alone cannot fill 256 CUs
    (low-resolution layers: K = 9*I is deep, the pixel grid is tiny)."""
    bm = 64 if mode == 2 else (128 if (ksize == 3 or o > 64) else 32)
    tw, th = (32, 4) if gw > 16 else (16, 8)
    blocks = -(-gw // tw) * -(-gh // th) * -(-o // bm) * n
    icb = 16 if mode == 2 else (8 if ksize == 3 else 32)
    ks = 1
    while blocks * ks < 256 and (i // (ks * 2)) >= 4 * icb:
        ks *= 2
    return ks


def pick_ksplit_bf16x3(n, i, o, h, w, mode=0):
    """Split-K factor for the split-bf16 kernels from the library's own tile plan (n3d_conv2d_bf16x3_blocks)."""
    blocks = _lib.lib().n3d_conv2d_bf16x3_blocks(n, o, h, w, mode)
    # transposed: 8-wave workgroups, one per CU -> split only until ~2/3 of the CUs have one (measured: 160 blocks are
    # faster unsplit); stride-1: smaller 4-wave workgroups, several per CU
    want = 512 if mode == 0 else 160       # modes 1 / 2: 8-wave workgroups, one per CU
    ks = 1
    while blocks * ks < want and (i // (ks * 2)) >= 64 and ks < KSPLIT_MAX:
        ks *= 2
    return ks


def conv_launch(x, wt, ksize, mode, out_channels, out=None, style=None, epilogue=None, ksplit=None, bf16x3=False, row_pitch=False,
                out_c8=False, out_split8=False, side_style=None):
    """x [N,I,H,W] (any batch stride, dense planes), wt prepared weights [k*k,I,OP] (or the split-bf16 tiles when
    bf16x3=True) -> y [N,out_channels,OH,OW].  row_pitch=True returns y as the [..., :OW] view of a buffer whose rows are
    padded to a multiple of 4 floats (16-byte-aligned rows for the odd-width transposed-conv output; upfirdn2d accepts it).
    `out` may itself be such a view.  out_c8=True (un-split transposed split-bf16 layer, O % 64 == 0, demodulation-only epilogue):
    the result is a `_lib.C8` (channel-interleaved float32) for upfirdn2d._fir4_split8.  out_split8=True (1x1 split-bf16 layer,
    O % 32 == 0): the result is a `_lib.Split8` for a following pre-split 3x3 layer without modulation.  side_style [N,I] (1x1
    split-bf16 layer, O <= 128): returns (y, `_lib.Split8` of x * side_style) — n3d_conv2d_desc.side_split8."""
    split8 = isinstance(x, _lib.Split8)
    if split8:      # pre-split activations (already modulated): the LDS-DMA kernel; x.data is the flat bf16 storage
        if not (bf16x3 and ksize == 3 and (mode in (0, 1) or (mode == 2 and out_c8)) and style is None):
            raise RuntimeError('conv2d: a split8 input goes to the 3x3 stride-1 / stride-2 (or transposed, c8 output) split-bf16 kernel, without a style')
        n, i, h, w = x.shape
        xs = x
        x = torch.empty([n, i, h, w], dtype=torch.float32, device='meta')      # shape / stride bookkeeping only
        if mode != 1:
            ksplit = 1                                                         # (the stride-2 kernel keeps split-K for its small grids)
    _lib.require_device(None if split8 else x, wt, style, out)
    n, i, h, w = x.shape
    o = out_channels
    out_dtype = x.dtype
    if x.dtype != torch.float32:            # fp16 activations: fp32 arithmetic, fp16 storage (module docstring)
        if out is not None:
            raise RuntimeError('conv2d: an `out` buffer needs a float32 input')
        x = _as_f32(x, 'conv2d input')
        row_pitch = False
    if style is not None and style.dtype != torch.float32:
        raise RuntimeError(f'conv2d: float32 styles expected, got {style.dtype}')
    if not bf16x3 and wt.dtype != torch.float32:
        raise RuntimeError(f'conv2d: prepared weights must be float32 (prep_weight), got {wt.dtype}')
    if out is not None and out.dtype != torch.float32:
        raise RuntimeError(f'conv2d: float32 output buffer expected, got {out.dtype}')
    if bf16x3:
        assert wt.dtype == torch.bfloat16 and tuple(wt.shape) == (ksize * ksize, i // 16, 2, 2, (o + 63) // 64 * 64, 8)
        assert (ksize == 3 and mode in (0, 1, 2)) or (ksize == 1 and mode == 0)
    else:
        assert wt.shape[0] == ksize * ksize and wt.shape[1] == i and wt.shape[2] == (o + 3) // 4 * 4, (tuple(wt.shape), ksize, i, o)
    pitched_in = bf16x3 and mode == 1 and ksize == 3 and x.stride(3) == 1 and x.stride(2) > w and x.stride(1) == h * x.stride(2)
    if not split8 and not pitched_in and x.stride()[1:] != (h * w, w, 1):
        x = x.contiguous()
    oh, ow = out_shape(h, w, mode)
    c8 = s8 = None
    if out_c8:
        if not (bf16x3 and mode == 2 and out is None and o % 64 == 0 and out_dtype == torch.float32):
            raise RuntimeError('conv2d: the channel-interleaved output is written by the transposed split-bf16 kernel (O % 64 == 0)')
        c8 = _lib.C8(n, o, oh, ow, wt.device)
        ksplit = 1
        y = torch.empty([n, o, oh, ow], dtype=torch.float32, device='meta')
    elif out_split8:
        if not (bf16x3 and ksize == 1 and out is None and o % 32 == 0 and out_dtype == torch.float32 and not split8):
            raise RuntimeError('conv2d: the split8 output is written by the 1x1 split-bf16 kernel (O % 32 == 0)')
        s8 = _lib.Split8(n, o, oh, ow, wt.device)
        y = torch.empty([n, o, oh, ow], dtype=torch.float32, device='meta')
    elif out is not None:
        y = out
    elif row_pitch:
        y = torch.empty([n, o, oh, (ow + 3) // 4 * 4], dtype=torch.float32, device=wt.device)[..., :ow]
    else:
        y = torch.empty([n, o, oh, ow], dtype=torch.float32, device=wt.device)
    assert tuple(y.shape) == (n, o, oh, ow) and y.stride(3) == 1 and y.stride(2) >= ow and y.stride(1) == oh * y.stride(2)
    gh, gw = (h + 1, w + 1) if mode == 2 else (oh, ow)
    if bf16x3 and ksize == 3 and mode == 0 and not split8 and c8 is None and _lib.lib().n3d_conv2d_sk_eligible(n, i, o, h, w):
        ksplit = 1                                       # the few-pixel kernel splits K inside its workgroups: no partial-sum workspace
    if ksplit is None:
        ksplit = (1 if ksize == 1 else pick_ksplit_bf16x3(n, i, o, h, w, mode)) if bf16x3 else pick_ksplit(n, i, o, gh, gw, ksize, mode)
    ws = torch.empty([ksplit * n * o * oh * ow], dtype=torch.float32, device=wt.device) if ksplit > 1 else None
    d = _lib.Conv2dDesc()
    d.x, d.wt, d.style, d.y, d.workspace = _lib.ptr(xs.data if split8 else x), _lib.ptr(wt), _lib.ptr(style), _lib.ptr(c8.data if c8 else (s8.data if s8 else y)), _lib.ptr(ws)
    d.x_layout, d.y_layout = (1 if split8 else 0), (2 if c8 else (1 if s8 else 0))
    d.N, d.I, d.O, d.H, d.W = n, i, o, h, w
    d.ksize, d.mode, d.ksplit = ksize, mode, ksplit
    d.x_batch_stride, d.y_batch_stride = x.stride(0), y.stride(0)
    d.style_stride = style.stride(0) if style is not None else 0
    d.y_row_stride = y.stride(2)                         # c8: pitch in pixels (= OW, dense), batch stride O * OH * OW floats
    d.x_row_stride = x.stride(2)
    d.epi = epilogue if epilogue is not None else _lib.make_epilogue()
    side = None
    if side_style is not None:
        if not (bf16x3 and ksize == 1 and not split8 and s8 is None and o <= 128 and i % 32 == 0 and out_dtype == torch.float32 and
                side_style.dtype == torch.float32 and side_style.stride(1) == 1 and tuple(side_style.shape) == (n, i)):
            raise RuntimeError('conv2d: the split8 side output is written by the 1x1 split-bf16 kernel (O <= 128, I % 32 == 0, float32 styles [N,I])')
        _lib.require_device(side_style)
        side = _lib.Split8(n, i, h, w, wt.device)
        d.side_split8, d.side_style, d.side_style_stride = _lib.ptr(side.data), _lib.ptr(side_style), side_style.stride(0)
    fn = _lib.lib().n3d_conv2d_bf16x3 if bf16x3 else _lib.lib().n3d_conv2d
    _lib.check(fn(d, _lib.stream()))
    if side is not None:
        return (y if out_dtype == torch.float32 else _lib.cast(y, out_dtype)), side
    if c8 is not None or s8 is not None:
        return c8 if c8 is not None else s8
    return y if out_dtype == torch.float32 else _lib.cast(y, out_dtype)


def _grouped(x, weight, groups, ksize, mode, transposed):
    """groups == batch-folded samples (modulated_conv2d's fused path reshapes x to [1, N*I, H, W])."""
    if weight.dtype != x.dtype:
        raise RuntimeError(f'conv2d: input ({x.dtype}) and weight ({weight.dtype}) must have the same dtype')
    n, ci, h, w = x.shape
    ig = ci // groups
    outs = []
    for g in range(groups):
        if transposed:      # weight [groups*I_g, O_g, k, k] -> per group [O_g, I_g, k, k]
            wg = weight[g * ig:(g + 1) * ig].transpose(0, 1)
        else:               # weight [groups*O_g, I_g, k, k]
            og = weight.shape[0] // groups
            wg = weight[g * og:(g + 1) * og]
        outs.append(conv_launch(x[:, g * ig:(g + 1) * ig], prep_weight(wg), ksize, mode, wg.shape[0]))
    return torch.cat(outs, dim=1)


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
    """F.conv2d subset used by the generator: k in {1,3}; (stride 1, padding k//2) or (stride 2, padding 0)."""
    _lib.require_device(input, weight)
    k = weight.shape[2]
    stride = stride if isinstance(stride, int) else stride[0]
    pad = padding if isinstance(padding, int) else padding[0]
    if dilation != 1 or bias is not None:
        raise RuntimeError('conv2d: dilation / bias are not supported by the n3d kernel')
    if stride == 1 and pad == k // 2:
        mode = 0
    elif stride == 2 and pad == 0 and k == 3:
        mode = 1
    else:
        raise RuntimeError(f'conv2d: stride={stride} padding={pad} kernel={k} is outside the generator-forward path')
    if weight.dtype != input.dtype:
        raise RuntimeError(f'conv2d: input ({input.dtype}) and weight ({weight.dtype}) must have the same dtype')
    if groups == 1:
        return conv_launch(input, prep_weight(weight), k, mode, weight.shape[0])
    return _grouped(input, weight, groups, k, mode, transposed=False)


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    """F.conv_transpose2d subset: 3x3, stride 2, padding 0 (the up-sampling layers, conv2d_resample.py:127)."""
    _lib.require_device(input, weight)
    stride = stride if isinstance(stride, int) else stride[0]
    pad = padding if isinstance(padding, int) else padding[0]
    if stride != 2 or pad != 0 or weight.shape[2] != 3 or output_padding != 0 or dilation != 1 or bias is not None:
        raise RuntimeError('conv_transpose2d: only 3x3 / stride 2 / padding 0 is on the generator-forward path')
    if weight.dtype != input.dtype:
        raise RuntimeError(f'conv_transpose2d: input ({input.dtype}) and weight ({weight.dtype}) must have the same dtype')
    if groups == 1:
        return conv_launch(input, prep_weight(weight.transpose(0, 1)), 3, 2, weight.shape[1])
    return _grouped(input, weight, groups, 3, 2, transposed=True)
