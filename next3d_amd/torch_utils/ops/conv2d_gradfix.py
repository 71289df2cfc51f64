"""conv2d / conv_transpose2d front-ends with the reference's names (torch_utils/ops/conv2d_gradfix.py:37-45), executed by libn3d.so
(forward only; `enabled` / `no_weight_gradients` exist for import compatibility and have no effect at inference).

This is what a pickled, UN-RELOADED reference network reaches (gen_samples_next3d.py:119,150-151: `--reload_modules False` is the
default): Conv2dLayer and the non-fused modulated convolution call with groups = 1, the FUSED modulated convolution
(training_avatar_texture/networks_stylegan2.py:82-88, the inference default) with groups = batch and per-sample weights
[N*O, I, k, k].  Either way ONE launch runs the whole batch:
  * float32: the split-bf16 kernels (`layers.PRECISION == 'bf16x3'`, the same switch as the model boundary) or the fp32-MFMA kernels;
    the weights are re-tiled by one launch (n3d_conv2d_prep_weight_grouped: all groups, coalesced both ways) and reach the kernels
    through n3d_conv2d_desc.wt_batch_stride; large 3x3 stride-1 layers convert their input once to the split8 layout and run on the
    LDS-DMA kernel, as at the model boundary;
  * float16 (both operands: what the reference's fp16 blocks pass, `w.to(x.dtype)`): the f16 matrix-core kernels (n3d_conv2d_f16) on
    h8 tensors — float16 x float16 products are exact in float32, the accumulation is float32 and the result is rounded once: the
    arithmetic of ATen's half convolution.  Shapes those kernels do not take (1x1, O % 64 != 0) are converted on the device
    (n3d_cast), multiplied on the split-bf16 kernels (exact for float16 operands) and stored as float16.
Prepared weights of groups == 1 calls are cached per weight TENSOR OBJECT (weak reference + version counter), so a persistent
parameter is re-tiled once; temporaries (Conv2dLayer's `self.weight * self.weight_gain`) are re-tiled per call — a cache keyed on
(data_ptr, _version) would hand layer B the tiles of layer A's freed temporary of the same shape.
Anything else raises RuntimeError (the analogue of ATen's dtype check) — a non-float32 pointer never reaches a float32 kernel."""
import contextlib
import ctypes

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


# The library's kernel-selection queries are pure functions of the shape for the shipped library: memoised (a generator forward asks ~15 of them per
# layer call, each a ctypes round trip; at batch 1 the eager call is bound by the host).  Not with N3D_LIB set: tuning builds re-read their switches.
import os as _os
_MEMO_OK = 'N3D_LIB' not in _os.environ


def _memo(fn):
    cache = {}

    def wrapped(*args):
        if not _MEMO_OK:
            return fn(*args)
        r = cache.get(args)
        if r is None:
            r = cache[args] = fn(*args)
        return r
    wrapped.__name__, wrapped.__doc__ = fn.__name__, fn.__doc__
    return wrapped


@_memo
def split8_ksplit(n, i, o, h, w):
    """n3d_conv2d_split8_ksplit: the split-K factor the pre-split stride-1 kernel uses for this shape (0 = not its layer)."""
    return int(_lib.lib().n3d_conv2d_split8_ksplit(n, i, o, h, w))


@_memo
def sk_eligible(n, i, o, h, w):
    """n3d_conv2d_sk_eligible: the one-launch few-pixel kernel takes this stride-1 3x3 layer."""
    return bool(_lib.lib().n3d_conv2d_sk_eligible(n, i, o, h, w))


@_memo
def sk_s2_eligible(n, i, o, h, w):
    """n3d_conv2d_sk_s2_eligible: the one-launch few-pixel kernel takes this STRIDE-2 3x3 layer (h x w = the input image)."""
    return bool(_lib.lib().n3d_conv2d_sk_s2_eligible(n, i, o, h, w))


@_memo
def sk_workspace(n, i, o, h, w, mode):
    """n3d_conv2d_sk_workspace: (floats of slab workspace, arrival counters) the few-pixel 3x3 kernels use for this layer when the descriptor carries
    `tickets` — K sliced over workgroups, reduced by the last arriver inside the launch; (0, 0) = not their layer / no slicing."""
    need = ctypes.c_int(0)
    fl = int(_lib.lib().n3d_conv2d_sk_workspace(n, i, o, h, w, mode, ctypes.byref(need)))
    return fl, need.value


@_memo
def _bf16x3_blocks(n, o, h, w, mode):
    return int(_lib.lib().n3d_conv2d_bf16x3_blocks(n, o, h, w, mode))


@_memo
def up_sk_eligible(n, i, o, h, w):
    """Does the one-launch few-position kernel take this transposed 3x3 layer (float32 NCHW in / out; n3d_conv2d_up_sk_eligible)?"""
    return bool(_lib.lib().n3d_conv2d_up_sk_eligible(n, i, o, h, w))


@_memo
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


PS_TICKETS = False   # hand the stream's arrival counters to the pre-split 3x3 kernels too (tuning builds: N3D_PS_PERSIST=2, tools/ps_persist_ab.py)
SK_SEAM = True       # few-pixel layers: K sliced over workgroups, reduced inside the launch (module constant; tools flip it in-process for A/B runs)
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
    blocks = _bf16x3_blocks(n, o, h, w, mode)
    # transposed: 8-wave workgroups, one per CU -> split only until ~2/3 of the CUs have one (measured: 160 blocks are
    # faster unsplit); stride-1: smaller 4-wave workgroups, several per CU
    want = 512 if mode == 0 else 160       # modes 1 / 2: 8-wave workgroups, one per CU
    ks = 1
    while blocks * ks < want and (i // (ks * 2)) >= 64 and ks < KSPLIT_MAX:
        ks *= 2
    return ks


class _Dense:
    """Shape / stride bookkeeping of a dense float32 [N,C,H,W] tensor that does not exist as one (a split8 / c8 operand, an output nobody writes):
    what conv_launch reads off `x` and `y` — a `torch.empty(..., device='meta')` costs the host ~4 us per layer for the same four numbers."""
    __slots__ = ('shape', '_st')
    dtype = torch.float32

    def __init__(self, n, c, h, w):
        self.shape = (n, c, h, w)
        self._st = (c * h * w, h * w, w, 1)

    def stride(self, k=None):
        return self._st if k is None else self._st[k]


def conv_launch(x, wt, ksize, mode, out_channels, out=None, style=None, epilogue=None, ksplit=None, bf16x3=False, row_pitch=False,
                out_c8=False, out_split8=False, side_style=None, _wt_batch_stride=0, _wt_flat=False, rgb=None):
    """x [N,I,H,W] (any batch stride, dense planes), wt prepared weights [k*k,I,OP] (or the split-bf16 tiles when
    bf16x3=True) -> y [N,out_channels,OH,OW].  row_pitch=True returns y as the [..., :OW] view of a buffer whose rows are
    padded to a multiple of 4 floats (16-byte-aligned rows for the odd-width transposed-conv output; upfirdn2d accepts it).
    `out` may itself be such a view.  out_c8=True (un-split transposed split-bf16 layer, O % 64 == 0, demodulation-only epilogue):
    the result is a `_lib.C8` (channel-interleaved float32) for upfirdn2d._fir4_split8.  out_split8=True (1x1 split-bf16 layer,
    O % 32 == 0): the result is a `_lib.Split8` for a following pre-split 3x3 layer without modulation.  side_style [N,I] (1x1
    split-bf16 layer, O <= 128): returns (y, `_lib.Split8` of x * side_style) — n3d_conv2d_desc.side_split8.
    _wt_batch_stride (bytes; `wt` is then the flat per-sample tensor of prep_weight_grouped, _wt_flat=True): per-sample weights,
    n3d_conv2d_desc.wt_batch_stride — the operator boundary's grouped calls.
    rgb = (weight [C,O] float32, styles [N,O]) (split8 input, mode 0, no split-K, C <= 32: up to 4 colours on the VALU, more on the matrix cores): the toRGB layer that is this layer's ONLY reader is
    evaluated in the epilogue (n3d_conv2d_desc.rgb_*): returns the partial colour images [N, ceil(O/64), C, H, W] for rgb_combine; the feature
    map itself is not written; with side_style [N,O] as well: returns (partial, `_lib.Split8` of the layer's output * side_style) — the operand
    image of the layer's second reader (the next block's transposed convolution), exactly split8_from_nchw(y, side_style)."""
    split8 = isinstance(x, _lib.Split8)
    if split8:      # pre-split activations (already modulated): the LDS-DMA kernel; x.data is the flat bf16 storage
        if not (bf16x3 and ksize == 3 and mode in (0, 1, 2) and style is None):
            raise RuntimeError('conv2d: a split8 input goes to the 3x3 split-bf16 kernels (stride 1, stride 2, transposed), without a style')
        n, i, h, w = x.shape
        xs = x
        x = _Dense(n, i, h, w)                                                 # shape / stride bookkeeping only
        if mode == 0:                                                          # the library's own split-K factor for this shape (few tiles, deep K)
            ksplit = max(1, split8_ksplit(n, i, out_channels, h, w))
        elif mode == 2:
            ksplit = 1                                                         # (the stride-2 kernel keeps the caller's split-K for its small grids)
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
        assert wt.dtype == torch.bfloat16 and (_wt_flat or tuple(wt.shape) == (ksize * ksize, i // 16, 2, 2, (o + 63) // 64 * 64, 8))
        assert (ksize == 3 and mode in (0, 1, 2)) or (ksize == 1 and mode == 0)
    elif not _wt_flat:
        assert wt.shape[0] == ksize * ksize and wt.shape[1] == i and wt.shape[2] == (o + 3) // 4 * 4, (tuple(wt.shape), ksize, i, o)
    if _wt_flat:
        per = ksize * ksize * i * ((o + 63) // 64 * 64 if bf16x3 else (o + 3) // 4 * 4) * 4
        assert wt.numel() * wt.element_size() == per * (n if _wt_batch_stride else 1) and _wt_batch_stride in (0, per), (wt.numel(), per, n, _wt_batch_stride)
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
        y = _Dense(n, o, oh, ow)
    elif out_split8:
        if not (bf16x3 and ksize == 1 and out is None and o % 32 == 0 and out_dtype == torch.float32 and not split8):
            raise RuntimeError('conv2d: the split8 output is written by the 1x1 split-bf16 kernel (O % 32 == 0)')
        s8 = _lib.Split8(n, o, oh, ow, wt.device)
        y = _Dense(n, o, oh, ow)
    elif rgb is not None:
        y = _Dense(n, o, oh, ow)                                                # not written (checked below)
    elif out is not None:
        y = out
    elif row_pitch:
        y = torch.empty([n, o, oh, (ow + 3) // 4 * 4], dtype=torch.float32, device=wt.device)[..., :ow]
    else:
        y = torch.empty([n, o, oh, ow], dtype=torch.float32, device=wt.device)
    assert tuple(y.shape) == (n, o, oh, ow) and y.stride(3) == 1 and y.stride(2) >= ow and y.stride(1) == oh * y.stride(2)
    gh, gw = (h + 1, w + 1) if mode == 2 else (oh, ow)
    sk_layer = False
    # (the few-pixel launchers decline a layer whose epilogue rounds to float16 or up-samples its residual, and every side output: mirrored here so that the split-K
    #  choice below and the library's dispatch cannot diverge — ADVICE r5)
    sk_epi_ok = (epilogue is None or (not epilogue.round_f16 and not epilogue.residual_up_filter)) and side_style is None and s8 is None
    if not sk_epi_ok:
        pass
    elif bf16x3 and ksize == 3 and mode == 0 and not split8 and c8 is None and not _wt_batch_stride and sk_eligible(n, i, o, h, w):
        ksplit, sk_layer = 1, True                       # the few-pixel kernel splits K inside its workgroups: no split-K reduce launch
    elif bf16x3 and ksize == 3 and mode == 2 and not split8 and c8 is None and not _wt_batch_stride and out_dtype == torch.float32 and (epilogue is None or not epilogue.residual) and up_sk_eligible(n, i, o, h, w):
        ksplit, sk_layer = 1, True                       # ... and so does its transposed twin (few-position up-sampling layers)
    elif bf16x3 and ksize == 3 and mode == 1 and not split8 and not pitched_in and not _wt_batch_stride and out_dtype == torch.float32 and sk_s2_eligible(n, i, o, h, w):
        ksplit, sk_layer = 1, True                       # ... and the few-pixel stride-2 layers
    if ksplit is None:
        ksplit = (1 if ksize == 1 else pick_ksplit_bf16x3(n, i, o, h, w, mode)) if bf16x3 else pick_ksplit(n, i, o, gh, gw, ksize, mode)
    ws = torch.empty([ksplit * n * o * oh * ow], dtype=torch.float32, device=wt.device) if ksplit > 1 else None
    tk = None
    if SK_SEAM and sk_layer:                             # the few-pixel kernels slice K over the chip when given slabs + this stream's arrival counters
        sk_floats, sk_need = sk_workspace(n, i, o, h, w, mode)
        if sk_floats and sk_need <= _lib.TICKET_COUNT and sk_floats <= _lib.SLAB_FLOATS:
            tk = _lib.seam_pool()
    d = _lib.Conv2dDesc()
    d.x, d.wt, d.style, d.y, d.workspace = _lib.ptr(xs.data if split8 else x), _lib.ptr(wt), _lib.ptr(style), (None if rgb is not None else _lib.ptr(c8.data if c8 else (s8.data if s8 else y))), (tk.slabs_ptr if tk is not None else _lib.ptr(ws))
    d.x_layout, d.y_layout = (1 if split8 else 0), (2 if c8 else (1 if s8 else 0))
    d.N, d.I, d.O, d.H, d.W = n, i, o, h, w
    d.ksize, d.mode, d.ksplit = ksize, mode, ksplit
    d.x_batch_stride, d.y_batch_stride = x.stride(0), y.stride(0)
    d.style_stride = style.stride(0) if style is not None else 0
    d.y_row_stride = y.stride(2)                         # c8: pitch in pixels (= OW, dense), batch stride O * OH * OW floats
    d.x_row_stride = x.stride(2)
    d.epi = epilogue if epilogue is not None else _lib.make_epilogue()
    d.wt_batch_stride = int(_wt_batch_stride)
    if tk is None and PS_TICKETS and split8:              # tools: the dynamic-queue persistent kernels of tuning builds draw their tiles through the stream's counters
        tk = _lib.seam_pool()
        d.tickets, d.ticket_count = tk.tickets_ptr, _lib.TICKET_COUNT
    elif tk is not None:
        d.tickets, d.ticket_count = tk.tickets_ptr, _lib.TICKET_COUNT
    partial = None
    if rgb is not None:
        rw, rs = rgb
        if not (split8 and bf16x3 and ksize == 3 and mode == 0 and ksplit == 1 and out is None and not row_pitch and c8 is None and s8 is None and rw.dtype == rs.dtype == torch.float32 and
                rw.is_contiguous() and tuple(rw.shape[1:]) == (o,) and rw.shape[0] <= 32 and tuple(rs.shape) == (n, o) and rs.stride(1) == 1):
            raise RuntimeError('conv2d: the fused toRGB is an option of the pre-split 3x3 stride-1 kernel without split-K (weights [C<=32, O], styles [N, O])')
        _lib.require_device(rw, rs)
        partial = torch.empty([n, (o + 63) // 64, rw.shape[0], h, w], dtype=torch.float32, device=wt.device)
        d.rgb_weight, d.rgb_style, d.rgb_partial, d.rgb_channels, d.rgb_style_stride = _lib.ptr(rw), _lib.ptr(rs), _lib.ptr(partial), rw.shape[0], rs.stride(0)
    side = None
    if side_style is not None and rgb is not None:        # the fused-toRGB layer's own OUTPUT times the next layer's styles, as split8 (its second reader)
        if not (o % 8 == 0 and side_style.dtype == torch.float32 and side_style.stride(1) == 1 and tuple(side_style.shape) == (n, o)):
            raise RuntimeError('conv2d: the split8 side output of a fused-toRGB layer needs O % 8 == 0 and float32 styles [N, O]')
        _lib.require_device(side_style)
        side = _lib.Split8(n, o, h, w, wt.device)
        d.side_split8, d.side_style, d.side_style_stride = _lib.ptr(side.data), _lib.ptr(side_style), side_style.stride(0)
    elif side_style is not None:
        if not (bf16x3 and ksize == 1 and not split8 and s8 is None and o <= 128 and i % 32 == 0 and out_dtype == torch.float32 and
                side_style.dtype == torch.float32 and side_style.stride(1) == 1 and tuple(side_style.shape) == (n, i)):
            raise RuntimeError('conv2d: the split8 side output is written by the 1x1 split-bf16 kernel (O <= 128, I % 32 == 0, float32 styles [N,I])')
        _lib.require_device(side_style)
        side = _lib.Split8(n, i, h, w, wt.device)
        d.side_split8, d.side_style, d.side_style_stride = _lib.ptr(side.data), _lib.ptr(side_style), side_style.stride(0)
    fn = _lib.lib().n3d_conv2d_bf16x3 if bf16x3 else _lib.lib().n3d_conv2d
    _lib.check(fn(d, _lib.stream()))
    if partial is not None:
        return partial if side is None else (partial, side)
    if side is not None:
        return (y if out_dtype == torch.float32 else _lib.cast(y, out_dtype)), side
    if c8 is not None or s8 is not None:
        return c8 if c8 is not None else s8
    return y if out_dtype == torch.float32 else _lib.cast(y, out_dtype)


def rgb_combine(partial, epilogue):
    """The second half of conv_launch(rgb=...): [N, M, C, H, W] partial colours -> epilogue(sum over M) [N, C, H, W] (n3d_rgb_combine)."""
    n, m, c, h, w = partial.shape
    y = torch.empty([n, c, h, w], dtype=torch.float32, device=partial.device)
    _lib.check(_lib.lib().n3d_rgb_combine(_lib.ptr(partial), _lib.ptr(y), n, m, c, h, w, epilogue, _lib.stream()))
    return y


# ------------------------------------------------------------------------------------------------------------------------------
# Operator boundary (B1): F.conv2d / F.conv_transpose2d with shared or per-group weights, one launch for the whole batch.

_PREP_CACHE = {}           # id(weight tensor) -> {(kind, transposed, groups): (weakref, version, data_ptr, prepared)}


_layers = None


def _precision():
    global _layers
    if _layers is None:
        from ... import layers
        _layers = layers
    return _layers.PRECISION


def prep_weight_grouped(weight, groups, transposed, kind):
    """weight [G*O, I, k, k] (F.conv2d) or [G*I, O, k, k] (F.conv_transpose2d), float32 or float16, -> the per-group prepared
    weights of `kind` (0 float32 K-major, 1 split-bf16 tiles, 2 float16 tiles: include/n3d.h n3d_conv2d_prep_weight_grouped) as one
    flat tensor + the byte stride between groups.  ONE launch; the transposition of F.conv_transpose2d's layout is done by the
    kernel's strides, not by a torch copy."""
    _lib.require_device(weight)
    if weight.dtype not in (torch.float32, torch.float16):
        raise RuntimeError(f'conv2d weight: float32 or float16 expected, got {weight.dtype}')
    w = weight if weight.is_contiguous() else weight.contiguous()
    a, b, kh, kw = w.shape
    if kh != kw or kh not in (1, 3):
        raise RuntimeError(f'conv2d: kernel {kh}x{kw} unsupported (1x1 or 3x3)')
    kk = kh * kw
    if transposed:
        i, o = a // groups, b
        sg, so, si = i * o * kk, kk, o * kk
    else:
        o, i = a // groups, b
        sg, so, si = o * i * kk, i * kk, kk
    if kind == 0:
        per = kk * i * ((o + 3) // 4 * 4) * 4
        out = torch.empty(groups * per // 4, dtype=torch.float32, device=w.device)
    elif kind == 1:
        per = kk * i * ((o + 63) // 64 * 64) * 4
        out = torch.empty(groups * per // 2, dtype=torch.bfloat16, device=w.device)
    else:
        per = kk * i * o * 2
        out = torch.empty(groups * per // 2, dtype=torch.float16, device=w.device)
    _lib.check(_lib.lib().n3d_conv2d_prep_weight_grouped(_lib.ptr(w), 0 if w.dtype == torch.float32 else 1, _lib.ptr(out), kind, groups, o, i, kh,
                                                         sg, so, si, _lib.stream()))
    return out, per, o, i


def _tensor_version(t):
    """`t._version`, or None where autograd keeps no version counter (inference tensors: everything created under
    `torch.inference_mode()`, e.g. Conv2dLayer's `self.weight * self.weight_gain` temporary) — such a tensor is never cached."""
    if t.is_inference():
        return None
    try:
        return t._version
    except RuntimeError:
        return None


def clear_prep_cache():
    """Forget every cached prepared weight.  The cache follows the tensor OBJECT and its version counter; an update that bypasses the
    counter (`p.data.copy_(...)`, `p.data.mul_(...)`: `.data` detaches the counter) is invisible to it — call this after such an update
    (generator.refresh() / load_state_dict() / .to() do)."""
    _PREP_CACHE.clear()


def _prepared(weight, groups, transposed, kind):
    """prep_weight_grouped with the per-tensor-object cache for groups == 1 (module docstring)."""
    if groups != 1:
        return prep_weight_grouped(weight, groups, transposed, kind)
    version = _tensor_version(weight)
    if version is None:                                  # no version counter: nothing a later call could be validated against
        return prep_weight_grouped(weight, 1, transposed, kind)
    import weakref
    key, sub = id(weight), (kind, transposed)
    ent = _PREP_CACHE.get(key)
    if ent is not None:
        hit = ent.get(sub)
        if hit is not None and hit[0]() is weight and hit[1] == version and hit[2] == weight.data_ptr():
            return hit[3]
        if not any(h[0]() is weight for h in ent.values()):
            ent.clear()                                  # the id was recycled by another tensor
    res = prep_weight_grouped(weight, 1, transposed, kind)
    if ent is None:
        ent = _PREP_CACHE[key] = {}
        weakref.finalize(weight, lambda k=key: _PREP_CACHE.pop(k, None) if all(h[0]() is None for h in _PREP_CACHE.get(k, {}).values()) else None)
    ent[sub] = (weakref.ref(weight), version, weight.data_ptr(), res)
    return res


_EPI0 = None


def _launch_prepared(x, wt, wbs, kind, ksize, mode, n, i, o, h, w, row_pitch=False):
    """x [N,I,H,W] float32 (dense) or a `_lib.Split8`, prepared weights of `kind` (0 float32 K-major / 1 split-bf16 tiles) with byte
    stride `wbs` between samples (0 = shared) -> y [N,O,OH,OW] float32 (row_pitch: rows padded to 16 bytes, the [..., :OW] view).
    The operator boundary's launcher: no epilogue, no style — a lean twin of conv_launch (one descriptor, one ctypes call; this
    runs ~100 times per generator forward of an un-reloaded pickle, where the host, not the GPU, sets the pace of the small layers)."""
    global _EPI0
    if _EPI0 is None:
        _EPI0 = _lib.make_epilogue()
    split8 = isinstance(x, _lib.Split8)
    oh, ow = out_shape(h, w, mode)
    dev = wt.device
    if row_pitch:
        y = torch.empty([n, o, oh, (ow + 3) // 4 * 4], dtype=torch.float32, device=dev)[..., :ow]
    else:
        y = torch.empty([n, o, oh, ow], dtype=torch.float32, device=dev)
    bf16x3 = kind == 1
    ksplit = 1
    if split8:
        ksplit = max(1, split8_ksplit(n, i, o, h, w)) if mode == 0 else 1
    else:
        if bf16x3:
            ksplit = 1 if ksize == 1 else pick_ksplit_bf16x3(n, i, o, h, w, mode)
        else:
            gh, gw = (h + 1, w + 1) if mode == 2 else (oh, ow)
            ksplit = pick_ksplit(n, i, o, gh, gw, ksize, mode)
    ws = torch.empty([ksplit * n * o * oh * ow], dtype=torch.float32, device=dev) if ksplit > 1 else None
    d = _lib.Conv2dDesc()
    d.x, d.wt, d.y, d.workspace = _lib.ptr(x.data if split8 else x), _lib.ptr(wt), _lib.ptr(y), _lib.ptr(ws)
    d.x_layout = 1 if split8 else 0
    d.N, d.I, d.O, d.H, d.W = n, i, o, h, w
    d.ksize, d.mode, d.ksplit = ksize, mode, ksplit
    d.x_batch_stride, d.y_batch_stride = i * h * w, y.stride(0)
    d.y_row_stride, d.x_row_stride = y.stride(2), w
    d.epi = _EPI0
    d.wt_batch_stride = wbs
    L = _lib.lib()
    _lib.check((L.n3d_conv2d_bf16x3 if bf16x3 else L.n3d_conv2d)(d, _lib.stream()))
    return y


def _conv_f16_native(xv, weight, groups, transposed, mode, o, i):
    """float16 x float16 3x3 convolution on the f16 matrix cores: xv [N,I,H,W] float16 -> [N,O,OH,OW] float16 (module docstring)."""
    n, _, h, w = xv.shape
    xv = xv if xv.is_contiguous() else xv.contiguous()
    w16, per, _, _ = prep_weight_grouped(weight, groups, transposed, 2)      # (the f16 kernels take one weight tile set per sample: groups == N)
    xh = _lib.H8(n, i, h, w, xv.device)
    _lib.check(_lib.lib().n3d_cast_h8_ex(_lib.ptr(xv), _lib.ptr(xh.data), n, i, h * w, 0, 1, 1, _lib.stream()))
    oh, ow = out_shape(h, w, mode)
    yh = _lib.H8(n, o, oh, ow, xv.device)
    d = _lib.Conv2dDesc()
    d.x, d.wt, d.style, d.y, d.workspace = _lib.ptr(xh.data), _lib.ptr(w16), None, _lib.ptr(yh.data), None
    d.N, d.I, d.O, d.H, d.W = n, i, o, h, w
    d.ksize, d.mode, d.ksplit = 3, mode, 1
    d.x_layout = d.y_layout = 3
    d.epi = _lib.make_epilogue()
    _lib.check(_lib.lib().n3d_conv2d_f16(d, _lib.stream()))
    y = torch.empty(n, o, oh, ow, dtype=torch.float16, device=xv.device)
    _lib.check(_lib.lib().n3d_cast_h8_ex(_lib.ptr(yh.data), _lib.ptr(y), n, o, oh * ow, 0, 0, 1, _lib.stream()))
    return y


def _conv1x1_f16_native(xv, weight, o, i):
    """float16 1x1 convolution with O <= 4 per-sample output channels (the toRGB layer of a reference fp16 block,
    networks_stylegan2.py:353-357 -> conv2d(groups = N) on [N*O, I, 1, 1] weights): n3d_torgb_h8 without bias / skip image — the
    reference's weight tensor already IS that kernel's [N][O][C] operand — instead of widening a [N, I, 512, 512] tensor to float32."""
    n, _, h, w = xv.shape
    xv = xv if xv.is_contiguous() else xv.contiguous()
    wv = weight if weight.is_contiguous() else weight.contiguous()
    xh = _lib.H8(n, i, h, w, xv.device)
    _lib.check(_lib.lib().n3d_cast_h8_ex(_lib.ptr(xv), _lib.ptr(xh.data), n, i, h * w, 0, 1, 1, _lib.stream()))
    y32 = torch.empty(n, o, h, w, dtype=torch.float32, device=xv.device)
    _lib.check(_lib.lib().n3d_torgb_h8(_lib.ptr(xh.data), _lib.ptr(wv), None, None, None, _lib.ptr(y32), n, i, o, h, w, -1.0, _lib.stream()))
    return _lib.cast(y32, torch.float16)             # (the kernel rounds the sum to float16 itself: the conversion is exact)


def _f16_native_ok(i, o, h, w, ksize, mode):
    return ksize == 3 and i % 16 == 0 and o % 64 == 0 and i * 9 <= 4608 and ((mode == 0 and h >= 16 and w >= 32) or (mode == 2 and h >= 4 and w >= 4))


def _b1_conv(x, weight, groups, ksize, mode, transposed, row_pitch=False):
    """The whole call in ONE launch: x [B, G*Ig, H, W], weight per F.conv2d / F.conv_transpose2d, -> [B, G*Og, OH, OW] in x's dtype
    (row_pitch: a float32 result may be the [..., :OW] view of rows padded to 16 bytes — for the FIR conv2d_resample runs next)."""
    if weight.dtype != x.dtype:
        raise RuntimeError(f'conv2d: input ({x.dtype}) and weight ({weight.dtype}) must have the same dtype')
    if x.dtype not in (torch.float32, torch.float16):
        raise RuntimeError(f'conv2d input: float32 or float16 expected, got {x.dtype}')
    b, c, h, w = x.shape
    if groups > 1 and b != 1:           # not a call the reference makes (the fused branch folds the batch: B == 1): row by row
        return torch.cat([_b1_conv(x[r:r + 1], weight, groups, ksize, mode, transposed, row_pitch) for r in range(b)], 0)
    if c % groups != 0 or weight.shape[0] % groups != 0:
        raise RuntimeError(f'conv2d: channels ({c}, weight {tuple(weight.shape)}) not divisible by groups ({groups})')
    ig = c // groups
    og = weight.shape[1] if transposed else weight.shape[0] // groups
    if (weight.shape[0] // groups if transposed else weight.shape[1]) != ig:
        raise RuntimeError(f'conv2d: weight {tuple(weight.shape)} does not match {c} input channels in {groups} groups')
    if not x.is_contiguous():
        x = x.contiguous()
    xv = x.reshape(groups, ig, h, w) if groups > 1 else x             # group g of the folded tensor = sample g
    n = xv.shape[0]
    if x.dtype == torch.float16 and groups == n and _f16_native_ok(ig, og, h, w, ksize, mode):
        y = _conv_f16_native(xv, weight, groups, transposed, mode, og, ig)
    elif x.dtype == torch.float16 and groups == n and ksize == 1 and og <= 4 and ig % 8 == 0 and 8 <= ig <= 512:
        y = _conv1x1_f16_native(xv, weight, og, ig)
    else:
        xf = _as_f32(xv, 'conv2d input')
        # kernel family: split-bf16 (exact for float16 operands; float32 operands to 2^-16) unless PRECISION == 'fp32' or the shape is not its
        kind = 1 if ((_precision() == 'bf16x3' or x.dtype == torch.float16) and bf16x3_eligible(ig, h, w, ksize, mode)) else 0
        wt, per, _, _ = _prepared(weight, groups, transposed, kind)
        wbs = per if groups > 1 else 0
        if kind == 1 and ksize == 3 and mode == 0 and split8_eligible(n, ig, og, h, w):
            # one conversion pass buys the LDS-DMA kernel (30 % faster on the >= 64 x 64 layers; at the MODEL boundary the producer's
            # epilogue writes this layout, here the operator's input is whatever torch tensor the reference code holds)
            xf = split8_from_nchw(xf)
        elif not xf.is_contiguous():
            xf = xf.contiguous()
        y = _launch_prepared(xf, wt, wbs, kind, ksize, mode, n, ig, og, h, w, row_pitch and x.dtype == torch.float32)
        if x.dtype == torch.float16:
            y = _lib.cast(y, torch.float16)
    return y.reshape(1, groups * og, *y.shape[2:]) if groups > 1 else y


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
    if mode == 1 and groups != 1:
        raise RuntimeError('conv2d: a grouped stride-2 convolution is outside the generator-forward path')
    return _b1_conv(input, weight, groups, k, mode, transposed=False)


def conv_transpose2d_conv_layout(input, weight, groups=1, row_pitch=False):
    """F.conv_transpose2d(input, W_t, stride=2, groups=groups) where W_t is the transposition conv2d_resample.py:117-125 builds from
    the CONVOLUTION-layout weight [G*Og, Ig, 3, 3] passed here — the transposition itself is skipped (module docstring of
    conv2d_resample.conv2d_resample's 'tconv' step)."""
    _lib.require_device(input, weight)
    if weight.shape[2] != 3 or weight.shape[3] != 3:
        raise RuntimeError('conv_transpose2d: only 3x3 / stride 2 / padding 0 is on the generator-forward path')
    return _b1_conv(input, weight, groups, 3, 2, transposed=False, row_pitch=row_pitch)


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
    """F.conv_transpose2d subset: 3x3, stride 2, padding 0 (the up-sampling layers, conv2d_resample.py:127)."""
    _lib.require_device(input, weight)
    stride = stride if isinstance(stride, int) else stride[0]
    pad = padding if isinstance(padding, int) else padding[0]
    if stride != 2 or pad != 0 or weight.shape[2] != 3 or output_padding != 0 or dilation != 1 or bias is not None:
        raise RuntimeError('conv_transpose2d: only 3x3 / stride 2 / padding 0 is on the generator-forward path')
    return _b1_conv(input, weight, groups, 3, 2, transposed=True)
