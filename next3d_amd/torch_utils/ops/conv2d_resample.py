"""conv2d_resample — the operator-boundary entry point with the signature of the reference's
torch_utils/ops/conv2d_resample.py:48 (`conv2d_resample(x, w, f, up, down, padding, groups, flip_weight, flip_filter)`),
forward only, every arithmetic step on libn3d.so (conv2d_gradfix -> n3d_conv2d, upfirdn2d -> n3d_upfirdn2d).

What it computes:   y = decimate_down( FIR_f( conv_w( FIR_f( zero_stuff_up(x) ) ) ) )   with `padding` counted on the
up-sampled image and the FIR margins chosen so that the image stays centred (upfirdn2d._margins).  How it is scheduled
depends on three facts, which `_plan` below turns into a short list of steps:

  * a 1x1 kernel commutes with resampling, so it runs on whichever side has fewer pixels;
  * zero-stuffing followed by a k x k convolution IS a stride-`up` transposed convolution (no multiplies by the stuffed
    zeros), so the up-sampling path is `conv_transpose2d` + one interpolation FIR;
  * low-pass filtering followed by decimation needs the filter at full rate, but the convolution after it can itself take
    the stride, so the down-sampling path is one FIR + a stride-`down` convolution.
"""
import torch

from . import conv2d_gradfix, upfirdn2d
from .upfirdn2d import _get_filter_size, _margins, _parse_padding


def _get_weight_shape(w):
    return [int(s) for s in w.shape]


def _conv2d_wrapper(x, w, stride=1, padding=0, groups=1, transpose=False, flip_weight=True):
    """One (transposed) convolution launch.  `flip_weight=True` means cross-correlation (what the kernels and F.conv2d
    compute); a true convolution is the correlation with the spatially mirrored kernel."""
    k_h, k_w = w.shape[2], w.shape[3]
    if not flip_weight and (k_h > 1 or k_w > 1):
        w = w.flip([2, 3])
    fn = conv2d_gradfix.conv_transpose2d if transpose else conv2d_gradfix.conv2d
    return fn(x, w, stride=stride, padding=padding, groups=groups)


def _transposed_weight(w, groups):
    """[G*Og, Ig, kh, kw] -> the [G*Ig, Og, kh, kw] layout conv_transpose2d expects (input and output channels swapped inside
    every group)."""
    o, ig, kh, kw = w.shape
    if groups == 1:
        return w.transpose(0, 1)
    og = o // groups
    return w.reshape(groups, og, ig, kh, kw).transpose(1, 2).reshape(groups * ig, og, kh, kw)


def _plan(ksize, up, down, pads):
    """-> list of steps, each ('fir', kwargs) | ('conv', kwargs) | ('tconv', kwargs), for the schedule described above.
    `pads` = [x0, x1, y0, y1] with the FIR margins already included; ksize = (kh, kw)."""
    kh, kw = ksize
    x0, x1, y0, y1 = pads
    pointwise = kh == 1 and kw == 1
    if up == 1 and down == 1:
        if x0 == x1 and y0 == y1 and x0 >= 0 and y0 >= 0:                 # the convolution's own zero padding does it
            return [('conv', dict(padding=[y0, x0]))]
        return [('fir', dict(filtered=False, padding=pads)), ('conv', {})]   # asymmetric / negative: pad or crop first
    if pointwise and down == 1:                                             # 1x1 before the interpolation (fewer pixels)
        return [('conv', {}), ('fir', dict(up=up, padding=pads, gain=up ** 2))]
    if pointwise and up == 1:                                               # 1x1 after the decimation (fewer pixels)
        return [('fir', dict(down=down, padding=pads)), ('conv', {})]
    if up == 1:                                                             # low-pass at full rate, strided convolution
        return [('fir', dict(padding=pads)), ('conv', dict(stride=down))]
    # up > 1.  A stride-`up` transposed convolution with zero padding returns (H-1)*up + k samples: compared with "stuff
    # zeros, pad, convolve" it already contains k-1 leading and k-up trailing border samples, so that much of the
    # requested padding is consumed.  What is left over is usually negative on both sides (a crop): the common part is
    # handed to conv_transpose2d's own `padding` (cropping inside the launch), the rest to the interpolation FIR.
    x0 -= kw - 1; x1 -= kw - up
    y0 -= kh - 1; y1 -= kh - up
    crop_x, crop_y = max(min(-x0, -x1), 0), max(min(-y0, -y1), 0)
    steps = [('tconv', dict(stride=up, padding=[crop_y, crop_x])),
             ('fir', dict(padding=[x0 + crop_x, x1 + crop_x, y0 + crop_y, y1 + crop_y], gain=up ** 2))]
    if down > 1:
        steps.append(('fir', dict(down=down)))
    return steps


def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True, flip_filter=False):
    """2-D convolution of x [N, G*Ig, H, W] with w [G*Og, Ig, kh, kw], optionally preceded by x`up` up-sampling and followed
    by x`down` down-sampling through the low-pass filter `f` (upfirdn2d.setup_filter)."""
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    assert isinstance(w, torch.Tensor) and w.ndim == 4 and w.dtype == x.dtype
    assert f is None or (isinstance(f, torch.Tensor) and f.ndim in [1, 2] and f.dtype == torch.float32)
    assert isinstance(up, int) and up >= 1 and isinstance(down, int) and down >= 1 and isinstance(groups, int) and groups >= 1
    _o, _ig, kh, kw = _get_weight_shape(w)
    fw, fh = _get_filter_size(f)
    pads = list(_parse_padding(padding))
    for factor, upsampling in ((up, True), (down, False)):                 # margins the FIR passes consume
        if factor > 1:
            mx, my = _margins(fw, factor, upsampling), _margins(fh, factor, upsampling)
            pads = [pads[0] + mx[0], pads[1] + mx[1], pads[2] + my[0], pads[3] + my[1]]

    for kind, kw_ in _plan((kh, kw), up, down, pads):
        if kind == 'fir':
            kw_ = dict(kw_)
            taps = f if kw_.pop('filtered', True) else None                 # filtered=False: pure pad / crop (single unit tap)
            x = upfirdn2d.upfirdn2d(x=x, f=taps, flip_filter=flip_filter, **kw_)
        elif kind == 'conv':
            x = _conv2d_wrapper(x=x, w=w, groups=groups, flip_weight=flip_weight, **kw_)
        else:   # 'tconv': conv_transpose2d scatters with the kernel as stored = a true convolution, hence the inverted flag
            if (kh, kw) == (3, 3) and kw_['stride'] == 2 and tuple(kw_['padding']) == (0, 0) and not flip_weight:
                # The generator's up-sampling layers.  F.conv_transpose2d wants [G*Ig, Og, k, k]; the kernels stream tiles indexed
                # (group, out, in, tap), which the one-launch re-tile reads from ANY strides — so the convolution-layout weight goes in
                # as it is (no transposition copy of the per-sample weights: 37.7 MB each way for a 512 -> 512 layer at batch 4), and the
                # (2W+1)-wide result is handed to the interpolation FIR with rows padded to 16 bytes (aligned vector loads there).
                x = conv2d_gradfix.conv_transpose2d_conv_layout(x, w, groups=groups, row_pitch=True)
            else:
                x = _conv2d_wrapper(x=x, w=_transposed_weight(w, groups), groups=groups, transpose=True,
                                    flip_weight=(not flip_weight), **kw_)
    return x
