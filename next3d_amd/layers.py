"""Fused StyleGAN2 layers on libn3d.so kernels.

A reference SynthesisLayer issues: addmm (affine) -> weight modulate / demodulate (5 elementwise + 1 reduction
over [N,O,I,3,3]) -> grouped conv -> (upfirdn2d) -> add noise -> bias_act  (reference
training_avatar_texture/networks_stylegan2.py:311-330, :34-91).  Here the same arithmetic is four launches at most:

    styles = fc(w)                                  n3d_fc
    dcoef  = rsqrt(fc(styles^2, sum_k W^2) + 1e-8)  n3d_fc   (demodulation factored through W2[o,i] = sum_k w^2)
    y      = conv(x * styles)                       n3d_conv2d, epilogue: *dcoef, +noise, +bias, lrelu, clamp
    [up]   y = FIR(y)                               n3d_upfirdn2d with the same epilogue moved behind the filter

i.e. the non-fused formulation of modulated_conv2d (:70-79) — algebraically identical to the grouped-conv branch,
but one weight tensor is shared by the whole batch, which is what the matrix cores want.
"""
import os

import numpy as np
import torch

from . import _lib
from .torch_utils.ops import conv2d_gradfix as cg
from .torch_utils.ops import upfirdn2d as uf

_SQRT2 = float(np.sqrt(2))

# Arithmetic used by the 3x3 stride-1 convolutions:
#   'bf16x3' (default): split-bf16 operands on the bf16 matrix cores, fp32 accumulation (include/n3d.h: n3d_conv2d_bf16x3)
#   'fp32'            : v_mfma_f32_32x32x2_f32 (bit-equivalent to an fmaf chain)
# Both meet the 1e-3 RGB tolerance against the reference (1.3e-5 with 'fp32', ~1e-4 with 'bf16x3').
PRECISION = os.environ.get('N3D_PRECISION', 'bf16x3')


# Pre-split hand-off between an up-sampling layer and the 3x3 convolution behind it (include/n3d.h "split8"): the FIR epilogue
# writes bf16 hi / lo planes already multiplied by the next layer's style, the convolution stages them by LDS-DMA.  These are
# module constants, not environment switches (tools/ and tests flip them in-process for A/B runs).
PRESPLIT = True
S2_PRESPLIT = True         # stride-2 encoder layers on split8 input
UP_PRESPLIT = True         # the transposed convolution's input (a block output with two consumers) converted once, LDS-DMA staging
TORGB_SIDE = True          # toRGB also writes its input as split8 for the next block's conv0 (torgb_layer)
FUSED_TORGB = True        # a block's conv1 evaluates its toRGB (<= 32 colours) in its epilogue where x has no float32 reader (fused_torgb_ok)
FUSED_TORGB_MAX = 32      # colours (module constant; tools flip it to 4 for A/B runs: only the super-resolution's toRGB layers fuse then)
FUSED_TORGB_MID = True     # ... also in blocks whose x feeds the next block (split8 side output from the same epilogue); False: last blocks only (A/B)
DIRECT_SPLIT8 = True       # 1x1 layers write split8 for their sole 3x3 consumer (conv2d_layer)
UP_PS_NCHW = True          # few-position up-sampling layers on the pre-split transposed kernel writing NCHW (networks._Block._ps_nchw)
NCHW_FIR_SPLIT8 = True     # up-sampling layers on the register-staged transposed kernel: their FIR writes split8 for conv1 (synthesis_layer)
SK_S2 = True               # few-pixel stride-2 layers on conv2d_sk_bf16x3_kernel<1, 2> (conv2d_layer)
SK_S2_MAX_IN = 16          # ... up to this input size (before the FIR); 32 would take the 32 x 32 -> 16 x 16 layer from the pre-split stride-2 kernel too (A/B: tools/ab_switch.py)
CONVERT_MAX_BYTES = int(70e6)     # see _conv3x3


def set_precision(mode):
    global PRECISION
    if mode not in ('bf16x3', 'fp32'):
        raise ValueError(mode)
    PRECISION = mode


def fc(x, weight, bias=None, wgain=1.0, bgain=1.0, act='linear', pre_square=False, post_rsqrt=False):
    """y = post(act(pre(x) @ weight.T * wgain + bias * bgain)); x [N,I], weight [O,I]."""
    n, i = x.shape
    o = weight.shape[0]
    x = x.contiguous()
    y = torch.empty([n, o], dtype=torch.float32, device=x.device)
    from .torch_utils.ops.bias_act import activation_funcs
    spec = activation_funcs[act]
    _lib.check(_lib.lib().n3d_fc(_lib.ptr(x), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(y), n, i, o, float(wgain),
                                 float(bgain), spec.cuda_idx, float(spec.def_alpha), float(spec.def_gain),
                                 1 if pre_square else 0, 1 if post_rsqrt else 0, _lib.stream()))
    return y


class PreparedConv:
    """Per-layer constants derived once from the parameters (K-major weights, squared-weight sums)."""

    def __init__(self, P, prefix, modulated, demodulate=True):
        w = P[f'{prefix}.weight']
        self.prefix = prefix
        self.out_channels, self.in_channels, self.ksize = w.shape[0], w.shape[1], w.shape[2]
        if modulated and demodulate:
            self.wt, self.wsq = cg.prep_weight(w, want_sq=True)
        else:
            self.wt, self.wsq = cg.prep_weight(w), None
        self.wt16 = cg.prep_weight_bf16x3(w) if ((self.ksize == 3 and self.in_channels % 16 == 0) or
                                                 (self.ksize == 1 and self.in_channels % 32 == 0)) else None
        self.weight = w                       # the raw parameter: the float16 blocks form per-sample weights from it (modulate_weights_f16)
        self.bias = P.get(f'{prefix}.bias')
        self.weight_gain = 1.0 / np.sqrt(self.in_channels * self.ksize ** 2)
        if modulated:
            self.affine_w = P[f'{prefix}.affine.weight']
            self.affine_b = P[f'{prefix}.affine.bias']
            self.noise_const = P.get(f'{prefix}.noise_const')
            self.noise_strength = P.get(f'{prefix}.noise_strength')


def _conv3x3(L, x, style=None, epilogue=None, out=None, rgb=None, side_style=None):
    """3x3 stride-1 convolution on the arithmetic selected by PRECISION.  A `_lib.Split8` input (written by the previous
    layer's epilogue with THIS layer's style multiplied in) goes to the pre-split kernel; `style` is then ignored."""
    if isinstance(x, _lib.Split8):
        return cg.conv_launch(x, L.wt16, 3, 0, L.out_channels, epilogue=epilogue, out=out, bf16x3=True, rgb=rgb, side_style=side_style)
    assert rgb is None and side_style is None
    n, i, h, w = x.shape
    if (PRESPLIT and PRECISION == 'bf16x3' and L.wt16 is not None and x.dtype == torch.float32 and 4 * n * i * h * w <= CONVERT_MAX_BYTES and
            cg.split8_eligible(n, i, L.out_channels, h, w) and (epilogue is None or epilogue.act in (1, 3))):
        # a float32 input whose producer could not write split8 (1x1 / stride-2 / concatenating producers): one conversion pass
        # (4 B in + 4 B out per element, with the style) buys the LDS-DMA kernel — worth it while the tensor is small next to
        # the layer's 9 * O MACs per element (measured: 10 us for 33 MB against ~75 us saved on the 64x64 x 512-channel layers)
        return cg.conv_launch(cg.split8_from_nchw(x, style), L.wt16, 3, 0, L.out_channels, epilogue=epilogue, out=out, bf16x3=True)
    if PRECISION == 'bf16x3' and L.wt16 is not None and cg.bf16x3_eligible(x.shape[1], x.shape[2], x.shape[3], 3, 0):
        return cg.conv_launch(x, L.wt16, 3, 0, L.out_channels, style=style, epilogue=epilogue, out=out, bf16x3=True)
    return cg.conv_launch(x, L.wt, 3, 0, L.out_channels, style=style, epilogue=epilogue, out=out)


class StyleBank:
    """Every style affine and demodulation coefficient of one network in TWO launches (n3d_fc_multi) instead of two per
    layer: entries = [(PreparedConv, ws slot, 'conv' | 'torgb')] in any order; `compute(ws)` returns, per entry, strided
    views (styles [N,I], dcoef [N,O] or None) into two packed buffers."""

    def __init__(self, entries, device):
        self.entries = entries
        jobs_s, jobs_d, rows_s, rows_d = [], [], [], []
        self.cols, self.dcols = [], []
        ci = co = 0
        for L, slot, kind in entries:
            g = L.weight_gain if kind == 'torgb' else 1.0
            j = _lib.FcJob()
            j.w, j.b = _lib.ptr(L.affine_w), _lib.ptr(L.affine_b)
            j.x_off, j.x_stride, j.y_off, j.y_stride = slot * 512, 0, ci, 0           # strides patched per call
            j.I, j.O, j.wgain, j.bgain = 512, L.in_channels, g / np.sqrt(512), g
            j.act, j.alpha, j.gain, j.pre_square, j.post_rsqrt = 1, 0.0, 1.0, 0, 0
            rows_s += [(len(jobs_s), o) for o in range(L.in_channels)]
            jobs_s.append(j)
            self.cols.append(ci)
            if kind == 'conv':
                d = _lib.FcJob()
                d.w, d.b = _lib.ptr(L.wsq), None
                d.x_off, d.x_stride, d.y_off, d.y_stride = ci, 0, co, 0
                d.I, d.O, d.wgain, d.bgain = L.in_channels, L.out_channels, 1.0, 1.0
                d.act, d.alpha, d.gain, d.pre_square, d.post_rsqrt = 1, 0.0, 1.0, 1, 1
                rows_d += [(len(jobs_d), o) for o in range(L.out_channels)]
                jobs_d.append(d)
                self.dcols.append(co)
                co += L.out_channels
            else:
                self.dcols.append(None)
            ci += L.in_channels
        self.total_i, self.total_o = ci, co
        self._jobs_s, self._jobs_d = jobs_s, jobs_d
        self.rows_s = torch.tensor(rows_s, dtype=torch.int32, device=device).contiguous()
        self.rows_d = torch.tensor(rows_d, dtype=torch.int32, device=device).contiguous() if rows_d else None
        self.device = device
        self._dev_jobs = {}

    def _jobs_on_device(self, ws_stride):
        """Job tables in device memory for a given ws batch stride (cached)."""
        if ws_stride not in self._dev_jobs:
            import ctypes
            def pack(jobs, xs, ys):
                arr = (_lib.FcJob * len(jobs))()
                for i, j in enumerate(jobs):
                    ctypes.memmove(ctypes.byref(arr[i]), ctypes.byref(j), ctypes.sizeof(_lib.FcJob))
                    arr[i].x_stride, arr[i].y_stride = xs, ys
                raw = bytes(arr)
                return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
            js = pack(self._jobs_s, ws_stride, self.total_i)
            jd = pack(self._jobs_d, self.total_i, self.total_o) if self._jobs_d else None
            self._dev_jobs[ws_stride] = (js, jd)
        return self._dev_jobs[ws_stride]

    def compute(self, ws):
        """ws [N, num_ws, 512] float32 with unit inner strides (any batch stride)."""
        assert ws.dtype == torch.float32 and ws.stride(2) == 1 and ws.stride(1) == 512 and ws.shape[2] == 512
        n = ws.shape[0]
        js, jd = self._jobs_on_device(ws.stride(0))
        styles = torch.empty([n, self.total_i], dtype=torch.float32, device=ws.device)
        L_ = _lib.lib()
        _lib.check(L_.n3d_fc_multi(_lib.ptr(js), _lib.ptr(self.rows_s), self.rows_s.shape[0], _lib.ptr(ws), _lib.ptr(styles), n, _lib.stream()))
        dcoef = None
        if jd is not None:
            dcoef = torch.empty([n, self.total_o], dtype=torch.float32, device=ws.device)
            _lib.check(L_.n3d_fc_multi(_lib.ptr(jd), _lib.ptr(self.rows_d), self.rows_d.shape[0], _lib.ptr(styles), _lib.ptr(dcoef), n, _lib.stream()))
        out = {}
        for (Lc, slot, kind), c0, d0 in zip(self.entries, self.cols, self.dcols):
            out[Lc.prefix] = (styles[:, c0:c0 + Lc.in_channels], None if d0 is None else dcoef[:, d0:d0 + Lc.out_channels])
        return out


def presplit_ok(n, next_layer, h, w):
    """May the layer in front of `next_layer` (a 3x3 stride-1 PreparedConv running on [n, I, h, w]) hand its output over in
    the split8 layout?"""
    return (PRESPLIT and PRECISION == 'bf16x3' and next_layer.wt16 is not None and next_layer.ksize == 3 and
            cg.split8_eligible(n, next_layer.in_channels, next_layer.out_channels, h, w))


def synthesis_layer(L, x, w, fir, up=1, noise_mode='const', conv_clamp=None, gain=1.0, styles=None, dcoef=None, out=None, _noise=None,
                    split_for=None, x_split8=None, split_for_nchw=None, ps_nchw=False, rgb=None, rgb_side_style=None):
    """SynthesisLayer.forward (reference networks_stylegan2.py:311-330).  `styles`/`dcoef` may come pre-computed from a
    StyleBank; otherwise they are computed here from the latent `w`.  noise_mode 'const' adds the learned noise image,
    'random' a fresh N(0,1) image PER SAMPLE (:318-319, the reference's training-time default; drawn with torch.randn from
    the device generator like the reference) — the kernels' epilogue takes one noise image per launch, so that mode runs
    the layer sample by sample.  `split_for` (up = 2 only): the styles [N,O] of the 3x3 layer that consumes this layer's
    output -> the result is a `_lib.Split8` carrying them (see presplit_ok).  `split_for_nchw`: the same hand-over for the up-sampling
    layers whose transposed convolution runs on the register-staged kernel (few positions: split-K, float32 NCHW result): their FIR
    reads that result and writes split8 (n3d_fir4_split8_nchw, pad 1) instead of float32 + a conversion pass in front of conv1.
    `rgb` (up = 1, x a `_lib.Split8`): see fused_torgb_ok — the layer's result is then the partial colour tensor for torgb_combine.
    `rgb_side_style` (with rgb): the styles of the layer's second reader -> (partial, `_lib.Split8` of the layer's output * those styles).
    (The reference's float16 blocks run on their own kernels: synthesis_layer_f16.)"""
    if styles is None:
        styles = fc(w, L.affine_w, L.affine_b, wgain=1.0 / np.sqrt(w.shape[1]))
        dcoef = fc(styles, L.wsq, pre_square=True, post_rsqrt=True)
    if noise_mode not in ('random', 'const', 'none'):
        raise RuntimeError(f'noise_mode {noise_mode!r}: random / const / none')
    if noise_mode == 'random' and L.noise_const is not None and _noise is None:
        n = x.shape[0]
        draws = torch.randn([n, *L.noise_const.shape], dtype=torch.float32, device=L.noise_const.device)
        assert split_for is None
        outs = [synthesis_layer(L, x[i:i + 1], w, fir, up=up, noise_mode='random', conv_clamp=conv_clamp, gain=gain, styles=styles[i:i + 1],
                                dcoef=dcoef[i:i + 1], out=None if out is None else out[i:i + 1], _noise=draws[i]) for i in range(n)]
        return out if out is not None else torch.cat(outs, 0)
    noise = _noise if _noise is not None else (L.noise_const if noise_mode == 'const' else None)
    act = dict(noise=noise, noise_strength=L.noise_strength if noise is not None else None, bias=L.bias, act='lrelu',
               gain=_SQRT2 * gain, clamp=None if conv_clamp is None else conv_clamp * gain)
    if up == 1:
        return _conv3x3(L, x, style=styles, epilogue=_lib.make_epilogue(row_scale=dcoef, **act), out=out, rgb=rgb, side_style=rgb_side_style)
    assert up == 2 and out is None and rgb is None
    if split_for is not None:       # transposed conv -> channel-interleaved z -> FIR + epilogue + next style + hi/lo split
        zepi = _lib.make_epilogue(row_scale=dcoef)
        if UP_PRESPLIT and x.shape[1] % 16 == 0:
            # modulation + operand split once, then pure LDS-DMA staging; `x_split8`: the previous block's toRGB made it already
            xs = x_split8 if x_split8 is not None else cg.split8_from_nchw(x, styles)
            t = cg.conv_launch(xs, L.wt16, 3, 2, L.out_channels, epilogue=zepi, bf16x3=True, out_c8=True)
        else:
            t = cg.conv_launch(x, L.wt16, 3, 2, L.out_channels, style=styles, epilogue=zepi, bf16x3=True, out_c8=True)
        return uf._fir4_split8(t, fir, 4, _lib.make_epilogue(**act), split_for)
    if PRECISION == 'bf16x3' and L.wt16 is not None and cg.bf16x3_eligible(x.shape[1], x.shape[2], x.shape[3], 3, 2):
        if ps_nchw:     # the pre-split transposed kernel writing float32 NCHW (same products as the register-staged kernel, no split-K reduction pass)
            xs = x_split8 if x_split8 is not None else cg.split8_from_nchw(x, styles)
            t = cg.conv_launch(xs, L.wt16, 3, 2, L.out_channels, epilogue=_lib.make_epilogue(row_scale=dcoef), bf16x3=True, row_pitch=True)
        else:
            t = cg.conv_launch(x, L.wt16, 3, 2, L.out_channels, style=styles, epilogue=_lib.make_epilogue(row_scale=dcoef), bf16x3=True,
                               row_pitch=True)
        if split_for_nchw is not None:
            return uf._fir4_split8_nchw(t, fir, 1, gain=4, epilogue=_lib.make_epilogue(**act), out_scale=split_for_nchw)
    else:
        t = cg.conv_launch(x, L.wt, 3, 2, L.out_channels, style=styles, epilogue=_lib.make_epilogue(row_scale=dcoef), row_pitch=True)
    return uf.upfirdn2d(t, fir, padding=[1, 1, 1, 1], gain=4, _epilogue=_lib.make_epilogue(**act))


def _conv1x1(L, x, style=None, epilogue=None, out=None, out_split8=False, side_style=None):
    """1x1 stride-1 convolution on the arithmetic selected by PRECISION.  out_split8: return a `_lib.Split8` (see conv2d_layer);
    side_style: return (y, `_lib.Split8` of x * side_style) (see torgb_layer)."""
    if PRECISION == 'bf16x3' and L.wt16 is not None and cg.bf16x3_eligible(x.shape[1], x.shape[2], x.shape[3], 1, 0):
        return cg.conv_launch(x, L.wt16, 1, 0, L.out_channels, style=style, epilogue=epilogue, out=out, bf16x3=True, out_split8=out_split8,
                              side_style=side_style)
    assert not out_split8 and side_style is None
    return cg.conv_launch(x, L.wt, 1, 0, L.out_channels, style=style, epilogue=epilogue, out=out)


def fused_torgb_ok(conv, torgb, x, noise_mode):
    """Can `conv` (a block's conv1, input `x`) evaluate `torgb` in its own epilogue (n3d_conv2d_desc.rgb_*: at most 32 colours — up to 4 on the
    VALU, the backbones' 32-channel toRGB layers as an epilogue contraction on the matrix cores —, the pre-split stride-1 kernel without
    split-K)?  The caller has established that x has no other float32 reader (a network's last block, or a next block that takes the split8
    side output).  The 512 x 512 x 128-channel feature map of the super-resolution's last block (537 MB per step at batch 4) and the 256 x 256 x
    128-channel maps of the texture / mouth / blending networks' last blocks (134 MB each) are then neither written nor read back."""
    return (FUSED_TORGB and PRECISION == 'bf16x3' and isinstance(x, _lib.Split8) and noise_mode != 'random' and conv.wt16 is not None and
            torgb.out_channels <= FUSED_TORGB_MAX and torgb.ksize == 1 and torgb.in_channels == conv.out_channels and
            cg.split8_ksplit(x.shape[0], x.shape[1], conv.out_channels, x.shape[2], x.shape[3]) == 1)


def torgb_combine(L, partial, conv_clamp=None, residual=None, residual_up_filter=None):
    """toRGB's epilogue on the partial colours of a fused last layer (synthesis_layer(..., rgb=...)): bias, clamp, skip image (torgb_layer)."""
    return cg.rgb_combine(partial, _lib.make_epilogue(bias=L.bias, clamp=conv_clamp, residual=residual, residual_up_filter=residual_up_filter))


def torgb_side_ok(L, x):
    """Can toRGB layer `L` on the float32 feature map `x` also write x's split8 form for the next block (torgb_layer side_style)?"""
    return (TORGB_SIDE and UP_PRESPLIT and PRECISION == 'bf16x3' and L.wt16 is not None and isinstance(x, torch.Tensor) and x.dtype == torch.float32 and
            L.out_channels <= 128 and x.shape[1] % 32 == 0 and cg.bf16x3_eligible(x.shape[1], x.shape[2], x.shape[3], 1, 0))


def torgb_layer(L, x, w, conv_clamp=None, residual=None, styles=None, residual_up_filter=None, side_style=None):
    """ToRGBLayer.forward (reference networks_stylegan2.py:353-357) + the skip-image accumulation (:580-584).  With
    `residual_up_filter`, `residual` is the PREVIOUS block's half-resolution image and upsample2d (:582) is evaluated
    inside the convolution's epilogue.  With `side_style` (the styles of the NEXT block's transposed convolution, the other reader
    of x: SynthesisBlock.forward :469-475) the kernel also writes x * side_style in the split8 layout -> (img, `_lib.Split8`):
    the feature map is read from HBM once for both readers instead of once by toRGB and once by n3d_split8_from_nchw."""
    g = L.weight_gain
    if styles is None:
        styles = fc(w, L.affine_w, L.affine_b, wgain=g / np.sqrt(w.shape[1]), bgain=g)
    return _conv1x1(L, x, style=styles, epilogue=_lib.make_epilogue(bias=L.bias, clamp=conv_clamp, residual=residual,
                                                                    residual_up_filter=residual_up_filter), side_style=side_style)


def conv2d_layer(L, x, fir, activation='linear', down=1, conv_clamp=None, gain=1.0, residual=None, out=None, sole_consumer=None):
    """Conv2dLayer.forward (reference networks_stylegan2.py:173-183), up=1.  `sole_consumer`: the un-modulated 3x3 stride-1
    PreparedConv that is the ONLY reader of this 1x1 layer's result; when that layer runs on the pre-split kernel the result is
    written as a `_lib.Split8` by this layer's epilogue (no conversion pass, and no CONVERT_MAX_BYTES limit)."""
    from .torch_utils.ops.bias_act import activation_funcs
    epi = _lib.make_epilogue(const_scale=L.weight_gain, bias=L.bias, act=activation,
                             gain=activation_funcs[activation].def_gain * gain,
                             clamp=None if conv_clamp is None else conv_clamp * gain, residual=residual)
    if down == 1 and L.ksize == 3:
        return _conv3x3(L, x, epilogue=epi, out=out)
    if down == 1 and L.ksize == 1:
        c = sole_consumer
        s8 = (DIRECT_SPLIT8 and c is not None and out is None and PRESPLIT and PRECISION == 'bf16x3' and L.wt16 is not None and c.wt16 is not None and c.ksize == 3 and
              x.dtype == torch.float32 and L.out_channels % 32 == 0 and cg.bf16x3_eligible(x.shape[1], x.shape[2], x.shape[3], 1, 0) and
              cg.split8_eligible(x.shape[0], L.out_channels, c.out_channels, x.shape[2], x.shape[3]))
        return _conv1x1(L, x, epilogue=epi, out=out, out_split8=bool(s8))
    if down == 1:
        return cg.conv_launch(x, L.wt, L.ksize, 0, L.out_channels, epilogue=epi, out=out)
    assert down == 2 and L.ksize == 3
    sk_s2 = (SK_S2 and PRECISION == 'bf16x3' and L.wt16 is not None and x.dtype == torch.float32 and x.shape[2] <= SK_S2_MAX_IN and
             cg.sk_s2_eligible(x.shape[0], x.shape[1], L.out_channels, x.shape[2] + 1, x.shape[3] + 1))
    if not sk_s2 and PRECISION == 'bf16x3' and L.wt16 is not None and S2_PRESPLIT and x.shape[1] % 16 == 0 and x.shape[2] >= 32 and epi.act in (1, 3):
        # FIR writing split8 -> the LDS-DMA stride-2 kernel (the register-staged one pays a stride-1 chunk's
        # staging for a quarter of its MFMAs per stage; docs/history/DESIGN_rounds1-4.md 3.1c)
        x = uf._fir4_split8_nchw(x, fir, 2) if x.shape[1] % 8 == 0 and tuple(fir.shape) == (4, 4) else cg.split8_from_nchw(uf.upfirdn2d(x, fir, padding=[2, 2, 2, 2]))
        return cg.conv_launch(x, L.wt16, 3, 1, L.out_channels, epilogue=epi, out=out, bf16x3=True)
    if not sk_s2 and PRECISION == 'bf16x3' and L.wt16 is not None and cg.bf16x3_eligible(x.shape[1], x.shape[2] + 1, x.shape[3] + 1, 3, 1):
        # the (W+1)-wide FIR output goes to the stride-2 kernel with rows padded to 16 bytes (aligned float4 FIR stores)
        x = uf.upfirdn2d(x, fir, padding=[2, 2, 2, 2], _row_pitch=True)
        return cg.conv_launch(x, L.wt16, 3, 1, L.out_channels, epilogue=epi, out=out, bf16x3=True)
    x = uf.upfirdn2d(x, fir, padding=[2, 2, 2, 2])
    if sk_s2:
        # few-pixel stride-2 layers (<= 17 x 17 behind the FIR): the one-launch split-bf16 kernel instead of fp32 MFMA + split-K 16 + reduce launch
        return cg.conv_launch(x, L.wt16, 3, 1, L.out_channels, epilogue=epi, out=out, bf16x3=True)
    return cg.conv_launch(x, L.wt, 3, 1, L.out_channels, epilogue=epi, out=out)


# ------------------------------------------------------------------------------------------------------------------------------
# The reference's float16 blocks (training/networks_stylegan2.py:417-452: `use_fp16 and not force_fp32`) on the f16 matrix cores:
# h8 activations, per-sample float16 weights (the FUSED modulated_conv2d branch, :53-91), float32 accumulation, one rounding per
# operator — include/n3d.h "FLOAT16 blocks".

def f16_layer_ok(L, h, w, up):
    """Can SynthesisLayer `L` on an [*, I, h, w] input run on the float16 kernels?"""
    return (L.ksize == 3 and L.in_channels % 16 == 0 and L.out_channels % 64 == 0 and L.in_channels * 9 <= 4608 and
            ((up == 1 and h >= 16 and w >= 32) or (up == 2 and h >= 4 and w >= 4)))


def modulate_weights_f16(L, styles, demodulate=True):
    """Per-sample float16 weights of layer `L` for styles [N,I] (n3d_modulate_weights_f16) -> flat float16 tensor."""
    n = styles.shape[0]
    o, i, k = L.out_channels, L.in_channels, L.ksize
    assert styles.dtype == torch.float32 and styles.stride(1) == 1 and styles.shape[1] == i
    w = L.weight if L.weight.is_contiguous() else L.weight.contiguous()
    out = torch.empty(n * o * i * k * k, dtype=torch.float16, device=styles.device)
    _lib.check(_lib.lib().n3d_modulate_weights_f16(_lib.ptr(w), _lib.ptr(styles), styles.stride(0), _lib.ptr(out), n, o, i, k,
                                                   1 if demodulate else 0, _lib.stream()))
    out._keep = (w, styles)
    return out


def conv2d_f16(x, w16, out_channels, mode, epilogue=None, rgb=None):
    """n3d_conv2d_f16: x `_lib.H8` [N,I,H,W], w16 from modulate_weights_f16 -> `_lib.H8` [N,O,H,W] (mode 0) or [N,O,2H+1,2W+1] (mode 2).
    rgb = (per-sample float16 toRGB weights [N, C, O] flat, C <= 4) (mode 0): the layer's only reader, its toRGB, is evaluated in the epilogue —
    returns the partial colours [N, O/64, C, H, W] float32 for torgb_combine_f16; the feature map is not written."""
    n, i, h, w = x.shape
    oh, ow = (h, w) if mode == 0 else (2 * h + 1, 2 * w + 1)
    d = _lib.Conv2dDesc()
    if rgb is not None:
        rw, c = rgb
        assert mode == 0 and rw.dtype == torch.float16 and rw.numel() == n * c * out_channels and 1 <= c <= 4
        y = None
        partial = torch.empty([n, out_channels // 64, c, h, w], dtype=torch.float32, device=x.device)
        d.rgb_weight, d.rgb_partial, d.rgb_channels = _lib.ptr(rw), _lib.ptr(partial), c
    else:
        y = _lib.H8(n, out_channels, oh, ow, x.device)
    d.x, d.wt, d.style, d.y, d.workspace = _lib.ptr(x.data), _lib.ptr(w16), None, (_lib.ptr(y.data) if y is not None else None), None
    d.N, d.I, d.O, d.H, d.W = n, i, out_channels, h, w
    d.ksize, d.mode, d.ksplit = 3, mode, 1
    d.x_layout = d.y_layout = 3
    d.epi = epilogue if epilogue is not None else _lib.make_epilogue()
    _lib.check(_lib.lib().n3d_conv2d_f16(d, _lib.stream()))
    if rgb is not None:
        partial._keep = (x, w16, d.epi, rgb[0])
        return partial
    y._keep = (x, w16, d.epi)
    return y


# bias_act rounding of the float16 blocks: False = bias_act.cu (float32 inside, one rounding: what the reference does on a GPU);
# True = _bias_act_ref on half tensors (what the reference does off-GPU: every step rounds) — only to compare with the reference's
# own CPU run of its float16 branch (tests/golden/*_fp16sr.npz).
F16_REF_CPU_ROUNDING = os.environ.get('N3D_F16_REF_CPU_ROUNDING', '0') == '1'
fir_factor = uf.fir_factor


def modulate_weights_f16_multi(entries, styles_base, n):
    """[(layer, styles view [N,I] into `styles_base` (StyleBank's packed buffer), demodulate)] -> list of flat float16 weight
    tensors, ONE launch (n3d_modulate_weights_f16_multi)."""
    import ctypes
    jobs = (_lib.ModwJob * len(entries))()
    outs, keep = [], []
    esz = styles_base.element_size()
    for j, (L, st, demod) in enumerate(entries):
        assert st.dtype == torch.float32 and st.stride(1) == 1 and st.stride(0) == styles_base.stride(0) and st.shape == (n, L.in_channels)
        w = L.weight if L.weight.is_contiguous() else L.weight.contiguous()
        out = torch.empty(n * L.out_channels * L.in_channels * L.ksize ** 2, dtype=torch.float16, device=styles_base.device)
        jobs[j].w, jobs[j].w16 = w.data_ptr(), out.data_ptr()
        jobs[j].styles_offset = (st.data_ptr() - styles_base.data_ptr()) // esz
        jobs[j].O, jobs[j].I, jobs[j].ksize, jobs[j].demodulate = L.out_channels, L.in_channels, L.ksize, 1 if demod else 0
        outs.append(out); keep.append(w)
    _lib.check(_lib.lib().n3d_modulate_weights_f16_multi(ctypes.cast(jobs, ctypes.c_void_p), len(entries), _lib.ptr(styles_base),
                                                         styles_base.stride(0), n, _lib.stream()))
    for o in outs:
        o._keep = (keep, styles_base)
    return outs


def synthesis_layer_f16(L, x, styles, fir, up=1, noise_mode='none', conv_clamp=None, gain=1.0, w16=None, rgb=None, _noise=None):
    """SynthesisLayer.forward of a float16 block (training/networks_stylegan2.py:311-330 with x.dtype == float16, fused_modconv):
    x `_lib.H8` -> `_lib.H8`.  `w16`: the layer's per-sample weights when already formed (modulate_weights_f16_multi).
    noise_mode 'random' (the reference's default, :318-319): a fresh N(0,1) image per sample, drawn with torch.randn from the device generator like the
    reference; the kernels' epilogue takes one noise image per launch, so the noisy part of the layer runs sample by sample (as synthesis_layer does)."""
    if noise_mode not in ('random', 'const', 'none'):
        raise RuntimeError(f'noise_mode {noise_mode!r}: random / const / none')
    if noise_mode == 'random' and L.noise_const is not None and _noise is None:
        n = x.shape[0]
        if w16 is None:
            w16 = modulate_weights_f16(L, styles, demodulate=True)
        draws = torch.randn([n, *L.noise_const.shape], dtype=torch.float32, device=L.noise_const.device)
        kw = dict(up=up, noise_mode='random', conv_clamp=conv_clamp, gain=gain)
        if up == 2:            # the transposed convolution carries no noise: one launch for the batch, the FIR + epilogue per sample
            z = conv2d_f16(x, w16, L.out_channels, 2)
            outs = [synthesis_layer_f16(L, z.sample(i), None, fir, w16=w16, _noise=(draws[i], True), **kw) for i in range(n)]
        else:
            per = w16.numel() // n
            rw = None if rgb is None else rgb[0].reshape(n, -1)
            outs = [synthesis_layer_f16(L, x.sample(i), None, fir, w16=w16[i * per:(i + 1) * per], rgb=None if rgb is None else (rw[i], rgb[1]),
                                        _noise=(draws[i], False), **kw) for i in range(n)]
        if rgb is not None:
            return torch.cat(outs, 0)                                    # partial colours [N, O/64, C, H, W]
        y = _lib.H8(n, L.out_channels, outs[0].shape[2], outs[0].shape[3], x.device)
        for i, o in enumerate(outs):
            y.data[i:i + 1].copy_(o.data)
        return y
    noise = _noise[0] if _noise is not None else (L.noise_const if noise_mode == 'const' else None)
    epi = _lib.make_epilogue(noise=noise, noise_strength=L.noise_strength if noise is not None else None, bias=L.bias, act='lrelu',
                             gain=_SQRT2 * gain, clamp=None if conv_clamp is None else conv_clamp * gain, round_f16=2 if F16_REF_CPU_ROUNDING else 0)
    if _noise is not None and _noise[1]:                                  # (per-sample random noise, up = 2: x is this sample's transposed-convolution result)
        return fir4_h8(x, fir, epi)
    if w16 is None:
        w16 = modulate_weights_f16(L, styles, demodulate=True)
    if up == 1:
        return conv2d_f16(x, w16, L.out_channels, 0, epi, rgb=rgb)
    assert rgb is None
    z = conv2d_f16(x, w16, L.out_channels, 2)
    return fir4_h8(z, fir, epi)


def fir4_h8(z, fir, epi, gain=4.0):
    """upfirdn2d(z, fir, padding=1, gain) + the layer epilogue on h8 tensors (n3d_fir4_h8): [N,C,H,W] -> [N,C,H-1,W-1]."""
    n, c, h, w = z.shape
    y = _lib.H8(n, c, h - 1, w - 1, z.device)
    f1d = fir_factor(fir) if uf.FIR_SEP else None
    _lib.check(_lib.lib().n3d_fir4_h8(_lib.ptr(z.data), _lib.ptr(fir), _lib.ptr(f1d), _lib.ptr(y.data), n, c, h, w, 0, float(gain), epi, _lib.stream()))
    y._keep = (z, epi, f1d)
    return y


def torgb_combine_f16(L, partial, fir, conv_clamp=None, img_lo=None):
    """toRGB's float16 epilogue + skip image on the partial colours of a fused last layer (synthesis_layer_f16(..., rgb=...)); the
    arithmetic of torgb_layer_f16: float32 sum -> float16, bias_act on float16, float32 skip image."""
    if img_lo is not None:
        img_lo = img_lo.contiguous()
        assert tuple(fir.shape) == (4, 4)
    return cg.rgb_combine(partial, _lib.make_epilogue(bias=L.bias, clamp=conv_clamp, residual=img_lo, residual_up_filter=fir if img_lo is not None else None,
                                                      round_f16=1))


def torgb_layer_f16(L, x, styles, fir, conv_clamp=None, img_lo=None, w16=None):
    """ToRGBLayer.forward of a float16 block + the skip-image update (training/networks_stylegan2.py:353-357, :446-451): x `_lib.H8`,
    styles [N,I] already multiplied by weight_gain (StyleBank 'torgb'), img_lo the previous block's float32 image or None ->
    float32 image [N,O,H,W]."""
    n, c, h, w = x.shape
    o = L.out_channels
    if w16 is None:
        w16 = modulate_weights_f16(L, styles, demodulate=False)
    img = torch.empty(n, o, h, w, dtype=torch.float32, device=x.device)
    if img_lo is not None:
        img_lo = img_lo.contiguous()
        assert tuple(img_lo.shape) == (n, o, h // 2, w // 2) and tuple(fir.shape) == (4, 4)
    _lib.check(_lib.lib().n3d_torgb_h8(_lib.ptr(x.data), _lib.ptr(w16), _lib.ptr(L.bias), _lib.ptr(img_lo), _lib.ptr(fir) if img_lo is not None else None,
                                       _lib.ptr(img), n, c, o, h, w, float(-1 if conv_clamp is None else conv_clamp), _lib.stream()))
    img._keep = (x, w16, img_lo)
    return img
