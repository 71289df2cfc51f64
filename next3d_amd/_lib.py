"""ctypes binding of libn3d.so (the C ABI declared in include/n3d.h).

There is no fallback: if the library is missing or a call fails, a RuntimeError is raised — the
analogue of the reference's TORCH_CHECK -> RuntimeError (torch_utils/ops/bias_act.cpp:39-55).
The reference's `_init()` (torch_utils/ops/bias_act.py:40-50) JIT-builds a pybind plugin; here
the library is built ahead of time by `python -m next3d_amd.build` and loaded once.
"""
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('N3D_LIB') or os.path.join(_HERE, 'libn3d.so')     # N3D_LIB: A/B-compare two builds on one box
ABI_VERSION = 8

c_void_p, c_int, c_int64, c_float, c_double = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_double

ACT_IDS = {'linear': 1, 'relu': 2, 'lrelu': 3, 'tanh': 4, 'sigmoid': 5, 'elu': 6, 'selu': 7, 'softplus': 8, 'swish': 9}
K_BIAS_ACT, K_UPFIRDN2D, K_CONV2D, K_FC, K_RENDER, K_RASTER, K_MISC, K_CONV2D_BF16X3 = range(8)
FAMILY_NAMES = ['bias_act', 'upfirdn2d', 'conv2d', 'fc', 'render', 'raster', 'misc', 'conv2d_bf16x3', 'conv1x1_bf16x3', 'conv2d_f16', 'render_rays']


class Epilogue(ctypes.Structure):
    _fields_ = [('row_scale', c_void_p), ('noise', c_void_p), ('noise_strength', c_void_p), ('bias', c_void_p),
                ('residual', c_void_p), ('residual_batch_stride', c_int64), ('row_scale_stride', c_int64),
                ('const_scale', c_float), ('act', c_int),
                ('alpha', c_float), ('gain', c_float), ('clamp', c_float), ('residual_up_filter', c_void_p), ('round_f16', c_int)]


class Conv2dDesc(ctypes.Structure):
    _fields_ = [('x', c_void_p), ('wt', c_void_p), ('style', c_void_p), ('y', c_void_p), ('workspace', c_void_p),
                ('N', c_int), ('I', c_int), ('O', c_int), ('H', c_int), ('W', c_int),
                ('ksize', c_int), ('mode', c_int), ('ksplit', c_int),
                ('x_batch_stride', c_int64), ('y_batch_stride', c_int64), ('style_stride', c_int64),
                ('x_row_stride', c_int64), ('y_row_stride', c_int64), ('epi', Epilogue), ('x_layout', c_int), ('y_layout', c_int),
                ('side_split8', c_void_p), ('side_style', c_void_p), ('side_style_stride', c_int64), ('wt_batch_stride', c_int64),
                ('rgb_weight', c_void_p), ('rgb_style', c_void_p), ('rgb_partial', c_void_p), ('rgb_channels', c_int), ('rgb_style_stride', c_int64),
                ('tickets', c_void_p), ('ticket_count', c_int)]


class RenderOpts(ctypes.Structure):
    _fields_ = [('white_back', c_int), ('disparity_space_sampling', c_int), ('ray_start', c_float), ('ray_end', c_float), ('auto_bounds', c_int),
                ('box_side', c_float), ('ray_bounds_ws', c_void_p), ('density_noise', c_float), ('density_noise_coarse', c_void_p),
                ('density_noise_fine', c_void_p), ('fine_depths_out', c_void_p), ('fine_depths_in', c_void_p), ('decoder_split_bf16', c_int)]


class ModwJob(ctypes.Structure):
    _fields_ = [('w', c_void_p), ('w16', c_void_p), ('styles_offset', c_int64), ('O', c_int), ('I', c_int), ('ksize', c_int), ('demodulate', c_int)]


class FcJob(ctypes.Structure):
    _fields_ = [('w', c_void_p), ('b', c_void_p), ('x_off', c_int64), ('x_stride', c_int64), ('y_off', c_int64),
                ('y_stride', c_int64), ('I', c_int), ('O', c_int), ('wgain', c_float), ('bgain', c_float), ('act', c_int),
                ('alpha', c_float), ('gain', c_float), ('pre_square', c_int), ('post_rsqrt', c_int)]


_SIGNATURES = {
    'n3d_abi_version': (c_int, []),
    'n3d_last_error': (ctypes.c_char_p, []),
    'n3d_prof_enable': (c_int, [c_int]),
    'n3d_prof_reset': (c_int, []),
    'n3d_prof_read': (c_int, [c_int, ctypes.POINTER(c_double), ctypes.POINTER(c_int64), ctypes.POINTER(c_double),
                              ctypes.POINTER(c_double)]),
    'n3d_conv2d_sk_workspace': (c_int64, [c_int] * 6 + [ctypes.POINTER(c_int)]),
    'n3d_rgb_combine': (c_int, [c_void_p, c_void_p] + [c_int] * 5 + [ctypes.POINTER(Epilogue), c_void_p]),
    'n3d_bias_act': (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int64, c_int, c_int, c_float, c_float,
                             c_float, c_void_p]),
    'n3d_upfirdn2d': (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 15 + [c_float, c_int64, c_int64,
                              ctypes.POINTER(Epilogue), c_void_p]),
    'n3d_upfirdn2d_pitched': (c_int, [c_void_p, c_void_p, c_void_p] + [c_int] * 4 + [c_int64, c_int64] + [c_int] * 11 +
                              [c_float, c_int64, c_int64, ctypes.POINTER(Epilogue), c_void_p]),
    'n3d_filtered_lrelu': (c_int, [c_void_p] * 5 + [c_int] * 14 + [c_float] * 3 + [c_int, c_void_p]),
    'n3d_fir4_split8': (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_int64, c_int64, c_int, c_float, ctypes.POINTER(Epilogue), c_void_p, c_int64, c_void_p]),
    'n3d_fir4_split8_sep': (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_int64, c_int64, c_int, c_float, ctypes.POINTER(Epilogue), c_void_p, c_int64, c_void_p]),
    'n3d_fir4_split8_nchw_sep': (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_int64, c_int64, c_int, c_int, c_float, ctypes.POINTER(Epilogue), c_void_p, c_int64, c_void_p]),
    'n3d_fir4_split8_nchw': (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_int64, c_int64, c_int, c_int, c_float, ctypes.POINTER(Epilogue), c_void_p, c_int64, c_void_p]),
    'n3d_conv2d_sk_eligible': (c_int, [c_int] * 5),
    'n3d_conv2d_sk_s2_eligible': (c_int, [c_int] * 5),
    'n3d_conv2d_up_sk_eligible': (c_int, [c_int] * 5),
    'n3d_conv2d_split8_eligible': (c_int, [c_int] * 5),
    'n3d_conv2d_split8_ksplit': (c_int, [c_int] * 5),
    'n3d_split8_from_nchw': (c_int, [c_void_p] * 3 + [c_int, c_int, c_int64, c_int64, c_int64, c_void_p]),
    'n3d_conv2d_prep_weight': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'n3d_conv2d': (c_int, [ctypes.POINTER(Conv2dDesc), c_void_p]),
    'n3d_conv2d_prep_weight_grouped': (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int64, c_void_p]),
    'n3d_conv2d_prep_weight_bf16x3': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'n3d_conv2d_bf16x3': (c_int, [ctypes.POINTER(Conv2dDesc), c_void_p]),
    'n3d_conv2d_bf16x3_blocks': (c_int, [c_int] * 5),
    'n3d_blend_planes': (c_int, [c_void_p] * 6 + [c_int] * 3 + [c_void_p]),
    'n3d_blend_planes_views': (c_int, [c_void_p] * 6 + [c_int] * 7 + [c_void_p]),
    'n3d_unpack_inputs': (c_int, [c_void_p, c_int64, c_void_p, c_int64] + [c_void_p] * 4 + [c_int] * 3 + [c_void_p]),
    'n3d_planes_to_channels_last': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'n3d_render_rays': (c_int, [c_void_p] * 14 + [c_int] * 6 + [c_float, c_float, c_void_p]),
    'n3d_render_rays_ex': (c_int, [c_void_p] * 14 + [c_int] * 6 + [c_float, c_float, ctypes.POINTER(RenderOpts), c_void_p]),
    'n3d_sample_points': (c_int, [c_void_p] * 8 + [c_int, c_int64, c_int, c_int, c_float, c_void_p]),
    'n3d_rasterize_views': (c_int, [c_void_p] * 6 + [c_int, c_int] + [c_void_p] * 5 + [c_int] * 7 + [c_float] * 4 +
                            [c_int, c_int, c_void_p]),
    'n3d_rasterize_meshes': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p] + [c_int] * 6 + [c_void_p]),
    'n3d_flood_fill': (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_float, c_float, c_void_p]),
    'n3d_texture_project': (c_int, [c_void_p] * 3 + [c_int] * 9 + [c_void_p]),
    'n3d_texture_project_planes': (c_int, [c_void_p] * 5 + [c_int] * 8 + [c_void_p]),
    'n3d_mouth_bbox': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    'n3d_resize_aa': (c_int, [c_void_p] * 4 + [c_int] * 7 + [c_void_p]),
    'n3d_resize_aa_strided': (c_int, [c_void_p, c_int64] + [c_void_p] * 3 + [c_int] * 7 + [c_void_p]),
    'n3d_normalize_2nd_moment': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_float, c_void_p]),
    'n3d_normalize_2nd_moment_f64': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_float, c_void_p]),
    'n3d_truncate_ws': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    'n3d_fma': (c_int, [c_void_p] * 4 + [c_int64] * 6 + [c_void_p]),
    'n3d_to_uint8': (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    'n3d_layout_grid_u8': (c_int, [c_void_p, c_void_p] + [c_int] * 7 + [c_void_p]),
    'n3d_cast': (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    'n3d_modulate_weights_f16': (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'n3d_conv2d_f16': (c_int, [ctypes.POINTER(Conv2dDesc), c_void_p]),
    'n3d_fir4_h8': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, ctypes.POINTER(Epilogue), c_void_p]),
    'n3d_modulate_weights_f16_multi': (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p]),
    'n3d_torgb_h8': (c_int, [c_void_p] * 6 + [c_int] * 5 + [c_float, c_void_p]),
    'n3d_cast_h8': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_int, c_void_p]),
    'n3d_cast_h8_ex': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int64, c_int64, c_int, c_int, c_void_p]),
    'n3d_fc_multi': (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    'n3d_fc': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_float, c_int, c_float,
                       c_float, c_int, c_int, c_void_p]),
}

_lib = None


def exported_symbols():
    """Names include/n3d.h declares; tests check every one resolves in the built library."""
    return sorted(_SIGNATURES)


def lib():
    """The library handle."""
    return _handle()


def _handle():
    """Load libn3d.so once.  Raises RuntimeError (never falls back) when it is missing or stale."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f'{LIB_PATH} not found: build it with `python -m next3d_amd.build` '
                               '(there is no CPU or PyTorch fallback for the n3d ops)')
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            if os.environ.get('N3D_LIB') and not hasattr(handle, name):
                continue                         # A/B runs against an older build (N3D_LIB) may lack newer entry points
            fn = getattr(handle, name)           # AttributeError -> missing symbol
            fn.restype, fn.argtypes = res, args
        if handle.n3d_abi_version() != ABI_VERSION:
            raise RuntimeError(f'libn3d.so ABI {handle.n3d_abi_version()} != binding ABI {ABI_VERSION}; rebuild')
        _lib = handle
    return _lib


_tls = threading.local()


def check(rc):
    """Result check of an entry point — and the end of the keep-alive window of `ptr()` (below): by now the launch is enqueued."""
    held = getattr(_tls, 'held', None)
    if held:
        held.clear()
    if rc != 0:
        raise RuntimeError('libn3d: ' + _handle().n3d_last_error().decode())


_raw_stream = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_get_device = getattr(torch._C, '_cuda_getDevice', None) or torch.cuda.current_device


def _stream_handle():
    """torch's current HIP stream of the current device as an integer handle — the raw getter (what torch's own launchers use): `torch.cuda.current_stream()`
    builds a Stream object through four Python layers, ~3 us of the ~20 us a launch costs the host, once per entry point (163 per batch-1 frame)."""
    if _raw_stream is None:
        return torch.cuda.current_stream().cuda_stream
    return _raw_stream(_get_device())


def stream():
    """The HIP stream kernels are enqueued on = torch's current stream (as the reference plugins do with
    at::cuda::getCurrentCUDAStream(), bias_act.cpp:88-94)."""
    return _stream_handle()


def ptr(t):
    """Device pointer of `t` for a C-ABI call.  Every tensor whose pointer crosses the boundary is HELD until the launch that
    takes it has been enqueued (`check()` releases the list): a temporary passed straight to an entry point —
    `fn(ptr(a.contiguous()), ptr(b.contiguous()))` — would otherwise be freed as soon as its pointer is taken, and the NEXT
    temporary of the same argument list could be carved out of the same block and overwrite it before the kernel is even launched.
    After the enqueue the caching allocator's stream ordering protects the block (it is only re-used by later work of the same
    stream).  One mechanism instead of a hand-kept `x = x.contiguous()` discipline at every call site."""
    if t is None:
        return None
    try:
        _tls.held.append(t)
    except AttributeError:
        _tls.held = [t]
    return t.data_ptr()             # (a plain int: ctypes converts it for c_void_p arguments and struct fields; ~1500 calls per forward)


TICKET_COUNT = 4096


SLAB_FLOATS = 1 << 21        # 8 MB: the largest slab workspace a few-pixel launch asks for (256 workgroups x 16 KB x 2)


class _SeamPool:
    """Arrival counters + slab workspace of ONE stream (or one captured graph's stream): launches of a stream are serialised, so they share both."""

    def __init__(self, device):
        self.tickets = torch.zeros(TICKET_COUNT, dtype=torch.int32, device=device)
        self.slabs = torch.empty(SLAB_FLOATS, dtype=torch.float32, device=device)
        self.tickets_ptr, self.slabs_ptr = self.tickets.data_ptr(), self.slabs.data_ptr()


def seam_pool():
    """The current stream's arrival counters (n3d_conv2d_desc.tickets: TICKET_COUNT zeroed int32 words) and slab workspace.  Kernels that split K over
    workgroups and reduce in the last-arriving one (csrc/conv2d_sk_bf16x3.hip) count arrivals per output tile there and leave every word zero, so a
    pool is zeroed once, when it is created; launches of ONE stream are serialised and share it, different streams (the side stream of the static
    backbone, the bench's lanes) get their own.  A HIP graph replays on whatever stream its caller picks, possibly beside eager work of the streams
    it was captured on: `ticket_pools(...)` gives a capture its own pools, allocated eagerly BEFORE the capture begins (generator.synthesis_graph);
    a capture without them (not ours) allocates inside the capture — the zero fill is then a memset node of that graph."""
    h = _stream_handle()
    over = getattr(_tls, 'ticket_override', None)
    if over is None:
        t = _seam_pools.get(h)
        if t is not None:
            return t
        if not torch.cuda.is_current_stream_capturing():
            t = _seam_pools[h] = _SeamPool(torch.device('cuda', torch.cuda.current_device()))
            return t
    else:
        t = over.assigned.get(h)
        if t is None and over.free:
            t = over.assigned[h] = over.free.pop()
        if t is not None:
            return t
    return _SeamPool(torch.device('cuda', torch.cuda.current_device()))


def tickets():
    return seam_pool().tickets


_seam_pools = {}


class ticket_pools:
    """Context manager: launches inside take their arrival counters and slabs from `pools` (`new_ticket_pools`: one per stream the region
    launches on, handed out in order of first use) instead of the streams' own pools — what a captured graph needs (see `seam_pool`)."""

    def __init__(self, pools):
        self.free, self.assigned, self.all = list(pools), {}, list(pools)

    def __enter__(self):
        self.prev = getattr(_tls, 'ticket_override', None)
        _tls.ticket_override = self
        return self

    def __exit__(self, *exc):
        _tls.ticket_override = self.prev
        return False


def new_ticket_pools(device, count=3):
    return [_SeamPool(device) for _ in range(count)]


def require_device(*tensors):
    for t in tensors:
        if t is not None and t.device.type != 'cuda':
            raise RuntimeError('n3d ops run on a HIP device only (got a %s tensor); the CPU restatement lives in '
                               'oracle/ and is test infrastructure, not a fallback' % t.device.type)


class Split8:
    """A [N,C,H,W] activation in the split8 layout of include/n3d.h — bf16 [N][2 (hi,lo)][C/8][H][W][8], 4 bytes per element —
    produced by a kernel epilogue (n3d_fir4_split8) for the pre-split 3x3 convolution; the consumer's style modulation is
    already multiplied in.  `data` is the flat bf16 storage."""

    def __init__(self, n, c, h, w, device):
        import torch as _t
        assert c % 8 == 0
        self.shape = (n, c, h, w)
        self.data = _t.empty(n * 2 * (c // 8) * h * w * 8, dtype=_t.bfloat16, device=device)
        self.device = self.data.device

    def to_float(self):
        """float32 [N,C,H,W] = hi + lo (tests / debugging only)."""
        n, c, h, w = self.shape
        t = self.data.reshape(n, 2, c // 8, h, w, 8).float()
        return (t[:, 0] + t[:, 1]).permute(0, 1, 4, 2, 3).reshape(n, c, h, w)


class C8:
    """A float32 [N,C,H,W] activation in the channel-interleaved "c8" layout of include/n3d.h ([N][C/8][H][W][8]): what the
    transposed convolution writes for n3d_fir4_split8.  `data` is the dense float32 storage."""

    def __init__(self, n, c, h, w, device):
        import torch as _t
        assert c % 8 == 0
        self.shape = (n, c, h, w)
        self.data = _t.empty(n, c // 8, h, w, 8, dtype=_t.float32, device=device)
        self.device = self.data.device

    def to_nchw(self):
        n, c, h, w = self.shape
        return self.data.permute(0, 1, 4, 2, 3).reshape(n, c, h, w)


class H8:
    """A [N,C,H,W] activation of one of the reference's float16 blocks in the "h8" layout of include/n3d.h — float16
    [N][C/8][H][W][8], 16-byte units of 8 consecutive channels of one pixel, unmodulated (the per-sample float16 weights carry
    the styles: n3d_modulate_weights_f16).  `data` is the dense float16 storage."""

    def __init__(self, n, c, h, w, device):
        import torch as _t
        assert c % 8 == 0
        self.shape = (n, c, h, w)
        self.data = _t.empty(n, c // 8, h, w, 8, dtype=_t.float16, device=device)
        self.device = self.data.device

    def to_float(self):
        """float32 [N,C,H,W] (tests / debugging only)."""
        n, c, h, w = self.shape
        return self.data.permute(0, 1, 4, 2, 3).reshape(n, c, h, w).float()

    def sample(self, i):
        """Sample i as an H8 of batch 1 sharing this tensor's storage (a view: the layers that run sample by sample — per-sample random noise)."""
        v = object.__new__(H8)
        n, c, h, w = self.shape
        v.shape, v.data, v.device, v._src = (1, c, h, w), self.data[i:i + 1], self.device, self
        return v

    def to_nchw(self):
        """`x.to(torch.float32)`: dense float32 [N,C,H,W] on libn3d.so (n3d_cast_h8) — where a float16 block's feature map is read by
        float32 code (the StyleUNets' fusion layers concatenate it with the float32 encoder features)."""
        n, c, h, w = self.shape
        y = torch.empty(n, c, h, w, dtype=torch.float32, device=self.device)
        check(lib().n3d_cast_h8(ptr(self.data), ptr(y), n, c, h * w, 0, 0, stream()))
        return y

    @classmethod
    def from_nchw(cls, x):
        """`x.to(torch.float16)` of a float32 [N,C,H,W] tensor (dense planes, any batch stride) on libn3d.so (n3d_cast_h8)."""
        require_device(x)
        n, c, h, w = x.shape
        if x.dtype != _torch_f32() or x.stride()[1:] != (h * w, w, 1) or (x.stride(0) == 0 and n > 1):
            x = x.to(_torch_f32()).contiguous()      # (n3d_cast_h8 reads a batch stride of 0 as "dense": an expand()ed batch is materialised)
        y = cls(n, c, h, w, x.device)
        check(lib().n3d_cast_h8(ptr(x), ptr(y.data), n, c, h * w, x.stride(0), 1, stream()))
        y._src = x                               # the source stays referenced until the launch is enqueued behind later work
        return y


def _torch_f32():
    return torch.float32


def cast(t, dtype):
    """float16 <-> float32 conversion on libn3d.so (n3d_cast); a tensor that already has `dtype` is returned as is.  Any other
    dtype raises: the kernels behind the operator layer compute in float32 and store float32 or float16 only."""
    import torch as _t
    if t.dtype == dtype:
        return t
    codes = {_t.float32: 0, _t.float16: 1}
    if t.dtype not in codes or dtype not in codes:
        raise RuntimeError(f'n3d ops take float32 or float16 tensors (got {t.dtype} -> {dtype})')
    require_device(t)
    t = t.contiguous()
    y = _t.empty(t.shape, dtype=dtype, device=t.device)
    check(lib().n3d_cast(ptr(t), ptr(y), t.numel(), codes[t.dtype], codes[dtype], stream()))
    return y


_ACT_SPEC = None


def make_epilogue(row_scale=None, noise=None, noise_strength=None, bias=None, residual=None, const_scale=1.0,
                  act='linear', alpha=None, gain=None, clamp=None, residual_up_filter=None, round_f16=False):
    """`residual_up_filter` (a [4,4] filter): `residual` is the half-resolution image and is upsampled x2 in the epilogue
    (== upfirdn2d.upsample2d(residual, filter)) instead of being read at full resolution."""
    global _ACT_SPEC
    if _ACT_SPEC is None:
        from .torch_utils.ops.bias_act import activation_funcs
        _ACT_SPEC = {k: (ACT_IDS[k], float(v.def_alpha), float(v.def_gain)) for k, v in activation_funcs.items()}
    act_id, def_alpha, def_gain = _ACT_SPEC[act]
    e = Epilogue()
    # (pointers straight from the tensors — the struct keeps them alive itself, below — and only the fields that are set: ~120 epilogues per forward)
    if row_scale is not None:
        e.row_scale, e.row_scale_stride = row_scale.data_ptr(), row_scale.stride(0)
    if noise is not None:
        e.noise, e.noise_strength = noise.data_ptr(), noise_strength.data_ptr() if noise_strength is not None else None
    elif noise_strength is not None:
        e.noise_strength = noise_strength.data_ptr()
    if bias is not None:
        e.bias = bias.data_ptr()
    if residual is not None:
        e.residual, e.residual_batch_stride = residual.data_ptr(), residual.stride(0)
    e.const_scale = float(const_scale)
    e.act = act_id
    e.alpha = def_alpha if alpha is None else float(alpha)
    e.gain = def_gain if gain is None else float(gain)
    e.clamp = -1.0 if clamp is None else float(clamp)
    if residual_up_filter is not None:
        assert residual is not None and residual.is_contiguous() and tuple(residual_up_filter.shape) == (4, 4) and residual_up_filter.is_contiguous()
        e.residual_up_filter = residual_up_filter.data_ptr()
    if round_f16:
        e.round_f16 = int(round_f16)             # False / True, or 2 (float16 blocks: the reference's off-GPU bias_act rounding, include/n3d.h)
    e._keepalive = (row_scale, noise, noise_strength, bias, residual, residual_up_filter)   # the struct only holds raw pointers
    return e


def prof_enable(on=True):
    check(lib().n3d_prof_enable(1 if on else 0))


def prof_reset():
    check(lib().n3d_prof_reset())


def prof_read():
    """{family: dict(ms, launches, flops, bytes)} since the last reset (synchronises the recorded events)."""
    out = {}
    for fam, name in enumerate(FAMILY_NAMES):
        ms, n, fl, by = c_double(), c_int64(), c_double(), c_double()
        check(lib().n3d_prof_read(fam, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl), ctypes.byref(by)))
        out[name] = dict(ms=ms.value, launches=n.value, flops=fl.value, bytes=by.value)
    return out
