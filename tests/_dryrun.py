"""Recording stand-in for libn3d.so used by the CPU dry-run tests (tests/test_cpu_orchestration.py): marshals every argument as
ctypes would, re-checks the convolution entry points' preconditions, launches nothing.  `patches()` -> (list of (object,
attribute, replacement), list the recorder appends the entry-point names to)."""
import contextlib

import torch


class _Stream:
    cuda_stream = 0

    def wait_stream(self, s): pass

    def record_event(self): return None

    def wait_event(self, e): pass


def _check_conv_desc(name, d):
    bf16x3 = name == 'n3d_conv2d_bf16x3'
    assert d.N >= 0 and d.I > 0 and d.O > 0 and d.H > 0 and d.W > 0, (name, d.N, d.I, d.O, d.H, d.W)
    assert d.x and d.wt and (d.y or d.rgb_partial), name
    if d.rgb_partial:                                    # toRGB fused into a network's last 3x3 layer: pre-split stride-1 kernel, no split-K, <= 32 colours
        assert bf16x3 and d.x_layout == 1 and d.ksize == 3 and d.mode == 0 and d.ksplit <= 1 and d.rgb_weight and d.rgb_style and 1 <= d.rgb_channels <= 32
        assert not d.epi.residual and not d.epi.round_f16
    assert d.ksize in (1, 3) and 0 <= d.mode <= 2
    assert d.x_layout in (0, 1) and d.y_layout in (0, 1, 2)
    if d.y_layout == 1:                                  # split8 OUTPUT: the 1x1 split-bf16 kernel in front of an un-modulated 3x3 layer
        assert bf16x3 and d.ksize == 1 and d.x_layout == 0 and d.O % 32 == 0 and d.ksplit <= 1
    if d.x_layout == 1:                                  # split8 input: the pre-split launchers' preconditions (conv2d_ps_bf16x3.hip)
        assert bf16x3 and d.ksize == 3 and not d.style and d.epi.act in (1, 3)
        assert d.ksplit <= 1 or d.mode in (0, 1)         # stride 1 (the library's own factor, n3d_conv2d_split8_ksplit) and stride 2 split K
        assert d.I % 16 == 0 and d.x_batch_stride % 4 == 0
        assert (d.mode == 0 and d.H >= 16 and d.W >= 32) or (d.mode == 2 and d.y_layout in (0, 2) and d.O % 64 == 0 and d.H >= 4 and d.W >= 4) or (d.mode == 1 and d.H >= 3 and d.W >= 3)
    if d.side_split8 and d.rgb_partial:                  # fused-toRGB layer with a second reader: its own output * that reader's styles as split8
        assert d.side_style and d.O % 8 == 0
    elif d.side_split8:                                  # toRGB's second output: x * the next block's styles as split8
        assert bf16x3 and d.ksize == 1 and d.x_layout == 0 and d.y_layout == 0 and d.O <= 128 and d.I % 32 == 0 and d.side_style
    if bf16x3:
        assert d.I % 16 == 0 and (d.ksize == 3 or d.mode == 0)
        assert d.x_row_stride in (0, d.W) or (d.ksize == 3 and d.mode == 1)
        assert d.I * d.H * d.W * 4 < 2 ** 31
        assert not (d.epi.residual_up_filter and d.ksize == 3)
    ow = 2 * d.W + 1 if d.mode == 2 else ((d.W - 3) // 2 + 1 if (d.mode == 1 and d.ksize == 3) else d.W)
    assert d.y_row_stride == 0 or d.y_row_stride >= ow, (name, d.mode, d.W, d.y_row_stride)
    assert d.ksplit <= 1 or d.workspace, name
    if d.tickets:                                        # arrival counters: only the few-pixel 3x3 kernels count on them, and they need their slab workspace
        assert bf16x3 and d.ksize == 3 and d.x_layout == 0 and d.workspace and d.ticket_count > 0 and d.ksplit <= 1
    assert not d.epi.noise or d.epi.noise_strength
    assert 1 <= d.epi.act <= 9


def _check_conv_f16_desc(d):
    """n3d_conv2d_f16's preconditions (conv2d_f16.hip)."""
    assert d.x and d.wt and (d.y or d.rgb_partial) and not d.style and d.ksplit <= 1
    if d.rgb_partial:
        assert d.mode == 0 and d.rgb_weight and not d.rgb_style and 1 <= d.rgb_channels <= 4
    assert d.ksize == 3 and d.mode in (0, 2) and d.x_layout == 3 and d.y_layout == 3
    assert d.I >= 16 and d.I % 16 == 0 and d.O >= 64 and d.O % 64 == 0
    assert d.x_batch_stride == 0 and d.y_batch_stride == 0 and d.x_row_stride == 0 and d.y_row_stride == 0
    assert not d.epi.row_scale and d.epi.const_scale == 1.0 and not d.epi.residual
    if d.mode == 0:
        assert d.H >= 16 and d.W >= 32 and d.epi.act in (1, 3)
    else:
        assert d.H >= 4 and d.W >= 4 and d.epi.act == 1 and not d.epi.bias and not d.epi.noise and d.epi.clamp < 0 and d.epi.gain == 1.0


def patches():
    from next3d_amd import _lib, generator
    real = _lib.lib()                                            # the built library loads without a GPU
    calls = []

    class Recorder:
        def __getattr__(self, name):
            res, argtypes = _lib._SIGNATURES[name]
            if name in ('n3d_conv2d_bf16x3_blocks', 'n3d_conv2d_split8_eligible', 'n3d_conv2d_split8_ksplit', 'n3d_conv2d_sk_eligible', 'n3d_conv2d_sk_s2_eligible', 'n3d_conv2d_up_sk_eligible', 'n3d_conv2d_sk_workspace', 'n3d_abi_version', 'n3d_last_error'):
                return getattr(real, name)                       # pure host functions: the real ones

            def fn(*args):
                assert len(args) == len(argtypes), (name, len(args), len(argtypes))
                for a, t in zip(args, argtypes):
                    t.from_param(a)                              # raises exactly where a real ctypes call would
                if name in ('n3d_conv2d', 'n3d_conv2d_bf16x3'):
                    _check_conv_desc(name, getattr(args[0], '_obj', args[0]))
                if name == 'n3d_conv2d_f16':
                    _check_conv_f16_desc(getattr(args[0], '_obj', args[0]))
                calls.append(name)
                return 0
            return fn

    rec = Recorder()
    class _Pool:
        tickets, slabs = torch.zeros(_lib.TICKET_COUNT, dtype=torch.int32), torch.empty(16, dtype=torch.float32)
        tickets_ptr, slabs_ptr = tickets.data_ptr(), slabs.data_ptr()
    pool = _Pool()
    return [(_lib, 'lib', lambda: rec), (_lib, 'require_device', lambda *a: None), (_lib, 'stream', lambda: None), (_lib, 'seam_pool', lambda: pool),
            (generator, '_require_hip', lambda d: None), (torch.cuda, 'current_stream', lambda *a, **k: _Stream()),
            (torch.cuda, 'Stream', lambda *a, **k: _Stream()), (torch.cuda, 'stream', lambda s: contextlib.nullcontext())], calls
