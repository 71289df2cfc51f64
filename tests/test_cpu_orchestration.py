"""CPU (-m "not gpu") dry run of the HOST side of the path: the generator's whole launch sequence is driven against a recording
stand-in for libn3d.so that marshals every argument exactly as ctypes would (so a wrong type, count or struct is caught) and
re-checks the preconditions the C entry points enforce (N3D_CHECK in conv2d_bf16x3.hip / conv2d.hip) — but launches nothing.
It covers configurations the GPU tests do not run: batch 8 per GPU (BASELINE.json configs[3]), batch 3, a 128² neural render
(no resize in front of the super-resolution, superresolution.py:282), coarse-only sampling, the cached call patterns.
No arithmetic happens here (outputs are uninitialised memory); numerics are the GPU tests' business."""
import contextlib
import ctypes
from collections import Counter

import pytest
import torch


class _Stream:
    cuda_stream = 0

    def wait_stream(self, s): pass

    def record_event(self): return None

    def wait_event(self, e): pass


def _check_conv_desc(name, d):
    bf16x3 = name == 'n3d_conv2d_bf16x3'
    assert d.N >= 0 and d.I > 0 and d.O > 0 and d.H > 0 and d.W > 0, (name, d.N, d.I, d.O, d.H, d.W)
    assert d.x and d.wt and d.y, name
    assert d.ksize in (1, 3) and 0 <= d.mode <= 2
    if bf16x3:
        assert d.I % 16 == 0 and (d.ksize == 3 or d.mode == 0)
        assert d.x_row_stride in (0, d.W) or (d.ksize == 3 and d.mode == 1)
        assert d.I * d.H * d.W * 4 < 2 ** 31
        assert not (d.epi.residual_up_filter and d.ksize == 3)
    ow = 2 * d.W + 1 if d.mode == 2 else ((d.W - 3) // 2 + 1 if (d.mode == 1 and d.ksize == 3) else d.W)
    assert d.y_row_stride == 0 or d.y_row_stride >= ow, (name, d.mode, d.W, d.y_row_stride)
    assert d.ksplit <= 1 or d.workspace, name
    assert not d.epi.noise or d.epi.noise_strength
    assert 1 <= d.epi.act <= 9


@pytest.fixture
def dry(monkeypatch):
    from next3d_amd import _lib, generator
    real = _lib.lib()                                            # the built library loads without a GPU
    calls = []

    class Recorder:
        def __getattr__(self, name):
            res, argtypes = _lib._SIGNATURES[name]
            if name in ('n3d_conv2d_bf16x3_blocks', 'n3d_abi_version', 'n3d_last_error'):
                return getattr(real, name)                       # pure host functions: the real ones

            def fn(*args):
                assert len(args) == len(argtypes), (name, len(args), len(argtypes))
                for a, t in zip(args, argtypes):
                    t.from_param(a)                              # raises exactly where a real ctypes call would
                if name in ('n3d_conv2d', 'n3d_conv2d_bf16x3'):
                    _check_conv_desc(name, getattr(args[0], "_obj", args[0]))
                calls.append(name)
                return 0
            return fn

    rec = Recorder()
    monkeypatch.setattr(_lib, 'lib', lambda: rec)
    monkeypatch.setattr(_lib, 'require_device', lambda *a: None)
    monkeypatch.setattr(_lib, 'stream', lambda: None)
    monkeypatch.setattr(generator, '_require_hip', lambda d: None)
    monkeypatch.setattr(torch.cuda, 'current_stream', lambda *a, **k: _Stream())
    monkeypatch.setattr(torch.cuda, 'Stream', lambda *a, **k: _Stream())
    monkeypatch.setattr(torch.cuda, 'stream', lambda s: contextlib.nullcontext())
    return calls


@pytest.mark.parametrize('N,R,Sc,Sf', [(1, 32, 24, 24), (3, 64, 48, 0), (8, 64, 48, 48), (2, 128, 96, 96)])
def test_generator_launch_sequence_dry_run(dry, N, R, Sc, Sf):
    from next3d_amd import demo
    rk = dict(demo.RENDERING_KWARGS, depth_resolution=Sc, depth_resolution_importance=Sf)
    G, _ = demo.build_generator(torch.device('cpu'), rendering_kwargs=rk)
    G.overlap_static = False
    z, c, c_cond, v = demo.demo_batch(list(range(N)))
    kw = dict(neural_rendering_resolution=R, noise_mode='const')
    ws = G.mapping(z, c_cond, truncation_psi=0.7, truncation_cutoff=14)
    assert tuple(ws.shape) == (N, 28, 512)
    G.synthesis(ws, c, v, **kw)                                  # first call: also prepares the weights
    dry.clear()
    out = G.synthesis(ws, c, v, cache_backbone=True, cache_identity=True, **kw)
    full = Counter(dry)
    assert tuple(out['image'].shape) == (N, 3, 512, 512) and tuple(out['image_raw'].shape) == (N, 3, R, R)
    assert tuple(out['image_depth'].shape) == (N, 1, R, R)
    assert full['n3d_conv2d_prep_weight'] == 0 and full['n3d_conv2d_prep_weight_bf16x3'] == 0      # prepared once per model
    assert full['n3d_rasterize_views'] == 1 and full['n3d_render_rays'] == 1 and full['n3d_texture_project_planes'] == 1
    assert full['n3d_resize_aa'] == (2 if R == 128 else 4)       # mouth crop + paste (+ feature / rgb resize unless R == 128)
    n_full = sum(full.values())
    # the launch count does not depend on the batch: one launch per layer, whatever N
    assert n_full == 153 + (0 if R == 128 else 2), full
    dry.clear()
    G.synthesis(ws, c, v, use_cached_backbone=True, **kw)        # camera orbit: renderer + super-resolution only
    orbit = Counter(dry)
    assert orbit['n3d_rasterize_views'] == 0 and orbit['n3d_render_rays'] == 1 and sum(orbit.values()) < n_full // 4
    dry.clear()
    G.synthesis(ws, c, v, use_cached_identity=True, **kw)        # reenactment: no texture / static backbone
    reenact = Counter(dry)
    assert reenact['n3d_rasterize_views'] == 1 and sum(orbit.values()) < sum(reenact.values()) < n_full
    dry.clear()
    pts = torch.zeros(N, 1000, 3)
    smp = G.sample_mixed(pts, None, ws, v, noise_mode='const', use_cached_backbone=True)
    assert tuple(smp['rgb'].shape) == (N, 1000, 32) and tuple(smp['sigma'].shape) == (N, 1000, 1) and dry == ['n3d_sample_points']
    with pytest.raises(RuntimeError):
        G.synthesis(ws, c, v, neural_rendering_resolution=R)     # noise_mode defaults to 'random' (training only)
