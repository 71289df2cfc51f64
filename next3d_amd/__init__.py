"""next3d_amd — MI355X (gfx950) implementation of the Next3D generator-forward hot path.

    next3d_amd.torch_utils.ops.*      operator boundary (same names as the reference's torch_utils.ops)
    next3d_amd.generator              TriPlaneGenerator (same constructor / mapping / synthesis / forward)
    next3d_amd.csrc + include/n3d.h   HIP kernels behind a C ABI (libn3d.so), built by `python -m next3d_amd.build`
"""
import importlib
import sys

_OP_MODULES = ('bias_act', 'upfirdn2d', 'conv2d_resample', 'conv2d_gradfix', 'fma', 'filtered_lrelu')


def install_dropin(model=False):
    """Alias the reference's module paths to this package so that reference code — including code un-pickled from a
    network .pkl, whose imports are resolved at load time (torch_utils/persistence.py:218) — calls libn3d.so.

    Always aliases `torch_utils.ops.{bias_act,upfirdn2d,conv2d_resample,conv2d_gradfix,fma}`.  With model=True also
    aliases `training_avatar_texture.triplane_next3d` so `--reload_modules=True` constructs this TriPlaneGenerator."""
    for parent in ('torch_utils', 'torch_utils.ops'):        # keep the reference's own packages when they are importable
        try:
            importlib.import_module(parent)
        except ImportError:
            sys.modules[parent] = importlib.import_module(f'{__name__}.{parent}')
    for name in _OP_MODULES:
        mod = importlib.import_module(f'{__name__}.torch_utils.ops.{name}')
        sys.modules['torch_utils.ops.' + name] = mod
        setattr(sys.modules['torch_utils.ops'], name, mod)
    if model:
        try:
            parent = importlib.import_module('training_avatar_texture')
        except ImportError:                                     # stand-alone use (no reference tree on sys.path): an empty parent package
            import types
            parent = types.ModuleType('training_avatar_texture')
            parent.__path__ = []
            sys.modules['training_avatar_texture'] = parent
        gen = importlib.import_module(f'{__name__}.generator')
        sys.modules['training_avatar_texture.triplane_next3d'] = gen
        parent.triplane_next3d = gen
