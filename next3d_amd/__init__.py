"""next3d_amd — MI355X (gfx950) implementation of the Next3D generator-forward hot path.

    next3d_amd.torch_utils.ops.*      operator boundary (same names as the reference's torch_utils.ops)
    next3d_amd.generator              TriPlaneGenerator (same constructor / mapping / synthesis / forward)
    next3d_amd.csrc + include/n3d.h   HIP kernels behind a C ABI (libn3d.so), built by `python -m next3d_amd.build`
"""
import importlib
import sys

_OP_MODULES = ('bias_act', 'upfirdn2d', 'conv2d_resample', 'conv2d_gradfix', 'fma', 'filtered_lrelu')


_SHIM_MODULES = {'pytorch3d': ('pytorch3d', 'pytorch3d.io', 'pytorch3d.structures', 'pytorch3d.renderer', 'pytorch3d.renderer.mesh'),
                 'cv2': ('cv2',)}


def install_dropin(model=False, third_party='auto'):
    """Alias the reference's module paths to this package so that reference code — including code un-pickled from a
    network .pkl, whose imports are resolved at load time (torch_utils/persistence.py:218) — calls libn3d.so.

    Always aliases `torch_utils.ops.{bias_act,upfirdn2d,conv2d_resample,conv2d_gradfix,fma}`.  With model=True also
    aliases `training_avatar_texture.triplane_next3d` so `--reload_modules=True` constructs this TriPlaneGenerator.
    third_party: the two packages the reference's rasterisation step imports (`pytorch3d`: rasterize_meshes / Meshes /
    io.load_obj, `cv2`: floodFill / imread) are served by next3d_amd.shims on libn3d.so kernels — 'auto': only where the real
    package cannot be imported; True: always (a CPU-only PyTorch3D / OpenCV build would otherwise put host round trips back on
    the path); False: never."""
    if third_party:
        for pkg, names in _SHIM_MODULES.items():
            if third_party == 'auto':
                try:
                    importlib.import_module(pkg)
                    continue
                except ImportError:
                    pass
            for name in names:
                sys.modules[name] = importlib.import_module(f'{__name__}.shims.{name}')
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
