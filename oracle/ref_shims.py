"""oracle/ref_shims.py — import shims that let the REAL reference Python (/root/reference) run on
CPU in the build container (TEST INFRASTRUCTURE; used only by oracle/pin_against_reference.py,
never at test/bench run time: /root/reference does not exist on the GPU box).

The reference imports third-party packages that are absent here:
pydantic.NoneStr, cv2, turtle(tkinter), torchvision, pytorch3d, mrcfile, imageio.  The stand-ins for
the two with arithmetic (pytorch3d.rasterize_meshes, cv2.floodFill) are this oracle's own C
restatements (oracle/raster_ref.c) — so those two boundaries stay PARITY UNPINNED.
"""
import os
import sys
import types

import numpy as np
import torch

REF = '/root/reference'


def install(uv_face_mask_np, third_party=True):
    """Register stub modules and put the reference on sys.path.  `uv_face_mask_np`: [H,W] float in [0,1]
    returned (as a 3-channel uint8 image) by the stubbed cv2.imread (triplane_next3d.py:91).
    third_party=False: leave `cv2` / `pytorch3d` alone (the caller registers next3d_amd.shims for them: the B1 dry run of
    tests/test_cpu_orchestration.py) and install only the import-time stubs (pydantic.NoneStr, turtle, torchvision, mrcfile, imageio)."""
    from . import raster
    if not third_party:
        import pydantic
        pydantic.__dict__['NoneStr'] = type(None)
        sys.modules.setdefault('turtle', types.ModuleType('turtle'))
        sys.modules['turtle'].update = lambda *a, **k: None
        tv = types.ModuleType('torchvision')
        tvu = types.ModuleType('torchvision.utils')
        tvu.save_image = lambda *a, **k: None
        tvt = types.ModuleType('torchvision.transforms')
        tv.utils, tv.transforms = tvu, tvt
        sys.modules.update({'torchvision': tv, 'torchvision.utils': tvu, 'torchvision.transforms': tvt})
        for name in ('mrcfile', 'imageio'):
            sys.modules.setdefault(name, types.ModuleType(name))
        if REF not in sys.path:
            sys.path.insert(0, REF)
        os.chdir(REF)
        return

    import pydantic
    pydantic.__dict__['NoneStr'] = type(None)      # pydantic 2 removed it; dnnlib/util.py:26 imports it

    cv2 = types.ModuleType('cv2')
    cv2.FLOODFILL_FIXED_RANGE = 1 << 16

    def imread(path, *a, **k):
        img = (np.clip(uv_face_mask_np, 0, 1) * 255.0).round().astype(np.uint8)
        return np.stack([img] * 3, -1)

    def floodFill(image, mask, seedPoint, newVal, loDiff=(0,), upDiff=(0,), flags=0):
        assert tuple(seedPoint) == (0, 0) and image.dtype == np.float32 and image.ndim == 2
        raster.floodfill_fixed_range(image, float(newVal[0]), float(loDiff[0]), float(upDiff[0]))
        return 0, image, mask, (0, 0, 0, 0)

    cv2.imread, cv2.floodFill = imread, floodFill
    cv2.line = cv2.circle = lambda img, *a, **k: img
    cv2.norm = lambda *a, **k: None                # `from cv2 import norm` (ray_marcher.py:16), unused
    sys.modules['cv2'] = cv2

    sys.modules.setdefault('turtle', types.ModuleType('turtle'))
    sys.modules['turtle'].update = lambda *a, **k: None
    tv = types.ModuleType('torchvision')
    tvu = types.ModuleType('torchvision.utils')
    tvu.save_image = lambda *a, **k: None
    tvt = types.ModuleType('torchvision.transforms')
    tv.utils, tv.transforms = tvu, tvt
    sys.modules.update({'torchvision': tv, 'torchvision.utils': tvu, 'torchvision.transforms': tvt})

    p3d = types.ModuleType('pytorch3d')
    p3d_io = types.ModuleType('pytorch3d.io')
    p3d_st = types.ModuleType('pytorch3d.structures')
    p3d_r = types.ModuleType('pytorch3d.renderer')
    p3d_rm = types.ModuleType('pytorch3d.renderer.mesh')

    def load_obj(path, *a, **k):
        v, fv, vt, ft = raster.load_obj(path)
        return v, types.SimpleNamespace(verts_idx=fv, textures_idx=ft), types.SimpleNamespace(verts_uvs=vt)

    class Meshes:
        def __init__(self, verts, faces):
            self.verts, self.faces = verts, faces

    def rasterize_meshes(meshes, image_size, blur_radius, faces_per_pixel, bin_size, max_faces_per_bin,
                         perspective_correct, cull_backfaces, **k):
        assert blur_radius == 0.0 and faces_per_pixel == 1 and not perspective_correct
        size = image_size[0] if isinstance(image_size, (list, tuple)) else image_size
        faces = meshes.faces
        assert bool((faces == faces[:1]).all()), "stand-in assumes shared topology across the batch"
        p2f, zbuf, bary = raster.rasterize_meshes(meshes.verts, faces[0], image_size=size,
                                                  cull_backfaces=cull_backfaces)
        return p2f, zbuf, bary, torch.zeros_like(zbuf)

    p3d_io.load_obj, p3d_st.Meshes, p3d_rm.rasterize_meshes = load_obj, Meshes, rasterize_meshes
    p3d.io, p3d.structures, p3d.renderer, p3d_r.mesh = p3d_io, p3d_st, p3d_r, p3d_rm
    sys.modules.update({'pytorch3d': p3d, 'pytorch3d.io': p3d_io, 'pytorch3d.structures': p3d_st,
                        'pytorch3d.renderer': p3d_r, 'pytorch3d.renderer.mesh': p3d_rm})
    for name in ('mrcfile', 'imageio'):
        sys.modules.setdefault(name, types.ModuleType(name))

    if REF not in sys.path:
        sys.path.insert(0, REF)
    os.chdir(REF)  # TriPlaneGenerator.__init__ opens data/ffhq/uv_face_eye_mask.png by relative path


def build_reference_generator(rendering_kwargs, topology_path='data/demo/demo.obj', num_fp16_res=0, conv_clamp=None, channel_base=32768, channel_max=512, mapping_kwargs=None):
    """Construct the reference TriPlaneGenerator with the kwargs train_next3d.py would pass
    (train_next3d.py:250-411)."""
    from training_avatar_texture.triplane_next3d import TriPlaneGenerator
    rk = dict(rendering_kwargs)
    rk.setdefault('superresolution_module', 'training_avatar_texture.superresolution.SuperresolutionHybrid8XDC')
    img_resolution = {'8XDC': 512, '8X': 512, '4X': 256, '2X': 128}[rk['superresolution_module'].rsplit('Hybrid', 1)[-1]]      # every module asserts its own
    G = TriPlaneGenerator(
        z_dim=512, c_dim=25, w_dim=512, img_resolution=img_resolution, img_channels=3, topology_path=topology_path,
        sr_num_fp16_res=4, mapping_kwargs=dict(num_layers=2) if mapping_kwargs is None else mapping_kwargs, rendering_kwargs=rk,
        sr_kwargs=dict(channel_base=32768, channel_max=512, fused_modconv_default='inference_only'),
        channel_base=channel_base, channel_max=channel_max, fused_modconv_default='inference_only', num_fp16_res=num_fp16_res,
        conv_clamp=conv_clamp)        # (num_fp16_res = 4, conv_clamp = 256: what legacy.load_network_pkl(force_fp16=True) rebuilds the model with)
    return G.eval().requires_grad_(False)
