"""pytorch3d.renderer.mesh.rasterize_meshes with the reference's settings (volumetric_rendering/renderer.py:389-397, call :415-424:
blur_radius 0, faces_per_pixel 1, perspective_correct False, cull_backfaces True, bin_size None) on libn3d.so
(n3d_rasterize_meshes: the face-parallel z-buffer kernel of n3d_rasterize_views)."""
import torch

from ..... import _lib


def rasterize_meshes(meshes, image_size=256, blur_radius=0.0, faces_per_pixel=8, bin_size=None, max_faces_per_bin=None,
                     perspective_correct=False, clip_barycentric_coords=False, cull_backfaces=False, z_clip_value=None,
                     cull_to_frustum=False):
    """-> (pix_to_face [N,H,W,1] int64 packed n*F+f / -1, zbuf [N,H,W,1], bary_coords [N,H,W,1,3], dists [N,H,W,1]).
    `dists` (signed squared pixel-to-edge distance) is not computed: the reference discards it (renderer.py:415); it is returned
    as -1 everywhere so that a consumer cannot mistake it for data."""
    if blur_radius != 0 or faces_per_pixel != 1 or perspective_correct or clip_barycentric_coords or z_clip_value is not None or cull_to_frustum:
        raise RuntimeError('pytorch3d shim: rasterize_meshes is implemented for the settings of the Next3D generator forward only '
                           '(blur_radius=0, faces_per_pixel=1, perspective_correct=False, no clipping options)')
    size = image_size if isinstance(image_size, int) else image_size[0]
    if not isinstance(image_size, int) and image_size[0] != image_size[1]:
        raise RuntimeError('pytorch3d shim: square images only (the reference renders 256 x 256)')
    verts, faces = meshes.verts_padded(), meshes.faces_padded()
    _lib.require_device(verts, faces)
    n, v = verts.shape[:2]
    f = faces.shape[1]
    verts = verts.to(torch.float32).contiguous()
    shared = faces.stride(0) == 0                                   # `self.faces.expand(batch_size, -1, -1)`: one table for the batch
    faces32 = (faces[0] if shared else faces).to(torch.int32).contiguous()
    dev = verts.device
    zws = torch.empty(n * size * size, dtype=torch.int64, device=dev)
    p2f = torch.empty(n, size, size, 1, dtype=torch.int64, device=dev)
    zbuf = torch.empty(n, size, size, 1, dtype=torch.float32, device=dev)
    bary = torch.empty(n, size, size, 1, 3, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().n3d_rasterize_meshes(_lib.ptr(verts), _lib.ptr(faces32), 0 if shared else 3 * f, _lib.ptr(zws), _lib.ptr(p2f), _lib.ptr(zbuf),
                                               _lib.ptr(bary), n, v, f, size, size, 1 if cull_backfaces else 0, _lib.stream()))
    p2f._keep = (verts, faces32, zws)
    return p2f, zbuf, bary, torch.full_like(zbuf, -1.0)
