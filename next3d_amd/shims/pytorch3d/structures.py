"""pytorch3d.structures.Meshes as the container the reference builds at volumetric_rendering/renderer.py:414
(`Meshes(verts=fixed_vertices.float(), faces=faces.long())`) for rasterize_meshes: batched tensors of one topology size."""
import torch


class Meshes:
    def __init__(self, verts=None, faces=None, textures=None, **kw):
        if isinstance(verts, (list, tuple)):
            verts = torch.stack(list(verts), 0)
        if isinstance(faces, (list, tuple)):
            faces = torch.stack(list(faces), 0)
        if verts.ndim != 3 or verts.shape[-1] != 3 or faces.ndim != 3 or faces.shape[-1] != 3 or faces.shape[0] != verts.shape[0]:
            raise RuntimeError('pytorch3d shim: Meshes takes verts [N,V,3] and faces [N,F,3] (one topology size per batch)')
        self._verts, self._faces, self.textures = verts, faces, textures
        self.device = verts.device

    def __len__(self):
        return self._verts.shape[0]

    def verts_padded(self):
        return self._verts

    def faces_padded(self):
        return self._faces

    def verts_packed(self):
        return self._verts.reshape(-1, 3)

    def faces_packed(self):
        n, v = self._verts.shape[:2]
        off = torch.arange(n, device=self._faces.device).reshape(n, 1, 1) * v
        return (self._faces + off).reshape(-1, 3)
