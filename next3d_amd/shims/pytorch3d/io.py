"""pytorch3d.io.load_obj for the reference's call `verts, faces, aux = load_obj(topology_path)` (triplane_next3d.py:79-82): it reads
`faces.verts_idx`, `faces.textures_idx` and `aux.verts_uvs`.  A host-side text parse (next3d_amd.mesh.parse_obj); materials,
normals and texture images are not loaded."""
import collections

from ... import mesh as _mesh

Faces = collections.namedtuple('Faces', 'verts_idx normals_idx textures_idx materials_idx')
Properties = collections.namedtuple('Properties', 'normals verts_uvs material_colors texture_images texture_atlas')


def load_obj(f, load_textures=True, create_texture_atlas=False, texture_atlas_size=4, texture_wrap='repeat', device='cpu', path_manager=None):
    if create_texture_atlas:
        raise RuntimeError('pytorch3d shim: load_obj(create_texture_atlas=True) is outside the generator-forward path')
    verts, fv, uvs, ft = _mesh.parse_obj(f)
    verts, fv, uvs, ft = (t.to(device) for t in (verts, fv, uvs, ft))
    return verts, Faces(fv, None, ft, None), Properties(None, uvs, None, None, None)
