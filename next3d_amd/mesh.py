"""FLAME topology buffers and text-format mesh inputs for the generator.

Mirrors what TriPlaneGenerator.__init__ derives from `topology_path`
(reference training_avatar_texture/triplane_next3d.py:79-103) and the `.obj` / landmark parsing the
inference scripts do per frame (gen_samples_next3d.py:165-179, reenact_avatar_next3d.py:139-154).
"""
import numpy as np
import torch


def parse_obj(path):
    """Wavefront text -> (verts [V,3] f32, vert-index faces [F,3] i64, uvs [VT,2] f32, uv-index faces [F,3] i64).
    Only `v`, `vt` and triangular `f a/b[/c]` records are used, as in pytorch3d.io.load_obj's output
    consumed at triplane_next3d.py:79-82."""
    verts, uvs, fv, ft = [], [], [], []
    with open(path, 'r') as fh:
        for line in fh:
            if line.startswith('v '):
                verts.append([float(t) for t in line.split()[1:4]])
            elif line.startswith('vt '):
                uvs.append([float(t) for t in line.split()[1:3]])
            elif line.startswith('f '):
                corners = [c.split('/') for c in line.split()[1:4]]
                fv.append([int(c[0]) - 1 for c in corners])
                ft.append([int(c[1]) - 1 if len(c) > 1 and c[1] else 0 for c in corners])
    as_t = lambda a, dt: torch.tensor(a, dtype=dt)
    return as_t(verts, torch.float32), as_t(fv, torch.int64), as_t(uvs, torch.float32), as_t(ft, torch.int64)


def parse_obj_vertices(path):
    """`v ` records only (the per-frame driving mesh), [1,V,3] float32 — gen_samples_next3d.py:165-175."""
    rows = []
    with open(path, 'r') as fh:
        for line in fh:
            if line[:2] == 'v ':
                rows.append([float(t) for t in line.split()[1:]])
    return torch.from_numpy(np.asarray(rows, dtype=np.float64).reshape(-1, 3)).float().unsqueeze(0)


def parse_landmarks(path):
    """68x3 landmark text file -> [1,68,3] float32 — gen_samples_next3d.py:177-179."""
    return torch.from_numpy(np.loadtxt(path)).float().unsqueeze(0)


def dense_triangles(h, w, margin_x=2, margin_y=5):
    """Regular grid triangulation kept only as the `dense_faces` buffer (unused by the forward path);
    reference volumetric_rendering/renderer.py:463-479."""
    xs = np.arange(margin_x, w - 1 - margin_x)
    ys = np.arange(margin_y, h - 1 - margin_y)
    x, y = np.meshgrid(xs, ys, indexing='ij')
    x, y = x.reshape(-1), y.reshape(-1)
    t0 = np.stack([y * w + x, (y + 1) * w + x, y * w + x + 1], 1)
    t1 = np.stack([y * w + x + 1, (y + 1) * w + x, (y + 1) * w + x + 1], 1)
    return np.stack([t0, t1], 1).reshape(-1, 3)


def mesh_buffers_from_obj(topology_path, uv_resolution=256):
    _, faces, uvs, uvfaces = parse_obj(topology_path)
    return mesh_buffers(faces, uvs, uvfaces, uv_resolution)


def mesh_buffers(faces, uvs, uvfaces, uv_resolution=256):
    """The six registered mesh buffers (triplane_next3d.py:86-103) from topology arrays:
    faces [F,3] i64 (vertex indices), uvs [VT,2] f32, uvfaces [F,3] i64 (uv indices)."""
    faces, uvs, uvfaces = torch.as_tensor(faces).long(), torch.as_tensor(uvs).float(), torch.as_tensor(uvfaces).long()
    raw_uv = uvs[None]
    uvc = torch.cat([raw_uv, raw_uv[:, :, 0:1] * 0. + 1.], -1)
    uvc = uvc * 2 - 1
    uvc[..., 1] = -uvc[..., 1]
    face_uv = uvc[0][uvfaces][None]                      # face_vertices(uvcoords, uvfaces)
    return {
        'dense_faces': torch.from_numpy(dense_triangles(uv_resolution, uv_resolution)).long()[None].contiguous(),
        'faces': faces[None],
        'raw_uvcoords': raw_uv,
        'uvcoords': uvc,
        'uvfaces': uvfaces[None],
        'face_uvcoords': face_uv,
    }


def synthetic_uv_face_mask(res=256):
    """Stand-in for data/ffhq/uv_face_eye_mask.png (absent from the reference tree, triplane_next3d.py:91):
    ones everywhere except three elliptical holes, so that `fill_mouth` has interior holes to fill.
    Quantised to 8 bits like an image file.  Returns [1,1,res,res] float32."""
    yy, xx = np.meshgrid(np.arange(res, dtype=np.float32), np.arange(res, dtype=np.float32), indexing='ij')
    m = np.ones((res, res), np.float32)
    for cy, cx, ry, rx in ((0.42, 0.38, 0.035, 0.06), (0.42, 0.62, 0.035, 0.06), (0.66, 0.50, 0.05, 0.10)):
        d = ((yy / res - cy) / ry) ** 2 + ((xx / res - cx) / rx) ** 2
        m = np.minimum(m, np.clip((d - 0.8) / 0.4, 0.0, 1.0))       # soft edge -> fractional alphas
    m = np.round(m * 255.0) / 255.0
    return torch.from_numpy(m.astype(np.float32))[None, None].contiguous()
