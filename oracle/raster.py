"""oracle/raster.py — CPU restatement of the FLAME-mesh rasterisation stage (TEST INFRASTRUCTURE).

Wraps oracle/raster_ref.c (third-party PyTorch3D rasterizer + OpenCV floodFill restated in C,
PARITY UNPINNED — see that file's header) and restates the reference's own Python around
them.  Paths relative to /root/reference; `vr/` = training_avatar_texture/volumetric_rendering/.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'liboracle_raster.so')
_lib = None


def build(force=False):
    src = os.path.join(_HERE, 'raster_ref.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['gcc', '-O2', '-ffp-contract=off', '-shared', '-fPIC', '-o', _SO, src, '-lm'])
    return _SO


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.oracle_rasterize_meshes.restype = None
        _lib.oracle_floodfill_fixed_range.restype = None
    return _lib


def load_obj(path):
    """Stand-in for pytorch3d.io.load_obj (training_avatar_texture/triplane_next3d.py:79-82):
    returns verts [V,3] f32, faces.verts_idx [F,3] i64, verts_uvs [VT,2] f32, faces.textures_idx [F,3] i64."""
    v, vt, fv, ft = [], [], [], []
    with open(path) as fh:
        for line in fh:
            t = line.split()
            if not t:
                continue
            if t[0] == 'v':
                v.append([float(a) for a in t[1:4]])
            elif t[0] == 'vt':
                vt.append([float(a) for a in t[1:3]])
            elif t[0] == 'f':
                idx = [p.split('/') for p in t[1:4]]
                fv.append([int(p[0]) - 1 for p in idx])
                ft.append([int(p[1]) - 1 for p in idx])
    return (torch.tensor(v, dtype=torch.float32), torch.tensor(fv, dtype=torch.int64),
            torch.tensor(vt, dtype=torch.float32), torch.tensor(ft, dtype=torch.int64))


def rasterize_meshes(verts_ndc, faces, image_size=256, cull_backfaces=True):
    """PyTorch3D rasterize_meshes restatement.  verts_ndc [N,V,3] (PyTorch3D NDC), faces [F,3]
    -> pix_to_face [N,H,W,1] i64 (packed n*F+f, -1 empty), zbuf [N,H,W,1], bary [N,H,W,1,3]."""
    lib = _load()
    v = np.ascontiguousarray(verts_ndc.detach().cpu().numpy(), dtype=np.float32)
    f = np.ascontiguousarray(faces.detach().cpu().numpy(), dtype=np.int32)
    N, V, _ = v.shape
    F_ = f.shape[0]
    H = W = int(image_size)
    p2f = np.empty((N, H, W), dtype=np.int64)
    zb = np.empty((N, H, W), dtype=np.float32)
    bary = np.empty((N, H, W, 3), dtype=np.float32)
    lib.oracle_rasterize_meshes(v.ctypes.data_as(ctypes.c_void_p), f.ctypes.data_as(ctypes.c_void_p),
                                ctypes.c_int(N), ctypes.c_int(V), ctypes.c_int(F_), ctypes.c_int(H),
                                ctypes.c_int(W), ctypes.c_int(1 if cull_backfaces else 0),
                                p2f.ctypes.data_as(ctypes.c_void_p), zb.ctypes.data_as(ctypes.c_void_p),
                                bary.ctypes.data_as(ctypes.c_void_p))
    return (torch.from_numpy(p2f).unsqueeze(-1), torch.from_numpy(zb).unsqueeze(-1),
            torch.from_numpy(bary).unsqueeze(-2))


def floodfill_fixed_range(img, new_val=255.0, lo=0.0, up=254.0):
    """cv2.floodFill(img, mask, (0,0), new_val, lo, up, FLOODFILL_FIXED_RANGE) on a float32 [H,W] array, in place."""
    lib = _load()
    assert img.dtype == np.float32 and img.flags['C_CONTIGUOUS'] and img.ndim == 2
    lib.oracle_floodfill_fixed_range(img.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(img.shape[0]),
                                     ctypes.c_int(img.shape[1]), ctypes.c_float(new_val), ctypes.c_float(lo),
                                     ctypes.c_float(up))
    return img


def fill_mouth(images):
    """vr/renderer.py:583-602 (fill_mouth): regions with alpha*255 <= 254 connected to the border keep
    their alpha, everything else (interior holes, saturated pixels) becomes 1."""
    out = []
    for image in images:
        img = np.ascontiguousarray((image[0].cpu().numpy() * 255.).astype(np.float32))
        floodfill_fixed_range(img)
        out.append((torch.from_numpy(img).to(torch.float32) / 127.5 - 1).unsqueeze(0))
    mm = torch.stack(out, 0)
    mm = ((mm * 2. - 1.) * -1. + 1.) / 2.
    return (images + mm).clip(0, 1)


def angle2matrix(angles_deg):
    """vr/renderer.py:518-547 (angle2matrix), one [3] triple of degrees -> [1,3,3]."""
    a = torch.tensor(angles_deg, dtype=torch.float32).reshape(1, -1) * (np.pi) / 180.
    s, c = torch.sin(a), torch.cos(a)
    cx, cy, cz = c[:, 0], c[:, 1], c[:, 2]
    sx, sy, sz = s[:, 0], s[:, 1], s[:, 2]
    R = torch.stack([cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx,
                     sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx,
                     -sy, cy * sx, cy * cx], dim=0)
    return torch.reshape(R, (-1, 3, 3))


def orth_project(pts, tform, orth_shift, orth_scale):
    """training_avatar_texture/triplane_next3d.py:194-205 + vr/renderer.py:505-515 (batch_orth_proj, cam=[1,0,0])."""
    p = pts.clone()
    p[..., 1] *= -1
    t = (torch.bmm(p, tform.expand(p.shape[0], -1, -1)) + orth_shift) * orth_scale
    cam = torch.tensor([1., 0, 0]).view(-1, 1, 3)
    t = torch.cat([t[:, :, :2] + cam[:, :, 1:], t[:, :, 2:]], 2) * cam[:, :, 0:1]
    t[:, :, 1:] = -t[:, :, 1:]
    return t


def pytorch3d_rasterizer(vertices, faces, face_attrs, image_size=256):
    """vr/renderer.py:401-440 (Pytorch3dRasterizer.forward) -> [N, D+1, H, W] (attrs..., vismask)."""
    fixed = vertices.clone()
    fixed[..., :2] = -fixed[..., :2]
    p2f, _, bary = rasterize_meshes(fixed.float(), faces, image_size=image_size)
    vis = (p2f > -1).float()
    N, H, W, K, _ = bary.shape
    D = face_attrs.shape[-1]
    attrs = face_attrs.expand(N, -1, -1, -1).reshape(-1, 3, D)       # packed over the batch
    mask = p2f == -1
    idx = p2f.clone()
    idx[mask] = 0
    vals = attrs[idx.view(-1)].view(N, H, W, K, 3, D)
    pix = (bary[..., None] * vals).sum(dim=-2)
    pix[mask] = 0
    pix = pix[:, :, :, 0].permute(0, 3, 1, 2)
    return torch.cat([pix, vis[:, :, :, 0][:, None]], dim=1)


def gen_mouth_mask(lms2d):
    """training_avatar_texture/triplane_next3d.py:330-344 (gen_mouth_mask) -> int [N,4] (y0,y1,x0,x1)."""
    lm = lms2d.clone().cpu().numpy()
    lm[..., 0] = lm[..., 0] * 128 + 128
    lm[..., 1] = lm[..., 1] * 128 + 128
    outer = lm[:, 48:60]
    avg = (outer[:, 0] + outer[:, 6]) * 0.5
    ups, bottoms = np.max(outer[..., 0], axis=1, keepdims=True), np.min(outer[..., 0], axis=1, keepdims=True)
    lefts, rights = np.min(outer[..., 1], axis=1, keepdims=True), np.max(outer[..., 1], axis=1, keepdims=True)
    res = np.max(np.concatenate((ups - bottoms, rights - lefts), axis=1), axis=1, keepdims=True) * 1.2
    res = res.astype(int)
    return np.concatenate([(avg[:, 1:] - res // 2).astype(int), (avg[:, 1:] + res // 2).astype(int),
                           (avg[:, 0:1] - res // 2).astype(int), (avg[:, 0:1] + res // 2).astype(int)], 1)
