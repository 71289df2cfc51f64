#!/usr/bin/env python3
"""Writes the rasteriser's inputs for tools/probe/raster_coresidency_repro.cpp as raw little-endian arrays into ONE file
(tools/probe/raster_inputs.bin): header int32 [N, V, Lm, F, views, H, W, mask_h, mask_w], then verts f32 [N,V,3], lms f32 [N,Lm,3],
rot f32 [views,3,3], faces i32 [F,3], face_uv f32 [F,3,3], uv_mask f32 [mask_h,mask_w] — exactly what generator.raster_geometry passes
to n3d_rasterize_views for the demo mesh at batch 4.  (Build container or GPU box; the reproducer itself needs neither Python nor torch.)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from next3d_amd import demo, mesh  # noqa: E402
from next3d_amd.generator import RENDERING_VIEWS, angle2matrix  # noqa: E402

d = demo.demo_arrays()
mb = mesh.mesh_buffers(d['faces'], d['uvs'], d['uvfaces'])
faces = mb['faces'][0][:, [0, 2, 1]].to(torch.int32).contiguous().numpy()
face_uv = mb['face_uvcoords'][0][:, [0, 2, 1]].contiguous().numpy().astype(np.float32)
rot = torch.cat([angle2matrix(a) for a in RENDERING_VIEWS], 0).contiguous().numpy().astype(np.float32)
_, _, _, v = demo.demo_batch([0, 1, 2, 3])
v = v.numpy().astype(np.float32)
g = np.random.RandomState(7)
v[1:] += 0.0005 * g.randn(*v[1:].shape).astype(np.float32)             # four slightly different meshes
verts, lms = np.ascontiguousarray(v[:, :5023]), np.ascontiguousarray(v[:, 5023:])
mask = torch.nn.functional.interpolate(mesh.synthetic_uv_face_mask().float(), [256, 256])[0, 0].contiguous().numpy().astype(np.float32)
hdr = np.array([verts.shape[0], verts.shape[1], lms.shape[1], faces.shape[0], rot.shape[0], 256, 256, mask.shape[0], mask.shape[1]], dtype=np.int32)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'raster_inputs.bin')
with open(out, 'wb') as fh:
    for a in (hdr, verts, lms, rot, faces, face_uv, mask):
        fh.write(np.ascontiguousarray(a).tobytes())
print(out, os.path.getsize(out), 'bytes')
