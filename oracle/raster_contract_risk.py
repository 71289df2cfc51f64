#!/usr/bin/env python3
"""oracle/raster_contract_risk.py — bounds the parity risk of the two UNPINNED third-party boundaries (TEST INFRASTRUCTURE).

rows a13 / a15 of SURVEY.md §8: `pytorch3d...rasterize_meshes` (vr/renderer.py:415-424) and `cv2.floodFill` (:593) are in neither
the reference tree nor this container, so oracle/raster_ref.c restates their published algorithms and the HIP rasteriser is
bit-exact against THAT.  What a real PyTorch3D build may do differently at the floating-point level is fuse multiply-adds: nvcc
compiles rasterize_meshes.cu with -fmad=true, so its edge functions / barycentrics are FMA-contracted, while raster_ref.c (and
raster.hip) are built with contraction off.  This script measures how much that freedom can move the result:

  * builds raster_ref.c twice — `-ffp-contract=off` (the pinned restatement) and `-ffp-contract=fast -mfma` (every a*b+c the
    compiler can see becomes one FMA, the analogue of nvcc's default);
  * rasterises the four orthographic views of every golden mesh and of 100 seeded perturbations of the demo mesh (sigma 0.25 ..
    2 mm) with both, through the reference's own transforms (oracle.generator.rasterize);
  * counts pixels whose winning face / coverage / hole-filled alpha differ and the largest change of the projected neural-texture
    planes; for every mesh with a differing pixel it runs the whole oracle forward twice and records the largest change of the
    final 512x512 RGB image.

Writes profiles/r03_raster_contract_risk.json.   python oracle/raster_contract_risk.py [--meshes 100]
"""
import ctypes
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from next3d_amd import mesh as n3d_mesh, spec                # noqa: E402
from oracle import cases, generator as ogen, raster          # noqa: E402

GOLDEN = os.path.join(REPO, 'tests', 'golden')


def build_variant(flags, tag, tmp):
    so = os.path.join(tmp, f'liboracle_raster_{tag}.so')
    subprocess.check_call(['gcc', '-O2'] + flags + ['-shared', '-fPIC', '-o', so, os.path.join(REPO, 'oracle', 'raster_ref.c'), '-lm'])
    lib = ctypes.CDLL(so)
    lib.oracle_rasterize_meshes.restype = None
    lib.oracle_floodfill_fixed_range.restype = None
    return lib


def raster_outputs(lib, P, v, lms, textures, uv_mask):
    """The four views through the reference's transforms with `lib` as the rasteriser / flood fill -> (pix_to_face per view,
    alpha [N,3,256,256] after fill_mouth, projected planes)."""
    raster._lib = lib
    faces = P['faces'][0][:, [0, 2, 1]]
    p2fs = []
    for view in ogen.RENDERING_VIEWS:
        tv = raster.orth_project(v, raster.angle2matrix(view), ogen.ORTH_SHIFT, ogen.ORTH_SCALE)
        tv[:, :, 2] = tv[:, :, 2] + 10
        fixed = tv.clone()
        fixed[..., :2] = -fixed[..., :2]
        p2fs.append(raster.rasterize_meshes(fixed.float(), faces, image_size=256)[0])
    rend, alphas, _ = ogen.rasterize(P, v, lms, textures, uv_mask)
    return torch.stack(p2fs, 0), torch.cat(alphas, 1), torch.cat(rend, 1)


def main():
    n_pert = int(sys.argv[sys.argv.index('--meshes') + 1]) if '--meshes' in sys.argv else 100
    torch.set_num_threads(os.cpu_count())
    d = np.load(os.path.join(GOLDEN, 'demo_inputs.npz'))
    P = spec.synthetic_state_dict(0)
    P.update(n3d_mesh.mesh_buffers(d['faces'], d['uvs'], d['uvfaces']))
    uv_mask = n3d_mesh.synthetic_uv_face_mask()
    textures = torch.randn(1, 32, 256, 256, generator=torch.Generator().manual_seed(5))
    demo = torch.cat([torch.from_numpy(d['verts'])[None], torch.from_numpy(d['landmarks'])[None]], 1)        # [1, 5091, 3]
    meshes = []
    for case in ('case_r32_s24', 'case_r64_s48', 'case_r64_s96', 'case_r64_s48_b4'):
        g = np.load(os.path.join(GOLDEN, case + '.npz'))
        for i in range(g['v'].shape[0]):
            meshes.append((f'{case}[{i}]', torch.from_numpy(g['v'][i:i + 1])))
    gen = torch.Generator().manual_seed(2024)
    for k in range(n_pert):
        sigma = [0.00025, 0.0005, 0.001, 0.002][k % 4]
        meshes.append((f'demo+N(0,{sigma})#{k}', demo + sigma * torch.randn(demo.shape, generator=gen)))

    with tempfile.TemporaryDirectory() as tmp:
        lib_off = build_variant(['-ffp-contract=off'], 'off', tmp)
        lib_fma = build_variant(['-ffp-contract=fast', '-mfma'], 'fma', tmp)
        dis = subprocess.run(['objdump', '-d', os.path.join(tmp, 'liboracle_raster_fma.so')], capture_output=True, text=True).stdout
        n_fma_insn = sum(1 for ln in dis.splitlines() if 'vfmadd' in ln or 'vfmsub' in ln or 'vfnmadd' in ln)
        rows, flipped = [], []
        for name, v in meshes:
            vv, lms = v[:, :5023], v[:, 5023:]
            a = raster_outputs(lib_off, P, vv, lms, textures, uv_mask)
            b = raster_outputs(lib_fma, P, vv, lms, textures, uv_mask)
            face_px = int((a[0] != b[0]).sum())
            cover_px = int(((a[0] >= 0) != (b[0] >= 0)).sum())
            alpha_px = int((a[1] != b[1]).sum())
            dplane = float((a[2] - b[2]).abs().max())
            rows.append({'mesh': name, 'winning_face_pixels': face_px, 'coverage_pixels': cover_px, 'alpha_pixels_after_fill': alpha_px,
                         'max_abs_projected_planes': dplane})
            if face_px or alpha_px:
                flipped.append((name, v))
            print(rows[-1], flush=True)
        # whole forward for the meshes that changed: effect on the final image
        rk = dict(c_gen_conditioning_zero=True, c_scale=1.0, depth_resolution=48, depth_resolution_importance=48, ray_start=2.25, ray_end=3.3, box_warp=1)
        from next3d_amd import demo as n3d_demo
        z, c, c_cond, _ = n3d_demo.demo_batch([0], yaws=[0.3])
        jitter, u = cases.rng_inputs(1, 64, 48, 48)
        ws = ogen.mapping(P, z, c_cond, rk, truncation_psi=0.7, truncation_cutoff=14)
        image_effect = []
        for name, v in flipped[:8]:
            outs = []
            for lib in (lib_off, lib_fma):
                raster._lib = lib
                outs.append(ogen.synthesis(P, ws, c, v, uv_mask, rk, jitter, u, neural_rendering_resolution=64))
            image_effect.append({'mesh': name, 'max_abs_image': float((outs[0]['image'] - outs[1]['image']).abs().max()),
                                 'max_abs_image_raw': float((outs[0]['image_raw'] - outs[1]['image_raw']).abs().max())})
            print(image_effect[-1], flush=True)
        raster._lib = None
    tot = lambda k: int(sum(r[k] for r in rows))
    out = {
        'what': 'oracle/raster_ref.c built with -ffp-contract=off (pinned restatement) vs -ffp-contract=fast -mfma (FMA contraction as nvcc -fmad=true '
                'applies to PyTorch3D): pixels of the 4 x 256 x 256 view buffers per mesh that change',
        'fma_instructions_in_contracted_build': n_fma_insn,
        'meshes': len(rows), 'golden_meshes': len(rows) - n_pert, 'perturbed_meshes': n_pert,
        'meshes_with_any_difference': len(flipped),
        'total_winning_face_pixels': tot('winning_face_pixels'), 'total_coverage_pixels': tot('coverage_pixels'),
        'total_alpha_pixels_after_fill': tot('alpha_pixels_after_fill'),
        'pixels_examined': len(rows) * 4 * 256 * 256,
        'max_abs_projected_planes': max(r['max_abs_projected_planes'] for r in rows),
        'image_effect_on_changed_meshes': image_effect,
        'golden_rows': [r for r in rows if r['mesh'].startswith('case_')],
        'changed_rows': [r for r in rows if r['winning_face_pixels'] or r['alpha_pixels_after_fill']],
        'script': 'oracle/raster_contract_risk.py',
    }
    path = os.path.join(REPO, 'profiles', 'r03_raster_contract_risk.json')
    json.dump(out, open(path, 'w'), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k not in ('golden_rows', 'changed_rows')}, indent=1))


if __name__ == '__main__':
    main()
