"""-m gpu parity of the non-convolution kernels of the path — rasteriser + flood fill, texture projection, mouth box,
antialiased resize, plane blend, volume renderer, point queries — each called through the C ABI (include/n3d.h) on small
seeded inputs and compared with the CPU oracle (oracle/raster.py + raster_ref.c, oracle/renderer.py) or with the ATen op
the reference calls.  The whole-generator goldens (test_generator_gpu.py) cover the same kernels on the demo mesh only;
these cases add ragged sizes, degenerate / duplicated faces, out-of-range sampling coordinates and empty batches."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import cases, generator as ogen, raster, renderer

pytestmark = pytest.mark.gpu
VIEWS = ogen.RENDERING_VIEWS


def _gen(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def _rand(shape, seed, lo=0.0, hi=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.rand(shape, generator=g) * (hi - lo) + lo


def _md(a, b):
    return float((a.detach().cpu().double() - b.detach().cpu().double()).abs().max())


# ------------------------------------------------------------------------------------------------ rasteriser
def _soup(seed, n_faces=80):
    """Random overlapping triangles; the last three faces are a duplicate of face 0 (a z tie: the lower index must win,
    its uv attributes differ), a zero-area face and a face with a repeated vertex."""
    v = _rand((1, 3 * n_faces, 3), seed, -0.11, 0.11)
    faces = torch.arange(3 * n_faces, dtype=torch.int64).reshape(n_faces, 3)
    faces = torch.cat([faces, faces[:1], torch.tensor([[5, 5, 9]]), torch.tensor([[3, 4, 3]])], 0)
    uv = _rand((faces.shape[0], 3, 3), seed + 1, -1.0, 1.0)
    return v, faces, uv


def _cube(half=0.07):
    """Closed box, turned so that no face is edge-on in any of the four views (an edge-on face is a sliver one float ulp
    wide, whose coverage would depend on the rounding of the vertex transform): 2-3 big faces per view."""
    s = half
    v = torch.tensor([[[-s, -s, -s], [s, -s, -s], [s, s, -s], [-s, s, -s], [-s, -s, s], [s, -s, s], [s, s, s], [-s, s, s]]],
                     dtype=torch.float32) * torch.tensor([1.0, 0.83, 0.91])
    v = torch.matmul(v, raster.angle2matrix([17.0, 24.0, 9.0])[0])
    quads = [(0, 1, 2, 3), (5, 4, 7, 6), (4, 0, 3, 7), (1, 5, 6, 2), (3, 2, 6, 7), (4, 5, 1, 0)]
    faces = []
    for a, b, c, d in quads:                       # both windings: whatever the view, one of each pair is front-facing
        faces += [[a, b, c], [a, c, d], [a, c, b], [a, d, c]]
    faces = torch.tensor(faces, dtype=torch.int64)
    uvq = torch.tensor([[-0.9, -0.9, 1.0], [0.9, -0.9, 1.0], [0.9, 0.9, 1.0], [-0.9, 0.9, 1.0]])
    corner = {0: 0, 1: 1, 2: 2, 3: 3, 4: 1, 5: 0, 6: 3, 7: 2}
    uv = torch.stack([torch.stack([uvq[corner[int(i)]] for i in f]) for f in faces])
    return v, faces, uv


def _oracle_views(v, lms, faces, face_uv, mask, size, fill, binarize_view):
    """oracle.generator.rasterize (triplane_next3d.py:190-222) for an arbitrary mesh, mask and image size."""
    N = v.shape[0]
    grids, alphas = [], []
    for k, view in enumerate(VIEWS):
        tform = raster.angle2matrix(view)
        tv = raster.orth_project(v, tform, ogen.ORTH_SHIFT, ogen.ORTH_SCALE)
        tv[:, :, 2] = tv[:, :, 2] + 10
        rendering = raster.pytorch3d_rasterizer(tv, faces, face_uv, size)
        vis = rendering[:, -1:]
        grid = rendering[:, :-1].permute(0, 2, 3, 1)[:, :, :, :2]
        alpha = F.grid_sample(mask[None, None].expand(N, -1, -1, -1), grid, align_corners=False) * vis
        if fill:
            alpha = raster.fill_mouth(alpha)
            if k == binarize_view:
                alpha = alpha.bool().float()
        grids.append(grid)
        alphas.append(alpha[:, 0])
    lm2d = raster.orth_project(lms, raster.angle2matrix(VIEWS[0]), ogen.ORTH_SHIFT, ogen.ORTH_SCALE)[:, :, :2]
    return torch.stack(grids, 1), torch.stack(alphas, 1), lm2d          # [N,4,S,S,2], [N,4,S,S], [N,Lm,2]


def _gpu_views(dev, v, lms, faces, face_uv, mask, size, fill, binarize_view):
    from next3d_amd import _lib
    N, V, Lm, Fn, nv = v.shape[0], v.shape[1], lms.shape[1], faces.shape[0], len(VIEWS)
    rot = torch.cat([raster.angle2matrix(a) for a in VIEWS], 0).contiguous().to(dev)
    t = dict(dtype=torch.float32, device=dev)
    tv, zb = torch.empty(N * nv * V * 3, **t), torch.empty(N * nv * size * size, dtype=torch.int64, device=dev)
    grid, alpha, lm2d = torch.empty(N * nv, size, size, 2, **t), torch.empty(N, nv, size, size, **t), torch.empty(N, Lm, 2, **t)
    vd, ld, fd = v.contiguous().to(dev), lms.contiguous().to(dev), faces.to(torch.int32).contiguous().to(dev)
    ud, md = face_uv.contiguous().to(dev), mask.contiguous().to(dev)
    sh = ogen.ORTH_SHIFT.reshape(-1).tolist()
    _lib.check(_lib.lib().n3d_rasterize_views(_lib.ptr(vd), _lib.ptr(ld), _lib.ptr(rot), _lib.ptr(fd), _lib.ptr(ud), _lib.ptr(md),
                                              mask.shape[0], mask.shape[1], _lib.ptr(tv), _lib.ptr(zb), _lib.ptr(grid), _lib.ptr(alpha),
                                              _lib.ptr(lm2d), N, V, Lm, Fn, nv, size, size, sh[0], sh[1], sh[2], float(ogen.ORTH_SCALE.item()),
                                              1 if fill else 0, binarize_view, _lib.stream()))
    torch.cuda.synchronize()
    _gpu_views.tv = tv.reshape(N, nv, V, 3).cpu()                       # the transformed vertices the z-buffer pass read (PyTorch3D NDC: x, y negated)
    return grid.reshape(N, nv, size, size, 2).cpu(), alpha.cpu(), lm2d.cpu()


@pytest.mark.parametrize('size', [256, 100, 64])
def test_rasterize_views_triangle_soup(dev, size):
    """Coverage, z order, tie rule and barycentric uv of random overlapping triangles against oracle/raster_ref.c.  The
    vertex transform is evaluated in a different order on the two sides, so a pixel whose centre sits within rounding of an
    edge (or of two equally deep faces) may resolve differently: at most 0.1 % of the pixels may disagree end to end.  The second half of the test removes
    that slack: the transforms agree to 4e-7 relative, and the oracle rasterising the kernel's own transformed vertices agrees at EVERY pixel, bit for bit."""
    v0, faces, uv = _soup(11 + size)
    v = torch.cat([v0, v0.flip(1) * 0.9], 0)                                       # batch of 2 different meshes
    lms = _rand((2, 68, 3), 5, -0.1, 0.1)
    mask = torch.ones(8, 8)
    g_ref, a_ref, l_ref = _oracle_views(v, lms, faces, uv, mask, size, False, -1)
    g, a, l = _gpu_views(dev, v, lms, faces, uv, mask, size, False, -1)
    cov, cov_ref = a > 0, a_ref > 0                                                # alpha = mask sample (> 0 inside the mask) x visibility
    cov_bad = float((cov != cov_ref).float().mean())
    uv_err = (g - g_ref).abs().amax(-1)
    uv_bad = float((uv_err > 1e-4).float().mean())
    same = uv_err <= 1e-4
    print(f'size {size}: coverage {float(cov_ref.float().mean()):.3f}, coverage mismatch {cov_bad:.2e}, uv mismatch {uv_bad:.2e}, '
          f'lm2d {_md(l, l_ref):.2e}')
    assert 0.05 < float(cov_ref.float().mean()) < 0.95
    assert cov_bad <= 1e-3 and uv_bad <= 1e-3
    assert float((a - a_ref).abs()[same].max()) <= 1e-4
    assert _md(l, l_ref) <= 1e-6
    # ... and STRICTLY: the two sides' vertex transforms agree to rounding (ATen's bmm vs the kernel's fixed-order sum), and with the
    # oracle rasterising the kernel's OWN transformed vertices every pixel agrees — coverage, winning face and barycentric uv bit for bit
    tv = _gpu_views.tv
    N = v.shape[0]
    for k, view in enumerate(VIEWS):
        t_or = raster.orth_project(v, raster.angle2matrix(view), ogen.ORTH_SHIFT, ogen.ORTH_SCALE)
        t_or[:, :, 2] = t_or[:, :, 2] + 10
        back = tv[:, k].clone()
        back[..., :2] = -back[..., :2]                                              # undo Pytorch3dRasterizer.forward's sign flip (exact)
        assert float((back - t_or).abs().max()) <= 4e-7 * max(1.0, float(t_or.abs().max()))
        rendering = raster.pytorch3d_rasterizer(back, faces, uv, size)
        grid_k = rendering[:, :-1].permute(0, 2, 3, 1)[:, :, :, :2]
        alpha_k = (F.grid_sample(mask[None, None].expand(N, -1, -1, -1), grid_k, align_corners=False) * rendering[:, -1:])[:, 0]
        assert torch.equal(g[:, k], grid_k), (size, k, int((g[:, k] != grid_k).sum()))
        assert torch.equal(a[:, k] > 0, alpha_k > 0) and float((a[:, k] - alpha_k).abs().max()) <= 1e-6


def test_rasterize_views_tie_rule_and_empty_batch(dev):
    """Two coincident triangles (both windings, so every view that sees them sees a tie): the LOWER face index must win
    (PyTorch3D keeps the first face on equal depth) — the faces carry different uv attributes, so a wrong winner shows on
    every covered pixel.  N = 0 is a no-op."""
    from next3d_amd import _lib
    v = torch.tensor([[[-0.09, -0.07, 0.01], [0.08, -0.05, 0.03], [0.01, 0.09, -0.02]]])
    faces = torch.tensor([[0, 1, 2], [0, 1, 2], [0, 2, 1], [0, 2, 1]])
    uv = torch.stack([torch.full((3, 3), x) for x in (-0.6, 0.7, 0.2, -0.3)])
    lms = torch.zeros(1, 68, 3)
    g_ref, a_ref, _ = _oracle_views(v, lms, faces, uv, torch.ones(4, 4), 64, False, -1)
    g, a, _ = _gpu_views(dev, v, lms, faces, uv, torch.ones(4, 4), 64, False, -1)
    seen = a_ref > 0
    assert int(seen[0, 0].sum()) > 100
    vals = {round(float(x), 3) for x in torch.unique(g_ref[seen])}
    assert vals <= {-0.6, 0.2} and len(vals) >= 1                       # faces 0 / 2 win, never 1 / 3
    assert float(((g - g_ref).abs().amax(-1) > 1e-5).float().mean()) <= 1e-3
    p = _lib.ptr(torch.zeros(16, device=dev))
    assert _lib.lib().n3d_rasterize_views(p, p, p, p, p, p, 4, 4, p, p, p, p, p, 0, 3, 68, 4, 4, 64, 64, 0.0, 0.0, 0.0, 5.0, 1, 1,
                                          _lib.stream()) == 0


@pytest.mark.parametrize('size', [256, 100])
def test_rasterize_views_fill_mouth(dev, size):
    """fill_mouth (vr/renderer.py:583-602, cv2.floodFill from (0,0) in FIXED_RANGE mode): a box whose faces sample a uv
    mask with interior holes, a notch open to the border and grey (bilinearly interpolated) rims.  Interior holes become 1,
    the border-connected background stays, view 1 is additionally binarised (the reference's alpha_side)."""
    v, faces, uv = _cube()
    v = torch.cat([v, v * torch.tensor([0.8, 1.1, 0.7])], 0)
    lms = _rand((2, 68, 3), 6, -0.1, 0.1)
    mask = torch.ones(32, 32)
    mask[6:10, 7:12] = 0; mask[20:27, 18:22] = 0; mask[14:16, 3:5] = 0.5; mask[:5, 14:17] = 0; mask[28:, :] = 0
    _, a_nofill, _ = _oracle_views(v, lms, faces, uv, mask, size, False, -1)
    g_ref, a_ref, _ = _oracle_views(v, lms, faces, uv, mask, size, True, 1)
    g, a, _ = _gpu_views(dev, v, lms, faces, uv, mask, size, True, 1)
    filled = int(((a_ref == 1) & (a_nofill < 1)).sum())
    bad = int(((a * 255).round() != (a_ref * 255).round()).sum())
    print(f'size {size}: pixels the fill changed {filled}, mismatching {bad}, uv err {_md(g, g_ref):.2e}')
    assert filled > 50                                             # the case does exercise the fill
    assert bad <= 4
    assert _md(a[:, 1], a[:, 1].bool().float()) == 0
    g2, a2, _ = _gpu_views(dev, v, lms, faces, uv, mask, size, False, -1)
    assert int(((a2 * 255).round() != (a_nofill * 255).round()).sum()) <= 4 and torch.equal(g2, g)


def test_rasteriser_reproducible_under_coresident_convolutions(dev):
    """Round-1 finding: with 8-wave split-bf16 convolution workgroups of ANOTHER stream resident, the rasteriser
    returned different z-buffers / barycentrics from run to run (tools/dbg_race3.py: 12 of 12 with L1-served table loads).  The
    shipped kernels read their tables through agent-scope (L2-served) loads: the demo mesh rasterised while stride-1 and
    stride-2 convolutions run on a side stream must equal the quiet run bit for bit, every time."""
    import os
    from next3d_amd import mesh
    from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
    d = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'demo_inputs.npz'))
    c = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'case_r64_s48.npz'))
    mb = mesh.mesh_buffers(d['faces'], d['uvs'], d['uvfaces'])
    faces, face_uv = mb['faces'][0][:, [0, 2, 1]], mb['face_uvcoords'][0][:, [0, 2, 1]].contiguous()
    v = torch.from_numpy(c['v'])
    vv, lms = v[:, :5023], v[:, 5023:]
    mask = F.interpolate(mesh.synthetic_uv_face_mask().float(), [256, 256])[0, 0]
    quiet = _gpu_views(dev, vv, lms, faces, face_uv, mask, 256, True, 1)
    x0, x1 = _gen((4, 256, 128, 128), 90).to(dev), _gen((4, 256, 129, 129), 91).to(dev)
    wt = cg.prep_weight_bf16x3((_gen((256, 256, 3, 3), 92) / 48).to(dev))
    side = torch.cuda.Stream()
    for it in range(8):
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                cg.conv_launch(x0, wt, 3, 0, 256, bf16x3=True)
                cg.conv_launch(x1, wt, 3, 1, 256, bf16x3=True)
        busy = _gpu_views(dev, vv, lms, faces, face_uv, mask, 256, True, 1)
        torch.cuda.current_stream().wait_stream(side)
        for a, b, name in zip(busy, quiet, ('grid', 'alpha', 'lm2d')):
            assert torch.equal(a, b), (it, name, int((a != b).sum()))


def test_texture_project_matches_grid_sample(dev):
    """triplane_next3d.py:223-230: F.grid_sample(textures, uv, bilinear, zeros, align_corners=False), the side plane being
    the sum of two views; coordinates beyond [-1,1] exercise the zero padding.  One-plane and three-plane entry points."""
    from next3d_amd import _lib
    N, C, TH, TW, S, nv = 2, 5, 24, 40, 33, 4
    tex = _gen((N, C, TH, TW), 21)
    grid = _rand((N * nv, S, S, 2), 22, -1.3, 1.3)
    grid[0, :4, :4] = torch.tensor([[-1.0, -1.0], [1.0, 1.0], [0.0, 0.0], [-1.0 + 1.0 / TW, 1.0 - 1.0 / TH]])[None, :, :].expand(4, -1, -1)
    gs = lambda view: F.grid_sample(tex, grid.reshape(N, nv, S, S, 2)[:, view], mode='bilinear', padding_mode='zeros', align_corners=False)
    ref = [gs(0), gs(1) + gs(2), gs(3)]
    td, gd = tex.to(dev), grid.to(dev)
    outs = [torch.empty(N, C, S, S, device=dev) for _ in range(3)]
    L = _lib.lib()
    _lib.check(L.n3d_texture_project_planes(_lib.ptr(td), _lib.ptr(gd), (ctypes.c_void_p * 3)(*[o.data_ptr() for o in outs]),
                                            (ctypes.c_int * 3)(0, 1, 3), (ctypes.c_int * 3)(-1, 2, -1), 3, N, C, TH, TW, S, S, nv, _lib.stream()))
    for o, r in zip(outs, ref):
        assert _md(o, r) <= 2e-6 * max(1.0, float(r.abs().max()))
    one = torch.empty(N, C, S, S, device=dev)
    _lib.check(L.n3d_texture_project(_lib.ptr(td), _lib.ptr(gd), _lib.ptr(one), N, C, TH, TW, S, S, nv, 1, 2, _lib.stream()))
    assert torch.equal(one, outs[1])
    assert L.n3d_texture_project(None, None, _lib.ptr(one), 0, C, TH, TW, S, S, nv, 0, -1, _lib.stream()) == 0      # empty batch
    assert L.n3d_texture_project(_lib.ptr(td), _lib.ptr(gd), _lib.ptr(one), N, C, TH, TW, S, S, nv, 4, -1, _lib.stream()) != 0
    assert 'texture_project' in L.n3d_last_error().decode()


def test_mouth_bbox_matches_gen_mouth_mask(dev):
    """gen_mouth_mask (triplane_next3d.py:330-344): numpy float32 extents, int truncation, python floor division —
    bit-exact on random landmark sets, including boxes that leave the image and negative coordinates."""
    from next3d_amd import _lib
    lm = _rand((96, 68, 2), 31, -0.9, 0.9)
    lm[:32] *= 0.25                                    # small mouths (the realistic regime: 30-60 pixel boxes)
    lm[90:] -= 1.2                                     # negative pixel coordinates
    ref = raster.gen_mouth_mask(lm)
    out = torch.empty(96, 4, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().n3d_mouth_bbox(_lib.ptr(lm.to(dev)), _lib.ptr(out), 96, 68, _lib.stream()))
    assert np.array_equal(out.cpu().numpy().astype(np.int64), np.asarray(ref).astype(np.int64))


# ------------------------------------------------------------------------------------------------ antialiased resize
def _resize(dev, src, dst, src_box, dst_box, dst_square):
    from next3d_amd import _lib
    N, C, SH, SW = src.shape
    _lib.check(_lib.lib().n3d_resize_aa(_lib.ptr(src), _lib.ptr(dst), _lib.ptr(src_box), _lib.ptr(dst_box), N, C, SH, SW, dst.shape[2],
                                        dst.shape[3], dst_square, _lib.stream()))
    return dst


@pytest.mark.parametrize('shape,out', [((2, 3, 256, 256), (64, 64)), ((1, 4, 64, 64), (128, 128)), ((1, 2, 37, 53), (20, 64)),
                                       ((2, 2, 128, 128), (128, 128)), ((1, 1, 200, 90), (7, 5))])
def test_resize_aa_matches_aten(dev, shape, out):
    """F.interpolate(mode='bilinear', antialias=True, align_corners=False) (superresolution.py:282-286): down-, up- and
    mixed scaling, identity."""
    x = _gen(shape, 41)
    ref = F.interpolate(x, size=out, mode='bilinear', align_corners=False, antialias=True)
    y = _resize(dev, x.to(dev), torch.empty(shape[0], shape[1], *out, device=dev), None, None, 0)
    assert _md(y, ref) <= 5e-6


def test_resize_aa_strided_channel_slice(dev):
    """n3d_resize_aa_strided: a channel-slice view (the first 3 of the renderer's 32 feature channels = the super-resolution's RGB input,
    triplane_next3d.py:179-181) resized straight from the 32-channel tensor — bit-equal to the resize of its contiguous copy."""
    from next3d_amd import _lib, generator
    feat = _gen((3, 32, 64, 64), 45).to(dev)
    rgb = feat[:, :3]
    assert not rgb.is_contiguous()
    y = generator._resize_aa(rgb, 128)
    y_ref = generator._resize_aa(rgb.contiguous(), 128)
    assert torch.equal(y, y_ref)
    assert _md(y, F.interpolate(rgb.cpu(), size=(128, 128), mode='bilinear', align_corners=False, antialias=True)) <= 5e-6
    p = _lib.ptr(feat)
    assert _lib.lib().n3d_resize_aa_strided(p, 100, p, None, None, 1, 3, 64, 64, 8, 8, 0, _lib.stream()) != 0       # batch stride smaller than one sample


def test_resize_aa_crop_and_paste_boxes(dev):
    """The mouth crop (triplane_next3d.py:151-152: per-sample box -> 64²) and the paste back (:156-163: 256² -> s×s into the
    box, everything else untouched), boxes of different sizes per sample."""
    x = _gen((3, 4, 256, 256), 42)
    boxes = torch.tensor([[82, 122, 108, 148], [10, 67, 190, 247], [100, 200, 3, 103]], dtype=torch.int32)
    crop = _resize(dev, x.to(dev), torch.empty(3, 4, 64, 64, device=dev), boxes.to(dev), None, 0)
    for i, (y0, y1, x0, x1) in enumerate(boxes.tolist()):
        ref = F.interpolate(x[i:i + 1, :, y0:y1, x0:x1], size=(64, 64), mode='bilinear', antialias=True)
        assert _md(crop[i:i + 1], ref) <= 5e-6
    src = _gen((3, 4, 256, 256), 43)
    dst0 = _gen((3, 4, 256, 256), 44)
    dst = _resize(dev, src.to(dev), dst0.clone().to(dev), None, boxes.to(dev), 1)
    for i, (y0, y1, x0, x1) in enumerate(boxes.tolist()):
        s = y1 - y0
        ref = dst0[i:i + 1].clone()
        ref[:, :, y0:y1, x0:x1] = F.interpolate(src[i:i + 1], size=(s, s), mode='bilinear', antialias=True)
        assert _md(dst[i:i + 1], ref) <= 5e-6
        outside = torch.ones(256, 256, dtype=torch.bool); outside[y0:y1, x0:x1] = False
        assert torch.equal(dst[i].cpu()[:, outside], dst0[i][:, outside])
    # the paste kernel's other regimes: a tiny box (66 taps per axis), a box wider than its LDS weight tables (weights on the fly), a box that sticks out of the tensor
    for (SH, DH, box) in [(256, 64, [20, 28, 30, 38]), (1024, 800, [20, 720, 30, 730]), (256, 256, [200, 300, -20, 80])]:
        src, dst0 = _gen((1, 2, SH, SH), 45), _gen((1, 2, DH, DH), 46)
        y0, y1, x0, x1 = box
        s = y1 - y0
        dst = _resize(dev, src.to(dev), dst0.clone().to(dev), None, torch.tensor([box], dtype=torch.int32).to(dev), 1)
        full = F.interpolate(src, size=(s, s), mode='bilinear', antialias=True)
        ref = dst0.clone()
        ys, xs = slice(max(y0, 0), min(y1, DH)), slice(max(x0, 0), min(x1, DH))
        ref[:, :, ys, xs] = full[:, :, ys.start - y0:ys.stop - y0, xs.start - x0:xs.stop - x0]
        assert _md(dst, ref) <= 5e-6, (SH, DH, box)


# ------------------------------------------------------------------------------------------------ blend + renderer
def test_blend_planes_and_layout(dev):
    """triplane_next3d.py:171-174, written channels-last; n3d_planes_to_channels_last is the plain layout change."""
    from next3d_amd import _lib
    N, H, W = 2, 24, 40
    front, side, top, stat = _gen((N, 32, H, W), 51), _gen((N, 32, H, W), 52), _gen((N, 32, H, W), 53), _gen((N, 96, H, W), 54)
    alpha = _rand((N, 3, H, W), 55)
    alpha[:, :, :4] = 0; alpha[:, :, 4:8] = 1
    dyn = torch.cat([front, side, top], 1).view(N, 3, 32, H, W)
    a = alpha.unsqueeze(2)
    ref = dyn * a + stat.view(N, 3, 32, H, W) * (1 - a)
    out = torch.empty(N, 3, H, W, 32, device=dev)
    L = _lib.lib()
    d = [t.to(dev) for t in (front, side, top, stat, alpha)]
    _lib.check(L.n3d_blend_planes(*[_lib.ptr(t) for t in d], _lib.ptr(out), N, H, W, _lib.stream()))
    assert _md(out.permute(0, 1, 4, 2, 3), ref) <= 1e-6
    cl = torch.empty(N, 3, H, W, 32, device=dev)
    refd = ref.contiguous().to(dev)
    _lib.check(L.n3d_planes_to_channels_last(_lib.ptr(refd), _lib.ptr(cl), N, H, W, _lib.stream()))
    assert torch.equal(cl.permute(0, 1, 4, 2, 3).cpu(), ref)
    # n3d_blend_planes_views: alpha as the rasteriser leaves it, [N, 4 views, H, W], planes blended with views 0 / 1 / 3 — bit-equal to index_select + blend
    alpha4 = torch.stack([alpha[:, 0], alpha[:, 1], _rand((N, H, W), 56), alpha[:, 2]], 1).contiguous().to(dev)
    out4 = torch.empty_like(out)
    _lib.check(L.n3d_blend_planes_views(*[_lib.ptr(t) for t in d[:4]], _lib.ptr(alpha4), _lib.ptr(out4), N, H, W, 4, 0, 1, 3, _lib.stream()))
    assert torch.equal(out4, out)
    assert L.n3d_blend_planes_views(*[_lib.ptr(t) for t in d[:4]], _lib.ptr(alpha4), _lib.ptr(out4), N, H, W, 4, 0, 1, 4, _lib.stream()) != 0      # view index out of range


def test_unpack_inputs(dev):
    """n3d_unpack_inputs: synthesis' v [N, V + L, 3] / c [N, 25] -> dense verts / landmarks / cam2world / intrinsics in one launch (a batch-strided
    v: the reenactment loop hands over slices of a longer sequence) — pure data movement, bit-equal to the four torch slices it replaces."""
    from next3d_amd import _lib
    N, V, Lm = 3, 5023, 68
    seq = _gen((N, V + Lm + 7, 3), 57).to(dev)
    v = seq[:, :V + Lm]                                   # batch stride (V + L + 7) * 3
    c = _gen((N, 27), 58).to(dev)[:, :25]
    f32 = dict(dtype=torch.float32, device=dev)
    verts, lms, cam, intr = torch.empty(N, V, 3, **f32), torch.empty(N, Lm, 3, **f32), torch.empty(N, 16, **f32), torch.empty(N, 9, **f32)
    _lib.check(_lib.lib().n3d_unpack_inputs(_lib.ptr(v), v.stride(0), _lib.ptr(c), c.stride(0), _lib.ptr(verts), _lib.ptr(lms), _lib.ptr(cam), _lib.ptr(intr), N, V, Lm, _lib.stream()))
    assert torch.equal(verts, v[:, :V]) and torch.equal(lms, v[:, V:]) and torch.equal(cam, c[:, :16]) and torch.equal(intr, c[:, 16:25])
    # empty batch: a no-op (return code 0, nothing dereferenced); null tensors with N > 0: an error code + message, no launch
    L = _lib.lib()
    assert L.n3d_unpack_inputs(None, 0, None, 0, None, None, None, None, 0, V, Lm, _lib.stream()) == 0
    assert L.n3d_unpack_inputs(None, 0, _lib.ptr(c), 25, _lib.ptr(verts), _lib.ptr(lms), _lib.ptr(cam), _lib.ptr(intr), 1, V, Lm, _lib.stream()) != 0
    assert b'null tensor' in L.n3d_last_error()
    assert L.n3d_layout_grid_u8(None, None, 0, 3, 8, 8, 1, 1, 1, _lib.stream()) != 0          # 0 frames do not fill a 1 x 1 grid
    assert L.n3d_blend_planes_views(None, None, None, None, None, None, 0, 8, 8, 4, 0, 1, 3, _lib.stream()) == 0


def _decoder(seed):
    """A seeded OSGDecoder (triplane_next3d.py:353-357) as the oracle's parameter dict and as the kernel's pre-scaled arrays."""
    P = {'decoder.net.0.weight': _gen((64, 32), seed), 'decoder.net.0.bias': _gen((64,), seed + 1, 0.3),
         'decoder.net.2.weight': _gen((33, 64), seed + 2), 'decoder.net.2.bias': _gen((33,), seed + 3, 0.3)}
    w1 = (P['decoder.net.0.weight'] / np.sqrt(32)).contiguous()
    w2t = torch.cat([(P['decoder.net.2.weight'] / np.sqrt(64)).t(), torch.zeros(64, 1)], 1).contiguous()
    return P, (w1, P['decoder.net.0.bias'], w2t, P['decoder.net.2.bias'])


def _channels_last(planes):
    return planes.permute(0, 1, 3, 4, 2).contiguous()


@pytest.mark.parametrize('split', [0, 1], ids=['decoder_f32', 'decoder_split_bf16'])
@pytest.mark.parametrize('R,Sc,Sf,PH,PW', [(8, 48, 48, 32, 32), (6, 24, 24, 16, 40), (5, 96, 96, 32, 32), (4, 12, 0, 8, 8), (4, 64, 17, 24, 24),
                                          (6, 8, 8, 16, 16), (3, 47, 2, 16, 16), (1, 48, 48, 16, 16), (3, 20, 12, 24, 24), (4, 33, 31, 24, 24), (7, 40, 8, 32, 32)])
def test_render_rays_matches_oracle(dev, R, Sc, Sf, PH, PW, split):
    """Ray sampler + two-pass importance renderer + decoder + ray marcher (vr/renderer.py:95-268, vr/ray_marcher.py:27-66)
    on random tri-planes: 48+48 (the headline configuration), 24+24, 96+96 (gen_videos' sampling multiplier 2), coarse only,
    an odd split, and shapes whose LAST decode pass is partly filled in every way (few samples, one ray per wave, a two-sample importance pass: the lanes beyond the last
    sample repeat it — builds that left them idle miscomputed exactly these).  Tolerance 1e-3 max-abs on the composited features / depth (north_star) at EVERY ray with the importance depths teacher-forced
    (n3d_render_opts.fine_depths_in = the oracle's own); free-running, the importance pass is discontinuous in the coarse weights: the rays whose sampled
    depths agree (all but a handful, counted) are held to 1e-3 as well.  Both decoders: float32-input MFMAs and the split-bf16 form the default route uses
    (n3d_render_opts.decoder_split_bf16, ABI 8), same bounds."""
    from next3d_amd import _lib, demo as camera_utils
    N = 2
    planes = _gen((N, 3, 32, PH, PW), 60 + R, 2.0)
    P, (w1, b1, w2t, b2) = _decoder(61)
    c = torch.cat([camera_utils.demo_camera_params(angle_y=a, angle_p=-0.2)[0] for a in (0.35, -0.3)], 0).float()
    ray_o, ray_d = renderer.ray_sampler(c[:, :16].reshape(N, 4, 4), c[:, 16:25].reshape(N, 3, 3), R)
    jitter, u = cases.rng_inputs(N, R, Sc, max(Sf, 1))
    u = u[:, :Sf]
    opts = dict(depth_resolution=Sc, depth_resolution_importance=Sf, ray_start=2.25, ray_end=3.3, box_warp=1)
    fine = []
    rgb, depth, wsum = renderer.importance_renderer(P, 'decoder', planes, ray_o, ray_d, opts, jitter, u, fine_depths_out=fine)
    t = dict(dtype=torch.float32, device=dev)
    d = [x.contiguous().to(dev) for x in (_channels_last(planes), c[:, :16], c[:, 16:25], torch.linspace(2.25, 3.3, Sc), jitter,
                                           u if Sf else torch.zeros(1), w1, b1, w2t, b2)]

    def run(fine_in=None, fine_out=None):
        feat, dep, ws_, bounds = torch.empty(N, 32, R, R, **t), torch.empty(N, 1, R, R, **t), torch.empty(N, R * R, **t), torch.empty(2, **t)
        ro = _lib.RenderOpts()
        ro.ray_start, ro.ray_end, ro.box_side = 2.25, 3.3, 1.0
        ro.fine_depths_in, ro.fine_depths_out = _lib.ptr(fine_in), _lib.ptr(fine_out)
        ro.decoder_split_bf16 = split
        _lib.check(_lib.lib().n3d_render_rays_ex(*[_lib.ptr(x) for x in d], _lib.ptr(feat), _lib.ptr(dep), _lib.ptr(ws_), _lib.ptr(bounds), N, R, Sc,
                                                 Sf, PH, PW, float((3.3 - 2.25) / (Sc - 1)), 2.0, ro, _lib.stream()))
        e_rgb = (feat.cpu().reshape(N, 32, R * R).permute(0, 2, 1) - rgb).abs().amax(-1)
        e_dep = (dep.cpu().reshape(N, R * R) - depth[..., 0]).abs()
        e_w = (ws_.cpu() - wsum.reshape(N, R * R)).abs()
        return e_rgb, e_dep, e_w

    # (1) teacher-forced: the oracle's importance depths go in — second decode, merge, march and composite on IDENTICAL samples: EVERY ray within 1e-3
    if Sf:
        e_rgb, e_dep, e_w = run(fine_in=fine[0].reshape(N, R * R, Sf).contiguous().to(dev))
        print(f'R{R} {Sc}+{Sf} teacher-forced: rgb max {float(e_rgb.max()):.2e} median {float(e_rgb.median()):.2e}, depth max {float(e_dep.max()):.2e}, wsum max {float(e_w.max()):.2e}')
        for e in (e_rgb, e_dep, e_w):
            assert float(e.max()) <= 1e-3 and float(e.median()) <= 1e-4
    # (2) free-running: the kernel's own importance depths against the oracle's — equal to float32 rounding on all but the few samples whose u sits on a CDF step
    # (the seam is discontinuous there), and the composite within 1e-3 on all but those rays
    own = torch.zeros(N, R * R, max(Sf, 1), **t)
    e_rgb, e_dep, e_w = run(fine_out=own if Sf else None)
    print(f'R{R} {Sc}+{Sf}: rgb max {float(e_rgb.max()):.2e} median {float(e_rgb.median()):.2e}, depth max {float(e_dep.max()):.2e}, '
          f'wsum max {float(e_w.max()):.2e}')
    bad_rays = torch.zeros(N, R * R, dtype=torch.bool)
    if Sf:
        dd = (own.cpu() - fine[0].reshape(N, R * R, Sf)).abs()
        bad = dd > 1e-5
        bad_rays = bad.any(-1)
        print(f'   importance depths: {int(bad.sum())} of {bad.numel()} samples off by more than 1e-5 (max {float(dd.max()):.2e}) on {int(bad_rays.sum())} rays')
        assert float(bad.float().mean()) <= 2e-3
    for e in (e_rgb, e_dep, e_w):
        assert float(e[~bad_rays].max()) <= 1e-3                          # every ray whose samples agree
        assert float((e > 1e-3).float().mean()) <= 0.01 and float(e.median()) <= 1e-4


@pytest.mark.parametrize('name', ['white_back', 'disparity', 'auto', 'auto_wide_fov', 'density_noise', 'all'])
@pytest.mark.parametrize('split', [0, 1], ids=['decoder_f32', 'decoder_split_bf16'])
def test_render_rays_options_match_reference_golden(dev, name, split):
    """n3d_render_rays_ex — 'auto' ray bounds (per-ray box limits + the reference's repair of the rays that miss the box),
    disparity-space sampling, white_back, density noise — against the REFERENCE's own ImportanceRenderer outputs
    (tests/golden/render_opts.npz, oracle/pin_renderer_options.py), and through generator.render's reading of rendering_kwargs."""
    from next3d_amd import _lib
    from test_cpu_oracle import load_render_opts_golden
    inp, cases_ = load_render_opts_golden()
    opts, cams, rgb, depth, wsum = cases_[name]
    planes, R, Sc, Sf = inp['planes'], inp['R'], inp['Sc'], inp['Sf']
    N, PH, PW = planes.shape[0], planes.shape[3], planes.shape[4]
    P = inp['P']
    w1 = (P['decoder.net.0.weight'] / np.sqrt(32)).contiguous()
    w2t = torch.cat([(P['decoder.net.2.weight'] / np.sqrt(64)).t(), torch.zeros(64, 1)], 1).contiguous()
    auto, disp = opts['ray_start'] == 'auto', opts.get('disparity_space_sampling', False)
    lo, hi = (0.0, 1.0) if (auto or disp) else (opts['ray_start'], opts['ray_end'])
    t = dict(dtype=torch.float32, device=dev)
    feat, dep, ws_, bounds = torch.empty(N, 32, R, R, **t), torch.empty(N, 1, R, R, **t), torch.empty(N, R * R, **t), torch.empty(2, **t)
    d = [x.contiguous().to(dev) for x in (_channels_last(planes), cams[:, :16], cams[:, 16:25], torch.linspace(lo, hi, Sc), inp['jitter'], inp['u'], w1,
                                           P['decoder.net.0.bias'], w2t, P['decoder.net.2.bias'])]
    ro = _lib.RenderOpts()
    ro.decoder_split_bf16 = split
    ro.white_back, ro.disparity_space_sampling, ro.auto_bounds, ro.box_side = int(opts.get('white_back', False)), int(disp), int(auto), 1.0
    rb = torch.empty(N * R * R * 2, **t)
    nc, nf = (x.reshape(N, R * R, -1).contiguous().to(dev) for x in inp['noise'])
    if auto:
        ro.ray_bounds_ws = _lib.ptr(rb)
    else:
        ro.ray_start, ro.ray_end = opts['ray_start'], opts['ray_end']
    if opts.get('density_noise', 0) > 0:
        ro.density_noise, ro.density_noise_coarse, ro.density_noise_fine = opts['density_noise'], _lib.ptr(nc), _lib.ptr(nf)
    fine_ref = inp['fine'][name].reshape(N, R * R, Sf).contiguous().to(dev)      # the REFERENCE's own importance depths
    own = torch.zeros(N, R * R, Sf, **t)
    for forced in (True, False):
        ro.fine_depths_in, ro.fine_depths_out = (_lib.ptr(fine_ref), None) if forced else (None, _lib.ptr(own))
        _lib.check(_lib.lib().n3d_render_rays_ex(*[_lib.ptr(x) for x in d], _lib.ptr(feat), _lib.ptr(dep), _lib.ptr(ws_), _lib.ptr(bounds), N, R, Sc, Sf, PH, PW,
                                                 float((hi - lo) / (Sc - 1)), 2.0, ro, _lib.stream()))
        e_rgb = (feat.cpu().reshape(N, 32, R * R).permute(0, 2, 1) - rgb).abs().amax(-1)
        e_dep = (dep.cpu().reshape(N, R * R) - depth[..., 0]).abs()
        e_w = (ws_.cpu() - wsum.reshape(N, R * R)).abs()
        print(f'{name}{" teacher-forced" if forced else ""}: rgb max {float(e_rgb.max()):.2e} median {float(e_rgb.median()):.2e}, depth max {float(e_dep.max()):.2e}, wsum max {float(e_w.max()):.2e}')
        if forced:                                                        # identical samples behind the seam: EVERY ray within the tolerance
            for e in (e_rgb, e_dep, e_w):
                assert float(e.max()) <= 1e-3 and float(e.median()) <= 1e-4
        else:                                                             # free-running: the importance pass is discontinuous in the coarse weights — the rays whose sampled depths agree are held to 1e-3
            bad = (own.cpu() - fine_ref.cpu()).abs() > 1e-5 * max(1.0, float(fine_ref.abs().max()))
            assert float(bad.float().mean()) <= 5e-3
            for e in (e_rgb, e_dep, e_w):
                assert float(e[~bad.any(-1)].max()) <= 1e-3 and float((e > 1e-3).float().mean()) <= 0.01 and float(e.median()) <= 1e-4
    if auto:                                                              # the per-ray limits themselves: exactly the reference's slab test + repair
        o_ro, o_rd = renderer.ray_sampler(cams[:, :16].reshape(N, 4, 4), cams[:, 16:25].reshape(N, 3, 3), R)
        rs, re = renderer.ray_limits_box(o_ro, o_rd, 1)
        ok = re > rs
        rs[~ok], re[~ok] = rs[ok].min(), rs[ok].max()
        got = rb.cpu().reshape(N, R * R, 2)
        assert _md(got[..., 0:1], rs) <= 2e-6 and _md(got[..., 1:2], re) <= 2e-6


def test_render_rays_empty_space(dev):
    """Zero density everywhere: all weights are 0, the composite colour is 0 (-> -1 after rgb*2-1), the depth is 0/0 ->
    nan_to_num(inf) -> clamped to the GLOBAL maximum sample depth of the batch (ray_marcher.py:52-54; the kernel's
    depth-bounds pre-pass), identically for every ray."""
    from next3d_amd import _lib, demo as camera_utils
    N, R, Sc, Sf, PH, PW = 3, 16, 48, 48, 8, 8
    planes = torch.zeros(N, 3, 32, PH, PW)
    P, (w1, b1, w2t, b2) = _decoder(62)
    b2 = b2.clone(); b2[0] = -200.0
    P['decoder.net.2.bias'] = b2
    c = torch.cat([camera_utils.demo_camera_params(angle_y=a, angle_p=-0.1)[0] for a in (0.3, 0.0, -0.3)], 0).float()
    ray_o, ray_d = renderer.ray_sampler(c[:, :16].reshape(N, 4, 4), c[:, 16:25].reshape(N, 3, 3), R)
    jitter, u = cases.rng_inputs(N, R, Sc, Sf, seed=7)
    opts = dict(depth_resolution=Sc, depth_resolution_importance=Sf, ray_start=2.25, ray_end=3.3, box_warp=1)
    rgb, depth, wsum = renderer.importance_renderer(P, 'decoder', planes, ray_o, ray_d, opts, jitter, u)
    assert float(wsum.max()) == 0 and float(depth.min()) == float(depth.max()) > 3.3
    t = dict(dtype=torch.float32, device=dev)
    feat, dep, bounds = torch.empty(N, 32, R, R, **t), torch.empty(N, 1, R, R, **t), torch.empty(2, **t)
    d = [x.contiguous().to(dev) for x in (_channels_last(planes), c[:, :16], c[:, 16:25], torch.linspace(2.25, 3.3, Sc), jitter, u, w1, b1, w2t, b2)]
    _lib.check(_lib.lib().n3d_render_rays(*[_lib.ptr(x) for x in d], _lib.ptr(feat), _lib.ptr(dep), None, _lib.ptr(bounds), N, R, Sc, Sf, PH, PW,
                                          float((3.3 - 2.25) / (Sc - 1)), 2.0, _lib.stream()))
    assert float(feat.min()) == -1.0 and float(feat.max()) == -1.0
    assert _md(dep, torch.full_like(dep, float(depth.max())).cpu()) <= 5e-7
    lo = (torch.linspace(2.25, 3.3, Sc)[0] + jitter[:, :, 0, 0] * ((3.3 - 2.25) / (Sc - 1))).min()
    # the scratch holds the two bounds as order-preserving unsigned keys (render.hip f2key: 'auto' ray bounds can be negative)
    keys = bounds.cpu().view(torch.int32).numpy().astype(np.uint32)
    dec = [np.array([(k ^ 0x80000000) if (k & 0x80000000) else (~k & 0xffffffff)], dtype=np.uint32).view(np.float32)[0] for k in keys.tolist()]
    assert abs(float(dec[0]) - float(lo)) <= 5e-7 and abs(float(dec[1]) - float(depth.max())) <= 5e-7


def test_render_rays_rejects_bad_arguments(dev):
    from next3d_amd import _lib
    L = _lib.lib()
    x = torch.zeros(64, device=dev)
    p = _lib.ptr(x)
    assert L.n3d_render_rays(p, p, p, p, p, p, p, p, p, p, p, p, None, p, 1, 4, 3, 0, 8, 8, 0.1, 2.0, _lib.stream()) != 0      # Sc < 4
    assert 'render_rays' in L.n3d_last_error().decode()
    assert L.n3d_render_rays(p, p, p, p, p, p, p, p, p, p, p, p, None, p, 1, 4, 200, 100, 8, 8, 0.1, 2.0, _lib.stream()) != 0  # too many samples
    assert L.n3d_render_rays(None, p, p, p, p, p, p, p, p, p, p, p, None, p, 1, 4, 8, 0, 8, 8, 0.1, 2.0, _lib.stream()) != 0    # null planes


@pytest.mark.parametrize('M', [1, 63, 257, 4099])
def test_sample_points_matches_oracle(dev, M):
    """ImportanceRenderer.run_model (vr/renderer.py:149-155) at ragged point counts; a third of the points lie outside the
    box (grid_sample's zero padding)."""
    from next3d_amd import _lib
    N, PH, PW = 2, 20, 28
    planes = _gen((N, 3, 32, PH, PW), 70, 0.7)
    P, (w1, b1, w2t, b2) = _decoder(71)
    coords = _rand((N, M, 3), 72 + M, -0.75, 0.75)
    ref = ogen.run_model(P, planes, coords, dict(box_warp=1))
    rgb, sigma = torch.empty(N, M, 32, device=dev), torch.empty(N, M, 1, device=dev)
    d = [x.contiguous().to(dev) for x in (_channels_last(planes), coords, w1, b1, w2t, b2)]
    _lib.check(_lib.lib().n3d_sample_points(*[_lib.ptr(x) for x in d], _lib.ptr(rgb), _lib.ptr(sigma), N, M, PH, PW, 2.0, _lib.stream()))
    assert _md(rgb, ref['rgb']) <= 1e-4
    assert _md(sigma, ref['sigma']) <= 1e-4 * max(1.0, float(ref['sigma'].abs().max()))
    assert _lib.lib().n3d_sample_points(*[_lib.ptr(x) for x in d], _lib.ptr(rgb), _lib.ptr(sigma), N, 0, PH, PW, 2.0, _lib.stream()) == 0


def test_layout_grid_uint8_frames(dev):
    """The video scripts' output step (gen_videos_next3d.py:35-49, :171): float frames -> uint8 on the device -> tiled grid."""
    from next3d_amd import frames
    img = _gen((4, 3, 16, 24), 90, 0.8)
    ref = (img * 127.5 + 128).clamp(0, 255).to(torch.uint8)
    out = frames.layout_grid(img.to(dev), grid_w=2, grid_h=2)
    assert out.shape == (32, 48, 3) and out.dtype == np.uint8
    for b in range(4):
        gy, gx = b // 2, b % 2
        assert np.array_equal(out[gy * 16:(gy + 1) * 16, gx * 24:(gx + 1) * 24], ref[b].permute(1, 2, 0).numpy())
    # channels-first canvas, kept on the device (one n3d_layout_grid_u8 launch as well); 1 x 4 strip with the width inferred
    chw = frames.layout_grid(img.to(dev), grid_h=1, chw_to_hwc=False, to_numpy=False)
    assert chw.is_cuda and tuple(chw.shape) == (3, 16, 96) and chw.dtype == torch.uint8
    assert all(torch.equal(chw[:, :, b * 24:(b + 1) * 24].cpu(), ref[b]) for b in range(4))
    # a width the one-pass kernel does not take (W % 4 != 0): conversion kernel + copy tiling, same result
    odd = _gen((2, 3, 8, 10), 91, 0.8)
    out = frames.layout_grid(odd.to(dev), grid_w=1, grid_h=2)
    assert np.array_equal(out, (odd * 127.5 + 128).clamp(0, 255).to(torch.uint8).permute(0, 2, 3, 1).reshape(16, 10, 3).numpy())


# ------------------------------------------------------------------------------------------------ third-party shims (boundary B1)
def _reference_rasterizer_forward(vertices, faces, attributes, image_size):
    """The call sequence of the reference's Pytorch3dRasterizer.forward (vr/renderer.py:401-440) against whatever `pytorch3d` is
    importable — with next3d_amd.install_dropin(third_party=True): next3d_amd/shims/pytorch3d — same keyword arguments, same
    tensor handling (restated here: /root/reference does not exist on the GPU box)."""
    from pytorch3d.renderer.mesh import rasterize_meshes
    from pytorch3d.structures import Meshes
    fixed = vertices.clone()
    fixed[..., :2] = -fixed[..., :2]
    meshes = Meshes(verts=fixed.float(), faces=faces.long())
    pix_to_face, zbuf, bary, dists = rasterize_meshes(meshes, image_size=image_size, blur_radius=0.0, faces_per_pixel=1, bin_size=None,
                                                      max_faces_per_bin=None, perspective_correct=False, cull_backfaces=True)
    vis = (pix_to_face > -1).float()
    D = attributes.shape[-1]
    attributes = attributes.clone().view(attributes.shape[0] * attributes.shape[1], 3, D)
    N, H, W, K, _ = bary.shape
    mask = pix_to_face == -1
    p2f = pix_to_face.clone()
    p2f[mask] = 0
    idx = p2f.view(N * H * W * K, 1, 1).expand(N * H * W * K, 3, D)
    vals = attributes.gather(0, idx).view(N, H, W, K, 3, D)
    pix = (bary[..., None] * vals).sum(dim=-2)
    pix[mask] = 0
    pix = pix[:, :, :, 0].permute(0, 3, 1, 2)
    return torch.cat([pix, vis[:, :, :, 0][:, None, :, :]], dim=1), (pix_to_face, zbuf, bary)


def _reference_fill_mouth(images):
    """vr/renderer.py:583-602 verbatim in structure: per image a NumPy copy, cv2.floodFill, back to the device."""
    import cv2
    out = []
    for image in images:
        img = image[0].cpu().numpy() * 255.
        cp = img.copy()
        h, w = img.shape[:2]
        cv2.floodFill(cp, np.zeros([h + 2, w + 2], np.uint8), (0, 0), (255, 255, 255), (0, 0, 0), (254, 254, 254), cv2.FLOODFILL_FIXED_RANGE)
        out.append((torch.tensor(cp).to(images.device).to(torch.float32) / 127.5 - 1).unsqueeze(0))
    mm = torch.stack(out, 0)
    mm = ((mm * 2. - 1.) * -1. + 1.) / 2.
    return (images + mm).clip(0, 1)


@pytest.fixture
def shims():
    import sys
    import next3d_amd
    saved = {k: sys.modules.get(k) for k in ('cv2', 'pytorch3d', 'pytorch3d.io', 'pytorch3d.structures', 'pytorch3d.renderer', 'pytorch3d.renderer.mesh')}
    next3d_amd.install_dropin(third_party=True)
    yield
    for k, v in saved.items():
        if v is None:
            sys.modules.pop(k, None)
        else:
            sys.modules[k] = v


@pytest.mark.parametrize('cull', [True, False])
def test_third_party_shims_rasterize_meshes_matches_oracle(dev, shims, cull):
    """pytorch3d.renderer.mesh.rasterize_meshes (shim -> n3d_rasterize_meshes) against oracle/raster_ref.c on a triangle soup, batch 2,
    shared and per-sample face tables: packed pix_to_face identical, zbuf / bary bit-equal on the covered pixels."""
    from pytorch3d.renderer.mesh import rasterize_meshes
    from pytorch3d.structures import Meshes
    v0, faces, _ = _soup(77)
    v = torch.cat([v0, v0.flip(1) * 0.9], 0) * 5.0
    v[..., 2] += 10
    p_ref, z_ref, b_ref = raster.rasterize_meshes(v, faces, image_size=128, cull_backfaces=cull)
    for f_in in (faces[None].expand(2, -1, -1), faces[None].repeat(2, 1, 1)):
        p, z, b, d = rasterize_meshes(Meshes(verts=v.to(dev), faces=f_in.to(dev)), image_size=128, blur_radius=0.0, faces_per_pixel=1, bin_size=None,
                                      max_faces_per_bin=None, perspective_correct=False, cull_backfaces=cull)
        assert p.dtype == torch.int64 and tuple(p.shape) == (2, 128, 128, 1) and tuple(b.shape) == (2, 128, 128, 1, 3) and tuple(d.shape) == (2, 128, 128, 1)
        assert 0.05 < float((p_ref >= 0).float().mean()) < 0.95
        assert torch.equal(p.cpu(), p_ref) and torch.equal(z.cpu(), z_ref) and torch.equal(b.cpu(), b_ref)
    with pytest.raises(RuntimeError):
        rasterize_meshes(Meshes(verts=v.to(dev), faces=faces[None].expand(2, -1, -1).to(dev)), image_size=128, blur_radius=0.0, faces_per_pixel=2)


def test_third_party_shims_rasterize_meshes_rejects_faces_behind_the_camera_plane(dev, shims):
    """PyTorch3D's `z_invalid`: CheckPointOutsideBoundingBox reports every pixel outside the box of a face whose MINIMUM z is below kEpsilon = 1e-8
    (a vertex at or behind the camera plane), so such a face covers nothing — also when its other vertices are in front (zmax >= 0 alone would keep it).
    A soup pushed across z = 0: faces straddling the plane vanish in the kernel exactly as in oracle/raster_ref.c; the rest is bit-equal."""
    from pytorch3d.renderer.mesh import rasterize_meshes
    from pytorch3d.structures import Meshes
    v0, faces, _ = _soup(78)
    v = v0 * 5.0
    v[..., 2] = v[..., 2] * 6.0 + 0.05                                    # z roughly in [-0.5, 0.6]: behind, straddling and in front of the plane
    zf = v[0][faces][..., 2]                                              # [F, 3]
    straddle = (zf.min(1).values < 1e-8) & (zf.max(1).values >= 0)
    assert int(straddle.sum()) >= 5 and int((zf.min(1).values >= 1e-8).sum()) >= 5
    p_ref, z_ref, b_ref = raster.rasterize_meshes(v, faces, image_size=128, cull_backfaces=False)
    p, z, b, _ = rasterize_meshes(Meshes(verts=v.to(dev), faces=faces[None].to(dev)), image_size=128, blur_radius=0.0, faces_per_pixel=1, bin_size=None,
                                  max_faces_per_bin=None, perspective_correct=False, cull_backfaces=False)
    drawn = set(p_ref[p_ref >= 0].unique().tolist())
    assert len(drawn) >= 3 and not (drawn & set(straddle.nonzero().flatten().tolist()))            # no face with a vertex behind the plane is ever drawn
    assert torch.equal(p.cpu(), p_ref) and torch.equal(z.cpu(), z_ref) and torch.equal(b.cpu(), b_ref)


def test_third_party_shims_reference_call_sequence_equals_fused_kernel(dev, shims):
    """The reference's OWN rasterisation code path — rasterize() per view (triplane_next3d.py:190-222) -> Pytorch3dRasterizer.forward ->
    grid_sample of the uv mask -> fill_mouth with cv2.floodFill on NumPy copies — running on the shims, against the fused
    n3d_rasterize_views the reloaded generator uses: uv grid and hole-filled alpha of all four views bit for bit."""
    v, faces, uv = _cube()
    v = torch.cat([v, v * torch.tensor([0.8, 1.1, 0.7])], 0)
    lms = _rand((2, 68, 3), 6, -0.1, 0.1)
    mask = torch.ones(32, 32)
    mask[6:10, 7:12] = 0; mask[20:27, 18:22] = 0; mask[14:16, 3:5] = 0.5; mask[:5, 14:17] = 0; mask[28:, :] = 0
    g_k, a_k, _ = _gpu_views(dev, v, lms, faces, uv, mask, 256, True, -1)
    N = v.shape[0]
    for k, view in enumerate(VIEWS):
        tv = raster.orth_project(v, raster.angle2matrix(view), ogen.ORTH_SHIFT, ogen.ORTH_SCALE)      # the reference's own torch arithmetic
        tv[:, :, 2] = tv[:, :, 2] + 10
        rendering, _ = _reference_rasterizer_forward(tv.to(dev), faces[None].expand(N, -1, -1).to(dev), uv[None].expand(N, -1, -1, -1).to(dev), 256)
        grid = rendering[:, :-1].permute(0, 2, 3, 1)[:, :, :, :2]
        alpha = F.grid_sample(mask[None, None].expand(N, -1, -1, -1).to(dev), grid, align_corners=False) * rendering[:, -1:]
        alpha = _reference_fill_mouth(alpha)
        vis = rendering[:, -1].cpu() > 0
        assert int(vis.sum()) > 1000
        # the vertex transform runs in torch here and inside the kernel there: same values up to its rounding, coverage may differ on razor edges
        same_cov = float(((a_k[:, k] > 0) == (alpha[:, 0].cpu() > 0)).float().mean())
        guv = (grid.cpu() - g_k[:, k]).abs().amax(-1)
        print(f'view {k}: coverage agreement {same_cov:.6f}, uv max diff on covered pixels {float(guv[vis].max()):.2e}, alpha max diff {_md(alpha[:, 0], a_k[:, k]):.2e}')
        assert same_cov >= 0.9995 and float((guv > 1e-4).float().mean()) <= 1e-3
        assert float(((alpha[:, 0].cpu() - a_k[:, k]).abs() > 1e-4).float().mean()) <= 1e-3


def test_third_party_shims_flood_fill_matches_oracle(dev, shims):
    """cv2.floodFill (shim -> n3d_flood_fill) against oracle/raster_ref.c's restatement on images with thresholds, spirals and a
    blocked seed: NumPy images in place (the reference's usage) and device tensors."""
    import cv2
    rng = np.random.RandomState(3)
    imgs = []
    a = (rng.rand(256, 256) * 300).astype(np.float32); imgs.append(a)                           # noise around the 254 threshold
    b = np.zeros((200, 256), np.float32); b[20:180, 30:34] = 255; b[20:24, 30:200] = 255; b[60:64, 60:256] = 255; imgs.append(b)   # walls, ragged size
    s = np.zeros((256, 256), np.float32)
    for r in range(8, 120, 8):                                                                    # nested rings with gaps: a long winding path
        s[r, r:256 - r] = 255; s[255 - r, r:256 - r] = 255; s[r:256 - r, r] = 255; s[r:256 - r, 255 - r] = 255
        s[r + (3 if (r // 8) % 2 else 0), 128] = 0 if (r // 8) % 2 == 0 else 255
        s[r if (r // 8) % 2 == 0 else 255 - r, 100] = 0
    imgs.append(s)
    c = np.full((64, 64), 255, np.float32); c[0, 0] = 255; imgs.append(c)                        # everything within range of the seed
    for im in imgs:
        ref = im.copy()
        raster.floodfill_fixed_range(ref, 255.0, 0.0, 254.0)
        got = im.copy()
        cv2.floodFill(got, np.zeros([im.shape[0] + 2, im.shape[1] + 2], np.uint8), (0, 0), (255, 255, 255), (0, 0, 0), (254, 254, 254), cv2.FLOODFILL_FIXED_RANGE)
        assert np.array_equal(got, ref), int((got != ref).sum())
        t = torch.from_numpy(im.copy()).to(dev)
        cv2.floodFill(t, None, (0, 0), 255, 0, 254, cv2.FLOODFILL_FIXED_RANGE)
        assert np.array_equal(t.cpu().numpy(), ref)
    with pytest.raises(RuntimeError):
        cv2.floodFill(imgs[0].copy(), None, (3, 3), 255, 0, 254, cv2.FLOODFILL_FIXED_RANGE)


@pytest.mark.parametrize('rows,D,pitch', [(1, 512, 1024), (5, 512, 1024), (3, 100, 100)])
def test_normalize_2nd_moment_float32_and_float64_input(dev, rows, D, pitch):
    """normalize_2nd_moment (tat/networks_stylegan2.py:27-29) on float32 rows, and on the scripts' float64 z (gen_samples_next3d.py:165), whose
    `z.to(torch.float32)` (:239) happens inside the kernel: bit-identical to converting first; both against the torch expression."""
    from next3d_amd import _lib
    L = _lib.lib()
    z64 = torch.from_numpy(np.random.RandomState(rows * 7 + D).randn(rows, D))                    # float64, as the scripts draw it
    z32 = z64.float()
    want = z32 * (z32.square().mean(dim=1, keepdim=True) + 1e-8).rsqrt()
    ya = torch.full((rows, pitch), -7.0, device=dev)
    yb = torch.full((rows, pitch), -7.0, device=dev)
    a, b = z32.to(dev), z64.to(dev)
    _lib.check(L.n3d_normalize_2nd_moment(_lib.ptr(a), _lib.ptr(ya), rows, D, pitch, 1e-8, _lib.stream()))
    _lib.check(L.n3d_normalize_2nd_moment_f64(_lib.ptr(b), _lib.ptr(yb), rows, D, pitch, 1e-8, _lib.stream()))
    assert torch.equal(ya, yb)
    assert _md(ya[:, :D], want) <= 2e-6 and (pitch == D or bool((ya[:, D:] == -7.0).all()))
    assert L.n3d_normalize_2nd_moment_f64(None, _lib.ptr(yb), rows, D, pitch, 1e-8, _lib.stream()) != 0 and 'normalize_2nd_moment_f64' in L.n3d_last_error().decode()


def test_mapping_takes_float64_latents_without_a_conversion_pass(dev):
    """G.mapping(z float64) == G.mapping(z.float()) bit for bit (the conversion of MappingNetwork.forward :239 is folded into the first kernel)."""
    from next3d_amd import demo
    G, _ = demo.build_generator(dev)
    z, c, c_cond, v = demo.demo_batch([0, 1, 2], device=dev)
    assert z.dtype == torch.float64
    assert torch.equal(G.mapping(z, c_cond, truncation_psi=0.7, truncation_cutoff=14), G.mapping(z.float(), c_cond, truncation_psi=0.7, truncation_cutoff=14))
