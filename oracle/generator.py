"""oracle/generator.py — CPU fp32 restatement of TriPlaneGenerator.mapping / synthesis
(TEST INFRASTRUCTURE).  Follows /root/reference/training_avatar_texture/triplane_next3d.py.

`P` is the flat state dict (reference parameter names).  Non-parameter state the reference keeps
as plain attributes (uv_face_mask, orth_scale/orth_shift, rendering_kwargs) is passed explicitly.
"""
import torch
import torch.nn.functional as F

from . import networks, raster, renderer

RENDERING_VIEWS = [[0, 0, 0], [0, 90, 0], [0, -90, 0], [90, 0, 0]]       # triplane_next3d.py:140-145
ORTH_SCALE = torch.tensor([[5.0]])                                       # :105
ORTH_SHIFT = torch.tensor([[0, -0.01, -0.01]])                           # :106
NUM_WS_HALF = 14                                                         # texture_backbone.num_ws (256² net)

DEFAULT_RENDERING_KWARGS = dict(                                         # train_next3d.py:313-341 (cfg=ffhq)
    depth_resolution=48, depth_resolution_importance=48, ray_start=2.25, ray_end=3.3, box_warp=1,
    c_scale=1.0, c_gen_conditioning_zero=True, avg_camera_radius=2.7, avg_camera_pivot=[0, 0, 0.2])


def mapping(P, z, c, rendering_kwargs, truncation_psi=1.0, truncation_cutoff=None):
    """triplane_next3d.py:111-115."""
    if rendering_kwargs['c_gen_conditioning_zero']:
        c = torch.zeros_like(c)
    c = c[:, :25]
    num_layers = sum(1 for k in P if k.startswith('backbone.mapping.fc') and k.endswith('.weight'))       # (mapping_kwargs.num_layers: read off the parameters)
    return networks.mapping_network(P, 'backbone.mapping', z, c * rendering_kwargs.get('c_scale', 0),
                                    num_ws=2 * NUM_WS_HALF, num_layers=num_layers, truncation_psi=truncation_psi,
                                    truncation_cutoff=truncation_cutoff)


def rasterize(P, v, lms, textures, uv_face_mask):
    """triplane_next3d.py:190-230 (rasterize)."""
    N = v.shape[0]
    rend, alphas, lm2ds = [], [], []
    faces = P['faces'][0][:, [0, 2, 1]]
    attrs = P['face_uvcoords'][:, :, [0, 2, 1]]
    for view in RENDERING_VIEWS:
        tform = raster.angle2matrix(view)
        tv = raster.orth_project(v, tform, ORTH_SHIFT, ORTH_SCALE)
        tv[:, :, 2] = tv[:, :, 2] + 10
        tl = raster.orth_project(lms, tform, ORTH_SHIFT, ORTH_SCALE)[:, :, :2]
        rendering = raster.pytorch3d_rasterizer(tv, faces, attrs, 256)
        alpha = rendering[:, -1:, :, :]
        grid = rendering[:, :-1].permute(0, 2, 3, 1)[:, :, :, :2]
        mask = F.grid_sample(uv_face_mask.expand(N, -1, -1, -1), grid, align_corners=False)
        alpha = raster.fill_mouth(mask * alpha)
        rend.append(F.grid_sample(textures, grid, align_corners=False))
        alphas.append(alpha)
        lm2ds.append(tl)
    side = rend[1] + rend[2]
    alpha_side = (alphas[1].bool() | alphas[1].bool()).float()           # (sic) :226
    return [rend[0], side, rend[3]], [alphas[0], alpha_side, alphas[3]], lm2ds


def blended_planes(P, ws, v, uv_face_mask, noise_mode='const', st=None, net_kw=None):
    """triplane_next3d.py:119-174 (synthesis) == :236-276 (sample) == :282-322 (sample_mixed): everything up to the
    blended tri-planes [N,3,32,256,256].  Returns (planes, eg3d_ws); `st` (a dict) collects the stage tensors."""
    st = {} if st is None else st
    net_kw = net_kw or {}           # fp16_resolution / conv_clamp / cpu_rounding of the four backbones (num_fp16_res > 0: legacy.py:49-59)
    v, lms = v[:, :5023], v[:, 5023:]
    N = ws.shape[0]
    eg3d_ws, texture_ws = ws[:, :NUM_WS_HALF], ws[:, NUM_WS_HALF:]

    textures = networks.synthesis_network(P, 'texture_backbone.synthesis', texture_ws, noise_mode=noise_mode, **net_kw)
    st['textures'] = textures
    rend, alphas, lm2ds = rasterize(P, v, lms, textures, uv_face_mask)
    st['rendering_front'], st['rendering_side'], st['rendering_top'] = rend
    st['alpha'] = torch.cat(alphas, 1)

    front = rend[0]
    mm = raster.gen_mouth_mask(lm2ds[0])
    st['mouth_mask'] = torch.from_numpy(mm.copy())
    crops = [front[i:i + 1, :, m[0]:m[1], m[2]:m[3]] for i, m in enumerate(mm)]
    crops = torch.cat([F.interpolate(cr, size=(64, 64), mode='bilinear', antialias=True) for cr in crops], 0)
    st['rendering_mouth'] = crops
    mouths = networks.styleunet_synthesis(P, 'mouth_backbone.synthesis', crops, eg3d_ws, in_size=64,
                                          final_size=4, num_cond_res=64, noise_mode=noise_mode, **net_kw)
    st['mouths_plane'] = mouths
    stitch = []
    for i, m in enumerate(mm):
        dummy = front[i:i + 1].clone()
        s = int(m[1] - m[0])
        dummy[:, :, m[0]:m[1], m[2]:m[3]] = F.interpolate(mouths[i:i + 1], size=(s, s), mode='bilinear',
                                                          antialias=True)
        stitch.append(dummy)
    stitch = torch.cat(stitch, 0)
    st['rendering_stitch_in'] = stitch
    stitch = networks.styleunet_synthesis(P, 'neural_blending.synthesis', stitch, eg3d_ws, in_size=256,
                                          final_size=32, num_cond_res=256, noise_mode=noise_mode, **net_kw)
    st['rendering_stitch'] = stitch

    static = networks.synthesis_network(P, 'backbone.synthesis', eg3d_ws, noise_mode=noise_mode, **net_kw)
    static = static.view(N, 3, 32, static.shape[-2], static.shape[-1])
    st['static_plane'] = static
    alpha = torch.cat(alphas, 1).unsqueeze(2)
    dyn = torch.cat((stitch, rend[1], rend[2]), 1).view(*static.shape)
    planes = dyn * alpha + static * (1 - alpha)
    st['blended_planes'] = planes
    return planes, eg3d_ws


def run_model(P, planes, coordinates, rendering_kwargs):
    """vr/renderer.py:149-155 (ImportanceRenderer.run_model): tri-plane features + decoder at arbitrary points
    (density_noise is 0 at inference) -> {'rgb' [N,M,32], 'sigma' [N,M,1]}."""
    feats = renderer.sample_from_planes(planes, coordinates, rendering_kwargs['box_warp'])
    rgb, sigma = renderer.osg_decoder(P, 'decoder', feats)
    return {'rgb': rgb, 'sigma': sigma}


def sample_mixed(P, coordinates, ws, v, uv_face_mask, rendering_kwargs, noise_mode='const'):
    """triplane_next3d.py:278-322 (sample_mixed; `sample` :232-276 is the same after `mapping`).  The view directions the
    reference passes are unused by OSGDecoder (:359)."""
    planes, _ = blended_planes(P, ws, v, uv_face_mask, noise_mode)
    return run_model(P, planes, coordinates, rendering_kwargs)


def synthesis(P, ws, c, v, uv_face_mask, rendering_kwargs, jitter, u, neural_rendering_resolution=64,
              noise_mode='const', return_stages=False, force_fp32=True, net_kw=None):
    """triplane_next3d.py:117-188 (synthesis).  `jitter`/`u`: see oracle/renderer.py.  force_fp32=False: the float16
    super-resolution blocks the reference runs on a GPU by default (oracle/networks.py::synthesis_block_fp16)."""
    st = {}
    N = ws.shape[0]
    cam2world = c[:, :16].view(-1, 4, 4)
    intrinsics = c[:, 16:25].view(-1, 3, 3)
    R = neural_rendering_resolution
    ray_o, ray_d = renderer.ray_sampler(cam2world, intrinsics, R)
    planes, eg3d_ws = blended_planes(P, ws, v, uv_face_mask, noise_mode, st, net_kw)

    feat, depth, wsum = renderer.importance_renderer(P, 'decoder', planes, ray_o, ray_d, rendering_kwargs,
                                                     jitter, u)
    feature_image = feat.permute(0, 2, 1).reshape(N, feat.shape[-1], R, R).contiguous()
    depth_image = depth.permute(0, 2, 1).reshape(N, 1, R, R)
    rgb = feature_image[:, :3]
    st['feature_image'] = feature_image
    sr_class = str(rendering_kwargs.get('superresolution_module', 'SuperresolutionHybrid8XDC')).rsplit('.', 1)[-1]       # tat/triplane_next3d.py:66
    alias = {}
    sr = networks.superresolution(P, 'superresolution', rgb, feature_image, eg3d_ws, force_fp32=force_fp32, sr_class=sr_class, aliased_raw=alias)
    out = {'image': sr, 'image_raw': alias.get('image_raw', rgb), 'image_depth': depth_image}       # (4X / 2X without a resize: see networks.superresolution)
    return (out, st) if return_stages else out
