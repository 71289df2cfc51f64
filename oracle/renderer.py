"""oracle/renderer.py — CPU fp32 restatement of the tri-plane volume renderer
(TEST INFRASTRUCTURE).  Paths relative to /root/reference; `vr/` =
training_avatar_texture/volumetric_rendering/.

Randomness is an explicit input (`jitter`, `u`): the reference draws it with
`torch.rand_like` (vr/renderer.py:205) and `torch.rand` (:252) from the device RNG, which
no other implementation can reproduce; the pin script monkey-patches those two calls to
return the same tensors.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import ops

# vr/renderer.py:30-44 (generate_planes) — NOTE the 3rd plane differs from upstream EG3D.
PLANE_AXES = torch.tensor([[[1, 0, 0], [0, 1, 0], [0, 0, 1]],
                           [[1, 0, 0], [0, 0, 1], [0, 1, 0]],
                           [[0, 0, 1], [0, 1, 0], [1, 0, 0]]], dtype=torch.float32)


def ray_sampler(cam2world, intrinsics, resolution):
    """vr/ray_sampler.py:24-63 (RaySampler.forward)."""
    N, M = cam2world.shape[0], resolution ** 2
    cam_locs = cam2world[:, :3, 3]
    fx, fy = intrinsics[:, 0, 0], intrinsics[:, 1, 1]
    cx, cy = intrinsics[:, 0, 2], intrinsics[:, 1, 2]
    sk = intrinsics[:, 0, 1]
    ar = torch.arange(resolution, dtype=torch.float32)
    uv = torch.stack(torch.meshgrid(ar, ar, indexing='ij')) * (1. / resolution) + (0.5 / resolution)
    uv = uv.flip(0).reshape(2, -1).transpose(1, 0).unsqueeze(0).repeat(N, 1, 1)
    x_cam, y_cam = uv[:, :, 0].view(N, -1), uv[:, :, 1].view(N, -1)
    z_cam = torch.ones((N, M))
    x_lift = (x_cam - cx[:, None] + cy[:, None] * sk[:, None] / fy[:, None]
              - sk[:, None] * y_cam / fy[:, None]) / fx[:, None] * z_cam
    y_lift = (y_cam - cy[:, None]) / fy[:, None] * z_cam
    pts = torch.stack((x_lift, y_lift, z_cam, torch.ones_like(z_cam)), dim=-1)
    world = torch.bmm(cam2world, pts.permute(0, 2, 1)).permute(0, 2, 1)[:, :, :3]
    dirs = F.normalize(world - cam_locs[:, None, :], dim=2)
    origins = cam_locs.unsqueeze(1).repeat(1, M, 1)
    return origins, dirs


def sample_from_planes(planes, coordinates, box_warp):
    """vr/renderer.py:46-72 (project_onto_planes + sample_from_planes)."""
    N, n_planes, C, H, W = planes.shape
    M = coordinates.shape[1]
    coordinates = (2 / box_warp) * coordinates
    coords = coordinates.unsqueeze(1).expand(-1, n_planes, -1, -1).reshape(N * n_planes, M, 3)
    inv = torch.linalg.inv(PLANE_AXES).unsqueeze(0).expand(N, -1, -1, -1).reshape(N * n_planes, 3, 3)
    proj = torch.bmm(coords, inv)[..., :2].unsqueeze(1)
    out = F.grid_sample(planes.reshape(N * n_planes, C, H, W), proj.float(), mode='bilinear',
                        padding_mode='zeros', align_corners=False)
    return out.permute(0, 3, 2, 1).reshape(N, n_planes, M, C)


def osg_decoder(P, prefix, feats):
    """training_avatar_texture/triplane_next3d.py:359-371 (OSGDecoder.forward);
    FullyConnectedLayer from training/networks_stylegan2.py:96."""
    x = feats.mean(1)
    N, M, C = x.shape
    x = x.reshape(N * M, C)
    x = ops.fully_connected(x, P[f'{prefix}.net.0.weight'], P[f'{prefix}.net.0.bias'])
    x = F.softplus(x)
    x = ops.fully_connected(x, P[f'{prefix}.net.2.weight'], P[f'{prefix}.net.2.bias'])
    x = x.view(N, M, -1)
    rgb = torch.sigmoid(x[..., 1:]) * (1 + 2 * 0.001) - 0.001
    return rgb, x[..., 0:1]


def ray_limits_box(rays_o, rays_d, box_side_length):
    """vr/math_utils.py:46-100 (get_ray_limits_box): slab test against [-side/2, side/2]^3 -> (tmin, tmax) [..., 1]; -1 / -2 for misses."""
    shape = rays_o.shape
    o, d = rays_o.reshape(-1, 3), rays_d.reshape(-1, 3)
    h = box_side_length / 2
    bounds = torch.tensor([[-h, -h, -h], [h, h, h]], dtype=o.dtype)
    valid = torch.ones(o.shape[:-1], dtype=bool)
    inv = 1 / d
    sign = (inv < 0).long()
    tmin = (bounds.index_select(0, sign[..., 0])[..., 0] - o[..., 0]) * inv[..., 0]
    tmax = (bounds.index_select(0, 1 - sign[..., 0])[..., 0] - o[..., 0]) * inv[..., 0]
    tymin = (bounds.index_select(0, sign[..., 1])[..., 1] - o[..., 1]) * inv[..., 1]
    tymax = (bounds.index_select(0, 1 - sign[..., 1])[..., 1] - o[..., 1]) * inv[..., 1]
    valid[torch.logical_or(tmin > tymax, tymin > tmax)] = False
    tmin, tmax = torch.max(tmin, tymin), torch.min(tmax, tymax)
    tzmin = (bounds.index_select(0, sign[..., 2])[..., 2] - o[..., 2]) * inv[..., 2]
    tzmax = (bounds.index_select(0, 1 - sign[..., 2])[..., 2] - o[..., 2]) * inv[..., 2]
    valid[torch.logical_or(tmin > tzmax, tzmin > tmax)] = False
    tmin, tmax = torch.max(tmin, tzmin), torch.min(tmax, tzmax)
    tmin[~valid] = -1
    tmax[~valid] = -2
    return tmin.reshape(*shape[:-1], 1), tmax.reshape(*shape[:-1], 1)


def ray_march(colors, densities, depths, white_back=False):
    """vr/ray_marcher.py:27-66 (MipRayMarcher2.run_forward), clamp_mode='softplus'."""
    deltas = depths[:, :, 1:] - depths[:, :, :-1]
    colors_mid = (colors[:, :, :-1] + colors[:, :, 1:]) / 2
    dens_mid = F.softplus((densities[:, :, :-1] + densities[:, :, 1:]) / 2 - 1)
    depths_mid = (depths[:, :, :-1] + depths[:, :, 1:]) / 2
    alpha = 1 - torch.exp(-(dens_mid * deltas))
    shifted = torch.cat([torch.ones_like(alpha[:, :, :1]), 1 - alpha + 1e-10], -2)
    weights = alpha * torch.cumprod(shifted, -2)[:, :, :-1]
    rgb = torch.sum(weights * colors_mid, -2)
    wtot = weights.sum(2)
    depth = torch.sum(weights * depths_mid, -2) / wtot
    depth = torch.nan_to_num(depth, float('inf'))
    depth = torch.clamp(depth, torch.min(depths), torch.max(depths))
    if white_back:                                               # :56-57
        rgb = rgb + 1 - wtot
    return rgb * 2 - 1, depth, weights


def sample_pdf(bins, weights, u, eps=1e-5):
    """vr/renderer.py:229-268 (sample_pdf) with `u` supplied."""
    n_samples = weights.shape[1]
    weights = weights + eps
    pdf = weights / torch.sum(weights, -1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[:, :1]), cdf], -1)
    u = u.contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = torch.clamp_min(inds - 1, 0)
    above = torch.clamp_max(inds, n_samples)
    idx = torch.stack([below, above], -1).view(u.shape[0], -1)
    cdf_g = torch.gather(cdf, 1, idx).view(*u.shape, 2)
    bins_g = torch.gather(bins, 1, idx).view(*u.shape, 2)
    denom = cdf_g[..., 1] - cdf_g[..., 0]
    denom[denom < eps] = 1
    return bins_g[..., 0] + (u - cdf_g[..., 0]) / denom * (bins_g[..., 1] - bins_g[..., 0])


def sample_importance(z_vals, weights, u):
    """vr/renderer.py:209-227 (sample_importance)."""
    B, R, S, _ = z_vals.shape
    z = z_vals.reshape(B * R, S)
    w = weights.reshape(B * R, -1)
    w = F.max_pool1d(w.unsqueeze(1).float(), 2, 1, padding=1)
    w = F.avg_pool1d(w, 2, 1).squeeze(1)
    w = w + 0.01
    z_mid = 0.5 * (z[:, :-1] + z[:, 1:])
    return sample_pdf(z_mid, w[:, 1:-1], u).reshape(B, R, u.shape[1], 1)


def importance_renderer(P, decoder_prefix, planes, ray_o, ray_d, opts, jitter, u, noise=None, fine_depths_out=None):
    """vr/renderer.py:95-147 (ImportanceRenderer.forward): fixed ray_start / ray_end, or both 'auto' (:98-106); sample_stratified with
    or without disparity_space_sampling (:184-207); white_back (ray_marcher.py:56-57); density_noise (:152-153).

    jitter: [N, R, S_coarse, 1] uniform [0,1)   (replaces torch.rand_like, :205)
    u     : [N*R, S_importance] uniform [0,1)    (replaces torch.rand, :252)
    noise : (coarse [N, R*Sc, 1], fine [N, R*Sf, 1]) normal draws (replace torch.randn_like, :153) when opts['density_noise'] > 0
    """
    N, R, _ = ray_o.shape
    Sc, Sf = opts['depth_resolution'], opts['depth_resolution_importance']
    t0, t1 = opts['ray_start'], opts['ray_end']
    white = opts.get('white_back', False)
    if t0 == t1 == 'auto':
        rs, re = ray_limits_box(ray_o, ray_d, opts['box_warp'])
        ok = re > rs
        if torch.any(ok).item():
            rs[~ok] = rs[ok].min()
            re[~ok] = rs[ok].max()
        steps = torch.arange(Sc, dtype=torch.float32) / (Sc - 1)                     # math_utils.linspace
        depths_c = (rs[None] + steps.reshape(-1, 1, 1, 1) * (re - rs)[None]).permute(1, 2, 0, 3)
        depths_c = depths_c + jitter * ((re - rs) / (Sc - 1))[..., None]
    elif opts.get('disparity_space_sampling', False):
        d = torch.linspace(0, 1, Sc).reshape(1, 1, Sc, 1).repeat(N, R, 1, 1)
        d = d + jitter * (1 / (Sc - 1))
        depths_c = 1. / (1. / t0 * (1. - d) + 1. / t1 * d)
    else:
        depths_c = torch.linspace(t0, t1, Sc).reshape(1, 1, Sc, 1).repeat(N, R, 1, 1)
        depths_c = depths_c + jitter * ((t1 - t0) / (Sc - 1))
    amp = opts.get('density_noise', 0) or 0

    def run(depths, S, nz):
        pts = (ray_o.unsqueeze(-2) + depths * ray_d.unsqueeze(-2)).reshape(N, -1, 3)
        feats = sample_from_planes(planes, pts, opts['box_warp'])
        rgb, sigma = osg_decoder(P, decoder_prefix, feats)
        if amp > 0:
            sigma = sigma + nz * amp
        return rgb.reshape(N, R, S, -1), sigma.reshape(N, R, S, 1)

    _rm = ray_march
    ray_march_ = lambda c_, d_, z_: _rm(c_, d_, z_, white)
    col_c, den_c = run(depths_c, Sc, noise[0] if amp > 0 else None)
    if Sf > 0:
        _, _, w = ray_march_(col_c, den_c, depths_c)
        depths_f = sample_importance(depths_c, w, u)
        if fine_depths_out is not None:          # (tests: the importance depths, to teacher-force the kernel under test with — n3d_render_opts.fine_depths_in)
            fine_depths_out.append(depths_f)
        col_f, den_f = run(depths_f, Sf, noise[1] if amp > 0 else None)
        all_d = torch.cat([depths_c, depths_f], -2)
        all_c = torch.cat([col_c, col_f], -2)
        all_s = torch.cat([den_c, den_f], -2)
        _, idx = torch.sort(all_d, dim=-2)                       # unify_samples :164-182
        all_d = torch.gather(all_d, -2, idx)
        all_c = torch.gather(all_c, -2, idx.expand(-1, -1, -1, all_c.shape[-1]))
        all_s = torch.gather(all_s, -2, idx)
        rgb, depth, w = ray_march_(all_c, all_s, all_d)
    else:
        rgb, depth, w = ray_march_(col_c, den_c, depths_c)
    return rgb, depth, w.sum(2)
