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


def ray_march(colors, densities, depths):
    """vr/ray_marcher.py:27-66 (MipRayMarcher2.run_forward), clamp_mode='softplus', no white_back."""
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


def importance_renderer(P, decoder_prefix, planes, ray_o, ray_d, opts, jitter, u):
    """vr/renderer.py:95-147 (ImportanceRenderer.forward), fixed ray_start/ray_end branch.

    jitter: [N, R, S_coarse, 1] uniform [0,1)   (replaces torch.rand_like, :205)
    u     : [N*R, S_importance] uniform [0,1)    (replaces torch.rand, :252)
    """
    N, R, _ = ray_o.shape
    Sc, Sf = opts['depth_resolution'], opts['depth_resolution_importance']
    t0, t1 = opts['ray_start'], opts['ray_end']
    depths_c = torch.linspace(t0, t1, Sc).reshape(1, 1, Sc, 1).repeat(N, R, 1, 1)
    depths_c = depths_c + jitter * ((t1 - t0) / (Sc - 1))

    def run(depths, S):
        pts = (ray_o.unsqueeze(-2) + depths * ray_d.unsqueeze(-2)).reshape(N, -1, 3)
        feats = sample_from_planes(planes, pts, opts['box_warp'])
        rgb, sigma = osg_decoder(P, decoder_prefix, feats)
        return rgb.reshape(N, R, S, -1), sigma.reshape(N, R, S, 1)

    col_c, den_c = run(depths_c, Sc)
    if Sf > 0:
        _, _, w = ray_march(col_c, den_c, depths_c)
        depths_f = sample_importance(depths_c, w, u)
        col_f, den_f = run(depths_f, Sf)
        all_d = torch.cat([depths_c, depths_f], -2)
        all_c = torch.cat([col_c, col_f], -2)
        all_s = torch.cat([den_c, den_f], -2)
        _, idx = torch.sort(all_d, dim=-2)                       # unify_samples :164-182
        all_d = torch.gather(all_d, -2, idx)
        all_c = torch.gather(all_c, -2, idx.expand(-1, -1, -1, all_c.shape[-1]))
        all_s = torch.gather(all_s, -2, idx)
        rgb, depth, w = ray_march(all_c, all_s, all_d)
    else:
        rgb, depth, w = ray_march(col_c, den_c, depths_c)
    return rgb, depth, w.sum(2)
