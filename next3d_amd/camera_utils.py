"""Camera helpers used by the inference scripts to build the 25-float conditioning vector
`c = [cam2world(16), intrinsics(9)]` (reference camera_utils.py:69-148)."""
import math

import torch


def _normalize(v):
    return v / torch.norm(v, dim=-1, keepdim=True)


def create_cam2world_matrix(forward_vector, origin):
    """y-up, no roll (reference camera_utils.py:117-138)."""
    forward_vector = _normalize(forward_vector)
    up = torch.tensor([0, 1, 0], dtype=torch.float, device=origin.device).expand_as(forward_vector)
    right = -_normalize(torch.cross(up, forward_vector, dim=-1))
    up = _normalize(torch.cross(forward_vector, right, dim=-1))
    n = forward_vector.shape[0]
    rot = torch.eye(4, device=origin.device).unsqueeze(0).repeat(n, 1, 1)
    rot[:, :3, :3] = torch.stack((right, up, forward_vector), dim=-1)
    trans = torch.eye(4, device=origin.device).unsqueeze(0).repeat(n, 1, 1)
    trans[:, :3, 3] = origin
    return trans @ rot


class LookAtPoseSampler:
    """Camera on a sphere of `radius` looking at `lookat_position` (reference camera_utils.py:57-85)."""

    @staticmethod
    def sample(horizontal_mean, vertical_mean, lookat_position, horizontal_stddev=0, vertical_stddev=0, radius=1,
               batch_size=1, device='cpu'):
        h = torch.randn((batch_size, 1), device=device) * horizontal_stddev + horizontal_mean
        v = torch.randn((batch_size, 1), device=device) * vertical_stddev + vertical_mean
        v = torch.clamp(v, 1e-5, math.pi - 1e-5)
        theta = h
        phi = torch.arccos(1 - 2 * (v / math.pi))
        origins = torch.zeros((batch_size, 3), device=device)
        origins[:, 0:1] = radius * torch.sin(phi) * torch.cos(math.pi - theta)
        origins[:, 2:3] = radius * torch.sin(phi) * torch.sin(math.pi - theta)
        origins[:, 1:2] = radius * torch.cos(phi)
        return create_cam2world_matrix(_normalize(lookat_position - origins), origins)


def FOV_to_intrinsics(fov_degrees, device='cpu'):
    """Normalised 3x3 intrinsics, principal point at the centre (reference camera_utils.py:140-148)."""
    focal = float(1 / (math.tan(fov_degrees * 3.14159 / 360) * 1.414))
    return torch.tensor([[focal, 0, 0.5], [0, focal, 0.5], [0, 0, 1]], device=device)


def demo_camera_params(angle_y=0.0, angle_p=-0.2, pivot=(0, 0, 0.2), radius=2.7, fov_deg=18.837, device='cpu'):
    """The (camera, conditioning) pair gen_samples_next3d.py:188-196 builds for one view -> two [1,25] tensors."""
    piv = torch.tensor(pivot, dtype=torch.float32, device=device)
    K = FOV_to_intrinsics(fov_deg, device=device)
    cam = LookAtPoseSampler.sample(math.pi / 2 + angle_y, math.pi / 2 + angle_p, piv, radius=radius, device=device)
    cond = LookAtPoseSampler.sample(math.pi / 2, math.pi / 2, piv, radius=radius, device=device)
    return (torch.cat([cam.reshape(-1, 16), K.reshape(-1, 9)], 1), torch.cat([cond.reshape(-1, 16), K.reshape(-1, 9)], 1))
