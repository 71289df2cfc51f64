"""Synthetic-weights generator + demo inputs shared by bench.py, __graft_entry__.py and the tests.

No pretrained pickle can be downloaded here (reference README.md:40 is a Drive link), so the generator is built with
seeded synthetic weights of the exact next3d_ffhq_512 architecture; the driving mesh / landmarks are the reference's
own demo fixture (data/demo/demo.obj, demo_kpt2d.txt) stored as arrays in tests/golden/demo_inputs.npz.
"""
import os

import numpy as np
import torch

from . import mesh, spec

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO_NPZ = os.path.join(REPO, 'tests', 'golden', 'demo_inputs.npz')

RENDERING_KWARGS = dict(                      # reference train_next3d.py:313-341 (cfg=ffhq)
    image_resolution=512, disparity_space_sampling=False, clamp_mode='softplus', c_gen_conditioning_zero=True,
    c_scale=1.0, superresolution_noise_mode='none', decoder_lr_mul=1.0, sr_antialias=True, depth_resolution=48,
    depth_resolution_importance=48, ray_start=2.25, ray_end=3.3, box_warp=1, avg_camera_radius=2.7,
    avg_camera_pivot=[0, 0, 0.2], superresolution_module='training_avatar_texture.superresolution.SuperresolutionHybrid8XDC')


def camera_label(yaw=0.0, pitch=0.0, pivot=(0.0, 0.0, 0.2), radius=2.7, fov_deg=18.837, focal=None):
    """The 25-float camera label `[cam2world (16, row-major), intrinsics (9)]` the inference scripts feed the generator
    (gen_samples_next3d.py:188-196, gen_videos_next3d.py:133-137: a camera on the sphere of `radius` around `pivot`, at azimuth
    pi/2 + yaw and polar angle pi/2 + pitch, looking at the pivot, no roll; normalised pinhole intrinsics for the field of view).
    Input helper of the demo / benchmark only — parity runs take their cameras from the reference-generated fixtures."""
    azim, polar = np.pi / 2 + yaw, float(np.clip(np.pi / 2 + pitch, 1e-5, np.pi - 1e-5))
    inclination = np.arccos(1.0 - 2.0 * polar / np.pi)               # the reference maps the polar angle through arccos(1 - 2 v / pi)
    eye = radius * np.array([np.sin(inclination) * np.cos(np.pi - azim), np.cos(inclination), np.sin(inclination) * np.sin(np.pi - azim)])
    unit = lambda a: a / np.linalg.norm(a)
    fwd = unit(np.asarray(pivot, dtype=np.float64) - eye)
    right = -unit(np.cross([0.0, 1.0, 0.0], fwd))
    up = unit(np.cross(fwd, right))
    pose = np.eye(4)
    pose[:3, 0], pose[:3, 1], pose[:3, 2], pose[:3, 3] = right, up, fwd, eye
    if focal is None:
        focal = 1.0 / (np.tan(fov_deg * 3.14159 / 360.0) * 1.414)    # the scripts' constants (3.14159, 1.414), kept for the same numbers
    K = np.array([[focal, 0.0, 0.5], [0.0, focal, 0.5], [0.0, 0.0, 1.0]])
    return torch.from_numpy(np.concatenate([pose.reshape(-1), K.reshape(-1)]).astype(np.float32))[None]


def demo_camera_params(angle_y=0.0, angle_p=-0.2):
    """(camera label, conditioning label) of one view as gen_samples_next3d.py:188-196 builds them -> two [1,25] tensors."""
    return camera_label(angle_y, angle_p), camera_label(0.0, 0.0)


def demo_arrays():
    return np.load(DEMO_NPZ)


def build_generator(device, seed=0, rendering_kwargs=None, force_fp16=False, channel_base=32768, channel_max=512, mapping_layers=2):
    """channel_base / channel_max: the backbones' widths (train_next3d.py --cbase / --cmax; the ffhq-512 pickle: 32768 / 512).  force_fp16: the generator as legacy.load_network_pkl(force_fp16=True) rebuilds it (legacy.py:49-59): num_fp16_res = 4 and
    conv_clamp = 256 in all four backbones."""
    from .generator import TriPlaneGenerator
    d = demo_arrays()
    topo = (d['faces'], d['uvs'], d['uvfaces'])
    G = TriPlaneGenerator(512, 25, 512, 512, 3, topo, sr_num_fp16_res=4, mapping_kwargs=dict(num_layers=mapping_layers),
                          rendering_kwargs=dict(rendering_kwargs or RENDERING_KWARGS),
                          sr_kwargs=dict(channel_base=32768, channel_max=512, fused_modconv_default='inference_only'),
                          uv_face_mask=mesh.synthetic_uv_face_mask(), channel_base=channel_base, channel_max=channel_max,
                          fused_modconv_default='inference_only', num_fp16_res=4 if force_fp16 else 0, conv_clamp=256 if force_fp16 else None)
    sd = spec.synthetic_state_dict(seed, channel_base=channel_base, channel_max=channel_max, mapping_layers=mapping_layers)
    sd.update(mesh.mesh_buffers(*topo))
    G.load_state_dict(sd, strict=True)
    return G.eval().requires_grad_(False).to(device), sd


def demo_batch(seeds, yaws=None, pitch=-0.2, device='cpu'):
    """(z [N,512] f64, c [N,25], c_cond [N,25], v [N,5091,3]) exactly as gen_samples_next3d.py:165-196 builds them."""
    d = demo_arrays()
    n = len(seeds)
    yaws = yaws if yaws is not None else [(0.4, 0.0, -0.4)[i % 3] for i in range(n)]
    z = torch.from_numpy(np.concatenate([np.random.RandomState(s).randn(1, 512) for s in seeds], 0))
    cs, cc = zip(*[demo_camera_params(angle_y=y, angle_p=pitch) for y in yaws])
    v = torch.cat([torch.from_numpy(d['verts']), torch.from_numpy(d['landmarks']).float()], 0).float()[None].repeat(n, 1, 1)
    return z.to(device), torch.cat(cs, 0).to(device), torch.cat(cc, 0).to(device), v.to(device)
