#!/usr/bin/env python3
"""oracle/pin_renderer_options.py — pins oracle/renderer.py's rendering options (ray_start = ray_end = 'auto', disparity_space_sampling,
white_back, density_noise: everything ImportanceRenderer / MipRayMarcher2 read from rendering_kwargs beyond the ffhq configuration)
against the REAL reference classes and writes tests/golden/render_opts.npz (TEST INFRASTRUCTURE; build container only).

Per option set: the reference's RaySampler + ImportanceRenderer + OSGDecoder (volumetric_rendering/{ray_sampler,renderer}.py,
training_avatar_texture/triplane_next3d.py:346-371) run on seeded planes / decoder weights with the random draws injected
(torch.rand_like :205, torch.rand :252, torch.randn_like :153); the oracle runs on the same inputs; inputs + REFERENCE outputs are stored.
"""
import os
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from next3d_amd import demo, mesh  # noqa: E402
from oracle import ref_shims, renderer as oren  # noqa: E402

CASES = {
    'white_back': dict(ray_start=2.25, ray_end=3.3, white_back=True),
    'disparity': dict(ray_start=2.25, ray_end=3.3, disparity_space_sampling=True),
    'auto': dict(ray_start='auto', ray_end='auto'),
    'auto_wide_fov': dict(ray_start='auto', ray_end='auto', fov=60.0),            # most rays miss the box: the repair of :101-104
    'density_noise': dict(ray_start=2.25, ray_end=3.3, density_noise=0.5),
    'all': dict(ray_start='auto', ray_end='auto', white_back=True, density_noise=0.25),
}


def main():
    ref_shims.install(mesh.synthetic_uv_face_mask()[0, 0].numpy())
    from training_avatar_texture.triplane_next3d import OSGDecoder
    from training_avatar_texture.volumetric_rendering.ray_sampler import RaySampler
    from training_avatar_texture.volumetric_rendering.renderer import ImportanceRenderer
    g = torch.Generator().manual_seed(4242)
    N, R, Sc, Sf, PH = 2, 12, 12, 12, 32
    planes = torch.randn(N, 3, 32, PH, PH, generator=g) * 2
    dec = OSGDecoder(32, {'decoder_lr_mul': 1, 'decoder_output_dim': 32}).eval().requires_grad_(False)
    P = {'decoder.net.0.weight': torch.randn(64, 32, generator=g), 'decoder.net.0.bias': torch.randn(64, generator=g) * 0.3,
         'decoder.net.2.weight': torch.randn(33, 64, generator=g), 'decoder.net.2.bias': torch.randn(33, generator=g) * 0.3}
    dec.load_state_dict({k[len('decoder.'):]: v for k, v in P.items()})
    jitter = torch.rand(N, R * R, Sc, 1, generator=g)
    u = torch.rand(N * R * R, Sf, generator=g)
    nz = (torch.randn(N, R * R * Sc, 1, generator=g), torch.randn(N, R * R * Sf, 1, generator=g))
    out, ok = {}, True
    for name, cfg in CASES.items():
        cfg = dict(cfg)
        fov = cfg.pop('fov', 18.837)
        cams = torch.cat([demo.camera_label(a, -0.2, fov_deg=fov) for a in (0.35, -0.3)], 0).float()
        c2w, K = cams[:, :16].reshape(N, 4, 4), cams[:, 16:25].reshape(N, 3, 3)
        ro, rd = RaySampler()(c2w, K, R)
        opts = dict(dict(depth_resolution=Sc, depth_resolution_importance=Sf, box_warp=1, clamp_mode='softplus', disparity_space_sampling=False), **cfg)
        draws = iter(nz)
        o_rand, o_rand_like, o_randn_like = torch.rand, torch.rand_like, torch.randn_like
        torch.rand_like = lambda t, *a, **k: jitter.clone() if tuple(t.shape) == tuple(jitter.shape) else o_rand_like(t, *a, **k)
        torch.rand = lambda *a, **k: u.clone() if (tuple(a) == tuple(u.shape) or (len(a) == 1 and tuple(a[0]) == tuple(u.shape))) else o_rand(*a, **k)
        torch.randn_like = lambda t, *a, **k: next(draws).clone()
        ir, fine_ref = ImportanceRenderer(), []
        ref_sample_importance = ir.sample_importance                     # the reference's own importance depths (:209-268): stored, so that a kernel under test can be teacher-forced with them
        ir.sample_importance = lambda *a, **k: (fine_ref.append(ref_sample_importance(*a, **k)) or fine_ref[-1])
        try:
            rgb, depth, wsum = ir(planes, dec, ro, rd, opts)
        finally:
            torch.rand, torch.rand_like, torch.randn_like = o_rand, o_rand_like, o_randn_like
        o_ro, o_rd = oren.ray_sampler(c2w, K, R)
        fine_o = []
        rgb_o, depth_o, wsum_o = oren.importance_renderer(P, 'decoder', planes, o_ro, o_rd, opts, jitter, u, noise=nz, fine_depths_out=fine_o)
        d = [float((a - b).abs().max()) for a, b in ((rgb, rgb_o), (depth, depth_o), (wsum, wsum_o), (fine_ref[0], fine_o[0]))]
        print(f'[{name}] max-abs(ref - oracle): rgb {d[0]:.2e} depth {d[1]:.2e} wsum {d[2]:.2e} importance depths {d[3]:.2e}')
        ok &= all(x <= 1e-6 for x in d)
        out.update({f'{name}_rgb': rgb.numpy(), f'{name}_depth': depth.numpy(), f'{name}_wsum': wsum.numpy(), f'{name}_cams': cams.numpy(), f'{name}_fine': fine_ref[0].numpy()})
    np.savez_compressed(os.path.join(REPO, 'tests', 'golden', 'render_opts.npz'), planes=planes.numpy(), jitter=jitter.numpy(), u=u.numpy(),
                        noise_c=nz[0].numpy(), noise_f=nz[1].numpy(), R=R, Sc=Sc, Sf=Sf, cases=np.array(list(CASES)),
                        **{k.replace('.', '__'): v.numpy() for k, v in P.items()}, **out)
    print('PIN', 'OK' if ok else 'FAILED')
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main())
