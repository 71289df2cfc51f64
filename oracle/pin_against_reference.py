#!/usr/bin/env python3
"""oracle/pin_against_reference.py — pins the oracle to the REAL reference and writes the golden
fixtures (TEST INFRASTRUCTURE; runs only in the build container, where /root/reference exists).

  python oracle/pin_against_reference.py                      # compare + (re)write tests/golden/*.npz
  python oracle/pin_against_reference.py --only case_r64_s96  # one case; the shared fixtures are left untouched
  python oracle/pin_against_reference.py --orbit              # gen_videos_next3d.py's render loop, frames 0 / 60 / 119 -> tests/golden/orbit_frames.npz

What it does, per case:
  1. builds the reference TriPlaneGenerator from its own constructors (oracle/ref_shims.py),
     loads the seeded synthetic weights (next3d_amd.spec.synthetic_state_dict) into it;
  2. monkey-patches torch.rand / torch.rand_like so the reference consumes the SAME depth jitter
     and importance `u` the oracle is given (vr/renderer.py:205,252);
  3. runs reference `mapping` + `synthesis` on CPU (fp32 forced off-GPU, networks_stylegan2.py:548),
     capturing stage outputs with forward hooks;
  4. runs the oracle on the same inputs and reports max-abs differences per stage;
  5. stores inputs + REFERENCE outputs (sub-sampled where large) in tests/golden/case_*.npz; the benched case
     (case_r64_s48_b4) additionally gets its FULL-resolution image in tests/golden/case_r64_s48_b4_image.npz;
  6. --fp16 (cases in FP16_CASES): runs the reference's OWN float16 super-resolution branch on the CPU — the
     `if ws.device.type != 'cuda': force_fp32 = True` guard of SynthesisBlock.forward (training/networks_stylegan2.py:421-422) is
     patched out at run time (the method's source is re-compiled with the guard disabled; nothing is copied into this repo) — on
     the fp32 run's own (rgb, features, ws) and stores the full-resolution result in tests/golden/<case>_fp16sr.npz;
  7. writes profiles/r03_cpu_reference.json: seconds of the reference's Python and of the oracle on this container's cores, per case
     (bench.py cites it next to its own `cpu_baseline`).
"""
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from next3d_amd import mesh as n3d_mesh                      # noqa: E402
from next3d_amd import spec as n3d_spec                      # noqa: E402
from oracle import cases                                     # noqa: E402
from oracle import generator as ogen                         # noqa: E402
from oracle import ref_shims                                 # noqa: E402

GOLDEN = os.path.join(REPO, 'tests', 'golden')

RENDERING_KWARGS = dict(
    image_resolution=512, disparity_space_sampling=False, clamp_mode='softplus',
    c_gen_conditioning_zero=True, gpc_reg_prob=None, c_scale=1.0, superresolution_noise_mode='none',
    density_reg=0.25, density_reg_p_dist=0.004, reg_type='l1', decoder_lr_mul=1.0, sr_antialias=True,
    gen_exp_cond=False, depth_resolution=48, depth_resolution_importance=48, ray_start=2.25, ray_end=3.3,
    box_warp=1, avg_camera_radius=2.7, avg_camera_pivot=[0, 0, 0.2])

CASES = {
    # BASELINE.json configs[0]: 1 seed, R=32, "48 depth samples" = 24 coarse + 24 importance
    'case_r32_s24': dict(seeds=[0], yaws=[0.4], R=32, Sc=24, Sf=24, psi=0.7),
    # BASELINE.json configs[1] shape at N=2: R=64, 48+48, two seeds / two cameras
    'case_r64_s48': dict(seeds=[1, 2], yaws=[0.0, -0.4], R=64, Sc=48, Sf=48, psi=0.7),
    # gen_videos_next3d.py's default sampling multiplier 2: 96 + 96 samples, wide yaws, another
    # truncation, BOTH meshes perturbed (4x the amplitude: other faces win the z-buffer, the mouth boxes change size)
    'case_r64_s96': dict(seeds=[3, 4], yaws=[0.6, -0.6], R=64, Sc=96, Sf=96, psi=0.5, mesh_jitter=0.002, jitter_all=True),
    # BASELINE.json configs[1] EXACTLY as bench.py runs it: batch 4, seeds 0-3, demo_batch's yaw pattern, R=64, 48+48, psi 0.7
    # (mesh 1 perturbed as in the other batched cases so that the four samples do not share one z-buffer)
    'case_r64_s48_b4': dict(seeds=[0, 1, 2, 3], yaws=[0.4, 0.0, -0.4, 0.4], R=64, Sc=48, Sf=48, psi=0.7),
    # BASELINE.json configs[3]'s per-GPU share: 8 seeds in ONE call (other kernels are selected at batch 8: split-K factors, pre-split
    # eligibility, grid shapes), demo_batch's yaw pattern; `lean`: only what the parity test reads is stored (file size)
    'case_r64_s48_b8': dict(seeds=list(range(8)), yaws=[0.4, 0.0, -0.4, 0.4, 0.0, -0.4, 0.4, 0.0], R=64, Sc=48, Sf=48, psi=0.7, lean=True),
    # round 5: the benched configuration on a SECOND, trained-like weight draw — spec.synthetic_state_dict(1, profile='wide'):
    # heavy-tailed weights, styles x 3, noise / bias 0.3 (activations of 10^2 in the blending network) — the split-bf16 error on another draw
    'case_r64_s48_b4_w1': dict(seeds=[0, 1, 2, 3], yaws=[0.4, 0.0, -0.4, 0.4], R=64, Sc=48, Sf=48, psi=0.7, lean=True, weights=(1, 'wide')),
    # round 5 (item 4c): BASELINE.json configs[2] / bench.py config3 — README.md:48's seeds as ONE 2x2-grid frame at gen_videos' default sampling
    # multiplier 2 (96 + 96 samples), four cameras of the orbit (yaw_range 0.35, gen_videos_next3d.py:131-137)
    'case_r64_s96_v4': dict(seeds=[10720, 12374, 13393, 17099], yaws=[0.35, 0.0, -0.35, 0.2], R=64, Sc=96, Sf=96, psi=0.7, lean=True),
}


FP16_CASES = ('case_r32_s24', 'case_r64_s48', 'case_r64_s48_b4')      # full-resolution image for batch 1 / 2; every second pixel for the benched batch 4 (file size)


def enable_reference_fp16_on_cpu(module_name='training.networks_stylegan2'):
    """Disable SynthesisBlock.forward's off-GPU float32 guard (training/networks_stylegan2.py:421-422; the same two lines in
    training_avatar_texture/networks_stylegan2.py:548-549 and networks_stylegan2_styleunet.py:445-446) in the imported reference
    module: the method's own source, with that one condition replaced, is compiled in the module's namespace."""
    import importlib
    import inspect
    import textwrap
    m = importlib.import_module(module_name)
    fn = m.SynthesisBlock.forward
    src = textwrap.dedent(inspect.getsource(fn))
    guard = "if ws.device.type != 'cuda':"
    assert src.count(guard) == 1, 'reference guard not found'
    ns = {}
    exec(compile(src.replace(guard, 'if False:'), '<SynthesisBlock.forward without the off-GPU fp32 guard>', 'exec'), m.__dict__, ns)
    m.SynthesisBlock.forward = ns['forward']
    return fn


def sub(t, step):
    return t[..., ::step, ::step].contiguous().numpy()


def fp16_backbones():
    """--fp16-backbones: the reference rebuilt the way legacy.load_network_pkl(force_fp16=True) rebuilds it (legacy.py:49-59:
    num_fp16_res = 4, conv_clamp = 256 in every backbone) and run END TO END on the CPU with the off-GPU float32 guards of all three
    SynthesisBlock classes disabled: the blocks of resolution >= 32 of the four backbones and both super-resolution blocks execute in
    float16.  Writes tests/golden/<case>_fp16bb.npz (inputs + the reference's outputs) and reports the oracle's emulation against it."""
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    uv_mask = n3d_mesh.synthetic_uv_face_mask()
    ref_shims.install(uv_mask[0, 0].numpy())
    import camera_utils as ref_cam
    G = ref_shims.build_reference_generator(RENDERING_KWARGS, num_fp16_res=4, conv_clamp=256)
    sd = n3d_spec.synthetic_state_dict(seed=0)
    verts, faces, uvs, uvfaces = n3d_mesh.parse_obj(os.path.join(ref_shims.REF, 'data/demo/demo.obj'))
    sd.update(n3d_mesh.mesh_buffers(faces, uvs, uvfaces))
    G.load_state_dict(sd, strict=True)
    v_demo = n3d_mesh.parse_obj_vertices(os.path.join(ref_shims.REF, 'data/demo/demo.obj'))
    lms = n3d_mesh.parse_landmarks(os.path.join(ref_shims.REF, 'data/demo/demo_kpt2d.txt'))
    stages = {}
    for name, mod in [('textures', G.texture_backbone.synthesis), ('mouths_plane', G.mouth_backbone.synthesis),
                      ('rendering_stitch', G.neural_blending.synthesis), ('static_plane', G.backbone.synthesis)]:
        mod.register_forward_hook(lambda m, i, o, name=name: stages.__setitem__(name, o))
    originals = [(mn, enable_reference_fp16_on_cpu(mn)) for mn in ('training.networks_stylegan2', 'training_avatar_texture.networks_stylegan2',
                                                                    'training_avatar_texture.networks_stylegan2_styleunet')]
    ok = True
    for cname in ('case_r32_s24', 'case_r64_s48'):
        cfg = CASES[cname]
        N, R, Sc, Sf = len(cfg['seeds']), cfg['R'], cfg['Sc'], cfg['Sf']
        G.rendering_kwargs['depth_resolution'], G.rendering_kwargs['depth_resolution_importance'] = Sc, Sf
        rk = dict(RENDERING_KWARGS, depth_resolution=Sc, depth_resolution_importance=Sf)
        z = torch.from_numpy(np.concatenate([np.random.RandomState(s).randn(1, 512) for s in cfg['seeds']], 0))
        pivot = torch.tensor(rk['avg_camera_pivot'])
        K = ref_cam.FOV_to_intrinsics(18.837)
        cams, conds = [], []
        for yaw in cfg['yaws']:
            c2w = ref_cam.LookAtPoseSampler.sample(np.pi / 2 + yaw, np.pi / 2 - 0.2, pivot, radius=2.7)
            cnd = ref_cam.LookAtPoseSampler.sample(np.pi / 2, np.pi / 2, pivot, radius=2.7)
            cams.append(torch.cat([c2w.reshape(-1, 16), K.reshape(-1, 9)], 1))
            conds.append(torch.cat([cnd.reshape(-1, 16), K.reshape(-1, 9)], 1))
        c, c_cond = torch.cat(cams, 0), torch.cat(conds, 0)
        v = torch.cat((v_demo, lms), 1).repeat(N, 1, 1)
        if N > 1:
            g = torch.Generator().manual_seed(1234)
            v[1] = v[1] + 0.0005 * torch.randn(v[1].shape, generator=g)
        jitter, u = cases.rng_inputs(N, R, Sc, Sf)
        orig_rand, orig_rand_like = torch.rand, torch.rand_like
        torch.rand_like = lambda t, *a, **k: jitter.clone() if tuple(t.shape) == tuple(jitter.shape) else orig_rand_like(t, *a, **k)
        torch.rand = lambda *a, **k: u.clone() if (tuple(a) == tuple(u.shape) or (len(a) == 1 and tuple(a[0]) == tuple(u.shape))) else orig_rand(*a, **k)
        try:
            t0 = time.time()
            ws_ref = G.mapping(z, c_cond, truncation_psi=cfg['psi'], truncation_cutoff=14)
            out_ref = G.synthesis(ws_ref, c, v, neural_rendering_resolution=R, noise_mode='const')
            t_ref = time.time() - t0
        finally:
            torch.rand, torch.rand_like = orig_rand, orig_rand_like
        net_kw = dict(fp16_resolution=32, conv_clamp=256, cpu_rounding=True)
        out_or, st = ogen.synthesis(sd, ogen.mapping(sd, z, c_cond, rk, truncation_psi=cfg['psi'], truncation_cutoff=14), c, v, uv_mask, rk, jitter, u,
                                    neural_rendering_resolution=R, return_stages=True, force_fp32=False, net_kw=net_kw)
        out32 = ogen.synthesis(sd, ws_ref, c, v, uv_mask, rk, jitter, u, neural_rendering_resolution=R)
        rep = {k: float((stages[k].float() - st[k].reshape(stages[k].shape)).abs().max()) for k in ('textures', 'static_plane', 'mouths_plane', 'rendering_stitch')}
        rep.update({k: float((out_ref[k] - out_or[k]).abs().max()) for k in ('image_raw', 'image_depth', 'image')})
        d32 = {k: float((out_ref[k] - out32[k]).abs().max()) for k in ('image_raw', 'image')}
        print(f'[{cname} fp16 backbones] reference {t_ref:.1f}s; max-abs(reference - oracle emulation): ' + ' '.join(f'{k}={x:.2e}' for k, x in rep.items()) +
              f'; reference fp16 vs float32 route: image_raw {d32["image_raw"]:.2e} image {d32["image"]:.2e}')
        assert all(t.dtype == torch.float32 for t in stages.values())
        np.savez_compressed(os.path.join(GOLDEN, f'{cname}_fp16bb.npz'),
                            z=z.numpy(), c=c.numpy(), c_cond=c_cond.numpy(), v=v.numpy(), R=R, Sc=Sc, Sf=Sf, psi=cfg['psi'], cutoff=14,
                            ws=ws_ref.numpy(), image_raw=out_ref['image_raw'].numpy(), image_depth=out_ref['image_depth'].numpy(),
                            image_sub2=sub(out_ref['image'], 2), textures_sub4=sub(stages['textures'], 4), static_plane_sub8=sub(stages['static_plane'], 8),
                            mouths_plane_sub4=sub(stages['mouths_plane'], 4), rendering_stitch_sub4=sub(stages['rendering_stitch'], 4),
                            stage_absmax=np.array([float(stages[k].abs().max()) for k in ('textures', 'static_plane', 'mouths_plane', 'rendering_stitch')]),
                            oracle_max_abs=np.array([rep[k] for k in ('textures', 'static_plane', 'mouths_plane', 'rendering_stitch', 'image_raw', 'image')]),
                            fp32_route_max_abs=np.array([d32['image_raw'], d32['image']]))
    import importlib
    for mn, fn in originals:
        importlib.import_module(mn).SynthesisBlock.forward = fn
    print('PIN fp16 backbones done')
    return 0 if ok else 1


def fp16_blocks():
    """--fp16-blocks (round 5): TEACHER-FORCED per-block goldens of the reference's float16 blocks — one per network and
    resolution.  The `force_fp16` reference (as --fp16-backbones) runs case_r32_s24 once with hooks on every float16 SynthesisBlock of the four
    backbones; then each block is evaluated ALONE by the reference on a stated input: its own captured input where that is small (the 16 x 16
    inputs of the first float16 block of the texture / static / mouth networks, stored), else a seeded input with the captured activation's
    per-channel rms (oracle.cases.block_inputs: regenerated identically by the test, not stored).  Stored: the network's latents, the block's
    outputs (x float16, img float32; sub-sampled).  One block deep, two float16 implementations agree bit for bit except where the accumulation
    order flips a rounding — a bound the float32 route cannot meet (tests/test_generator_gpu.py::test_fp16_blocks_teacher_forced)."""
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    uv_mask = n3d_mesh.synthetic_uv_face_mask()
    ref_shims.install(uv_mask[0, 0].numpy())
    import camera_utils as ref_cam
    G = ref_shims.build_reference_generator(RENDERING_KWARGS, num_fp16_res=4, conv_clamp=256)
    sd = n3d_spec.synthetic_state_dict(seed=0)
    verts, faces, uvs, uvfaces = n3d_mesh.parse_obj(os.path.join(ref_shims.REF, 'data/demo/demo.obj'))
    sd.update(n3d_mesh.mesh_buffers(faces, uvs, uvfaces))
    G.load_state_dict(sd, strict=True)
    v_demo = n3d_mesh.parse_obj_vertices(os.path.join(ref_shims.REF, 'data/demo/demo.obj'))
    lms = n3d_mesh.parse_landmarks(os.path.join(ref_shims.REF, 'data/demo/demo_kpt2d.txt'))
    for mn in ('training.networks_stylegan2', 'training_avatar_texture.networks_stylegan2', 'training_avatar_texture.networks_stylegan2_styleunet'):
        enable_reference_fp16_on_cpu(mn)
    nets = {'texture': G.texture_backbone.synthesis, 'static': G.backbone.synthesis, 'mouth': G.mouth_backbone.synthesis, 'blend': G.neural_blending.synthesis}
    cap, net_ws = {}, {}
    for nname, net in nets.items():
        def net_pre(mod, args, kwargs, nname=nname):
            ws = kwargs['ws'] if 'ws' in kwargs else [a for a in args if torch.is_tensor(a) and a.ndim == 3][0]
            net_ws[nname] = ws.detach().clone()
        net.register_forward_pre_hook(net_pre, with_kwargs=True)
        for bname, blk in net.named_children():
            if not (bname.startswith('b') and bname[1:].isdigit() and getattr(blk, 'use_fp16', False)):
                continue
            key = (nname, int(bname[1:]))

            def pre(mod, args, kwargs, key=key):
                cap[key] = dict(x=args[0].detach().clone(), img=None if args[1] is None else args[1].detach().clone(), ws=args[2].detach().clone(),
                                kw={k: v for k, v in kwargs.items()})

            def post(mod, args, kwargs, out, key=key):
                cap[key]['x_out'], cap[key]['img_out'] = out[0].detach().clone(), out[1].detach().clone()
            blk.register_forward_pre_hook(pre, with_kwargs=True)
            blk.register_forward_hook(post, with_kwargs=True)
    cfg = CASES['case_r32_s24']
    N, R, Sc, Sf = 1, cfg['R'], cfg['Sc'], cfg['Sf']
    G.rendering_kwargs['depth_resolution'], G.rendering_kwargs['depth_resolution_importance'] = Sc, Sf
    z = torch.from_numpy(np.concatenate([np.random.RandomState(s).randn(1, 512) for s in cfg['seeds']], 0))
    pivot = torch.tensor(RENDERING_KWARGS['avg_camera_pivot'])
    K = ref_cam.FOV_to_intrinsics(18.837)
    c2w = ref_cam.LookAtPoseSampler.sample(np.pi / 2 + cfg['yaws'][0], np.pi / 2 - 0.2, pivot, radius=2.7)
    cnd = ref_cam.LookAtPoseSampler.sample(np.pi / 2, np.pi / 2, pivot, radius=2.7)
    c, c_cond = torch.cat([c2w.reshape(-1, 16), K.reshape(-1, 9)], 1), torch.cat([cnd.reshape(-1, 16), K.reshape(-1, 9)], 1)
    v = torch.cat((v_demo, lms), 1)
    ws_ref = G.mapping(z, c_cond, truncation_psi=cfg['psi'], truncation_cutoff=14)
    G.synthesis(ws_ref, c, v, neural_rendering_resolution=R, noise_mode='const')
    full = dict(cap)                                                       # (the hooks keep firing on the stand-alone calls below)
    out = {}
    from oracle import networks as onet
    for (nname, res), e in sorted(full.items()):
        blk = getattr(nets[nname], f'b{res}')
        real_in = e['x'].shape[-1] <= 16
        tag = f'fp16blk:{nname}:b{res}'
        x_rms = e['x'].float().pow(2).mean(dim=(0, 2, 3)).sqrt()
        has_img = e['img'] is not None                                   # (a StyleUNet's first decoder block starts the skip image: img is None)
        img_rms = e['img'].float().pow(2).mean(dim=(0, 2, 3)).sqrt() if has_img else torch.zeros(1)
        if real_in:
            x_in, img_in = e['x'].to(torch.float16), (e['img'].float() if has_img else None)
            out[f'{nname}_b{res}_x_in'] = x_in.numpy()
            if has_img:
                out[f'{nname}_b{res}_img_in'] = img_in.numpy()
        else:
            x_in, img_in = cases.block_inputs(tag, tuple(e['x'].shape), tuple(e['img'].shape) if has_img else (1, 1, 1, 1), x_rms, img_rms)
            img_in = img_in if has_img else None
            out[f'{nname}_b{res}_x_rms'] = x_rms.numpy()
            if has_img:
                out[f'{nname}_b{res}_img_rms'] = img_rms.numpy()
        t0 = time.time()
        clone = lambda t: None if t is None else t.clone()
        x_out, img_out = blk(x_in.clone(), clone(img_in), e['ws'], **e['kw'])
        assert x_out.dtype == torch.float16 and img_out.dtype == torch.float32
        step = max(2, res // 16)
        out[f'{nname}_b{res}_x_out'], out[f'{nname}_b{res}_img_out'] = sub(x_out, step), sub(img_out, step)
        out[f'{nname}_b{res}_step'] = step
        # what the float32 route gives on the same input (the bound the test states must separate the two)
        x32, img32 = blk(x_in.clone(), clone(img_in), e['ws'], force_fp32=True, **e['kw'])
        ulp = lambda t: torch.exp2(torch.floor(torch.log2(t.abs().float().clamp_min(6.1e-5))) - 10)
        d32 = (x32.to(torch.float16).float() - x_out.float()).abs() / ulp(x_out)
        same32 = float((x32.to(torch.float16) == x_out).float().mean())
        out[f'{nname}_b{res}_fp32_route'] = np.array([same32, float(d32.mean()), float((img32 - img_out).abs().max())])
        print(f'[{nname} b{res}] input {"captured" if real_in else "seeded"} {tuple(x_in.shape)} |x| max {float(x_in.abs().max()):.1f}; reference block {time.time() - t0:.1f}s; '
              f'out |x| max {float(x_out.abs().max()):.1f}, img max {float(img_out.abs().max()):.1f}; float32 route: {100 * same32:.1f}% of x bit-equal, '
              f'mean {float(d32.mean()):.2f} ulp, img max-abs {float((img32 - img_out).abs().max()):.2e}')
    for nname, ws in net_ws.items():
        out[f'{nname}_ws'] = ws.numpy()
    np.savez_compressed(os.path.join(GOLDEN, 'fp16_blocks.npz'), **out)
    print('PIN fp16 blocks done:', os.path.getsize(os.path.join(GOLDEN, 'fp16_blocks.npz')) // 1024, 'KB')
    return 0


# (class, channel_base, channel_max, golden suffix): the reference's OTHER architectures, one small case each
VARIANTS_SR = [('SuperresolutionHybrid8X', 32768, 512, 'sr8X'), ('SuperresolutionHybrid4X', 32768, 512, 'sr4X'), ('SuperresolutionHybrid2X', 32768, 512, 'sr2X')]
# (SuperresolutionHybridDeepfp32, :127-154, cannot be constructed by the reference's own TriPlaneGenerator: triplane_next3d.py:66 passes sr_antialias=..., which that class does
#  not take and forwards to SynthesisLayer -> TypeError.  Tried here in round 5; the generator of this package raises for it too.)
# ... and the two modules with a SynthesisBlockNoUp at a render that needs NO resize (2X at 64 x 64, 4X at 128 x 128): the reference's in-place `img.add_(y)` then lands in
# the returned 'image_raw' (ADVICE r5): tests/golden/case_r32_s24_sr2X_r64.npz / _sr4X_r128.npz (the case's inputs at another neural rendering resolution)
VARIANTS_SR_NORESIZE = [('SuperresolutionHybrid2X', 32768, 512, 'sr2X_r64', 64), ('SuperresolutionHybrid4X', 32768, 512, 'sr4X_r128', 128)]
VARIANTS_WIDTH = [('SuperresolutionHybrid8XDC', 16384, 512, 'cb16384'), ('SuperresolutionHybrid8XDC', 16384, 256, 'cb16384_cm256')]


def sr_modules(variants=VARIANTS_SR, what='sr modules'):
    """--sr-modules (round 5, VERDICT r4 missing #4): the reference's OTHER super-resolution modules — SuperresolutionHybrid8X (512 x 512, other channel
    counts), 4X (256 x 256, a SynthesisBlockNoUp first, resizes only a smaller render) and 2X (128 x 128) (tat/superresolution.py:29-124) — built by the
    reference's own constructors (state-dict names diffed against next3d_amd.spec), run on case_r32_s24's inputs, compared with the oracle (max-abs 0.0
    expected) and committed as tests/golden/case_r32_s24_sr{8X,4X,2X}.npz.
    --channel-widths: the same for other backbone widths — `--cbase 16384` (and `--cmax 256`) of train_next3d.py:199-200, which every backbone receives as
    synthesis_kwargs (tat/networks_stylegan2.py:614 channels_dict): tests/golden/case_r32_s24_cb16384{,_cm256}.npz."""
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    uv_mask = n3d_mesh.synthetic_uv_face_mask()
    ref_shims.install(uv_mask[0, 0].numpy())
    import camera_utils as ref_cam
    verts, faces, uvs, uvfaces = n3d_mesh.parse_obj(os.path.join(ref_shims.REF, 'data/demo/demo.obj'))
    mb = n3d_mesh.mesh_buffers(faces, uvs, uvfaces)
    v_demo = n3d_mesh.parse_obj_vertices(os.path.join(ref_shims.REF, 'data/demo/demo.obj'))
    lms = n3d_mesh.parse_landmarks(os.path.join(ref_shims.REF, 'data/demo/demo_kpt2d.txt'))
    cfg = CASES['case_r32_s24']
    ok = True
    for cls, cb, cm, suffix, *r_over in variants:
        R, Sc, Sf = (r_over[0] if r_over else cfg['R']), cfg['Sc'], cfg['Sf']
        rk = dict(RENDERING_KWARGS, depth_resolution=Sc, depth_resolution_importance=Sf, superresolution_module=f'training_avatar_texture.superresolution.{cls}')
        G = ref_shims.build_reference_generator(rk, channel_base=cb, channel_max=cm)
        ref_sd = G.state_dict()
        sd = n3d_spec.synthetic_state_dict(seed=0, sr=cls, channel_base=cb, channel_max=cm)
        sd.update(mb)
        assert set(ref_sd) == set(sd) and all(tuple(ref_sd[k].shape) == tuple(sd[k].shape) for k in sd), (cls, set(ref_sd) ^ set(sd))
        G.load_state_dict(sd, strict=True)
        z = torch.from_numpy(np.concatenate([np.random.RandomState(s_).randn(1, 512) for s_ in cfg['seeds']], 0))
        pivot = torch.tensor(rk['avg_camera_pivot'])
        K = ref_cam.FOV_to_intrinsics(18.837)
        c2w = ref_cam.LookAtPoseSampler.sample(np.pi / 2 + cfg['yaws'][0], np.pi / 2 - 0.2, pivot, radius=2.7)
        cnd = ref_cam.LookAtPoseSampler.sample(np.pi / 2, np.pi / 2, pivot, radius=2.7)
        c, c_cond = torch.cat([c2w.reshape(-1, 16), K.reshape(-1, 9)], 1), torch.cat([cnd.reshape(-1, 16), K.reshape(-1, 9)], 1)
        v = torch.cat((v_demo, lms), 1)
        jitter, u = cases.rng_inputs(1, R, Sc, Sf)
        orig_rand, orig_rand_like = torch.rand, torch.rand_like
        torch.rand_like = lambda t, *a, **k: jitter.clone() if tuple(t.shape) == tuple(jitter.shape) else orig_rand_like(t, *a, **k)
        torch.rand = lambda *a, **k: u.clone() if (tuple(a) == tuple(u.shape) or (len(a) == 1 and tuple(a[0]) == tuple(u.shape))) else orig_rand(*a, **k)
        try:
            ws_ref = G.mapping(z, c_cond, truncation_psi=cfg['psi'], truncation_cutoff=14)
            out_ref = G.synthesis(ws_ref, c, v, neural_rendering_resolution=R, noise_mode='const')
        finally:
            torch.rand, torch.rand_like = orig_rand, orig_rand_like
        out_or = ogen.synthesis(sd, ogen.mapping(sd, z, c_cond, rk, truncation_psi=cfg['psi'], truncation_cutoff=14), c, v, uv_mask, rk, jitter, u, neural_rendering_resolution=R)
        rep = {k: float((out_ref[k] - out_or[k]).abs().max()) for k in ('image_raw', 'image_depth', 'image')}
        print(f'[{cls} channel_base {cb} channel_max {cm}] image {tuple(out_ref["image"].shape)}; max-abs(reference - oracle): ' + ' '.join(f'{k}={x:.2e}' for k, x in rep.items()))
        ok &= all(x <= 1e-4 for x in rep.values())
        step = out_ref['image'].shape[-1] // 128
        np.savez_compressed(os.path.join(GOLDEN, f'case_r32_s24_{suffix}.npz'), sr_class=cls, channel_base=cb, channel_max=cm,
                            z=z.numpy(), c=c.numpy(), c_cond=c_cond.numpy(), v=v.numpy(), R=R, Sc=Sc, Sf=Sf, psi=cfg['psi'], cutoff=14, ws=ws_ref.numpy(),
                            image_raw=out_ref['image_raw'].numpy(), image_depth=out_ref['image_depth'].numpy(), image_sub=sub(out_ref['image'], step), image_step=step,
                            image_mean=out_ref['image'].mean(dim=(2, 3)).numpy(), state_dict_names=np.array(sorted(k for k in ref_sd if k.startswith('superresolution'))))
    print(f'PIN {what}', 'OK' if ok else 'FAILED')
    return 0 if ok else 1


def mapping_depth():
    """--mapping-depth: `mapping_kwargs.num_layers` other than train_next3d.py's map_depth = 2 — the MappingNetwork's own default of 8 (what the reference builds when the
    key is absent, tat/networks_stylegan2.py:207) and 1: the reference's constructors with that depth, seeded synthetic weights, ws of four seeds compared with the
    oracle (max-abs 0.0 expected) and committed as tests/golden/mapping_depth.npz."""
    torch.manual_seed(0)
    uv_mask = n3d_mesh.synthetic_uv_face_mask()
    ref_shims.install(uv_mask[0, 0].numpy())
    import camera_utils as ref_cam
    verts, faces, uvs, uvfaces = n3d_mesh.parse_obj(os.path.join(ref_shims.REF, 'data/demo/demo.obj'))
    mb = n3d_mesh.mesh_buffers(faces, uvs, uvfaces)
    z = torch.from_numpy(np.concatenate([np.random.RandomState(s_).randn(1, 512) for s_ in (0, 1, 2, 3)], 0))
    pivot = torch.tensor(RENDERING_KWARGS['avg_camera_pivot'])
    K = ref_cam.FOV_to_intrinsics(18.837)
    cnd = ref_cam.LookAtPoseSampler.sample(np.pi / 2, np.pi / 2, pivot, radius=2.7)
    c_cond = torch.cat([cnd.reshape(-1, 16), K.reshape(-1, 9)], 1).repeat(4, 1)
    out, ok = dict(z=z.numpy(), c_cond=c_cond.numpy(), psi=0.7, cutoff=14), True
    for depth in (8, 1):
        G = ref_shims.build_reference_generator(RENDERING_KWARGS, mapping_kwargs=dict(num_layers=depth) if depth != 8 else {})
        sd = n3d_spec.synthetic_state_dict(seed=0, mapping_layers=depth)
        sd.update(mb)
        ref_sd = G.state_dict()
        assert set(ref_sd) == set(sd) and all(tuple(ref_sd[k].shape) == tuple(sd[k].shape) for k in sd), (depth, set(ref_sd) ^ set(sd))
        G.load_state_dict(sd, strict=True)
        ws_ref = G.mapping(z, c_cond, truncation_psi=0.7, truncation_cutoff=14)
        ws_or = ogen.mapping(sd, z, c_cond, RENDERING_KWARGS, truncation_psi=0.7, truncation_cutoff=14)
        err = float((ws_ref - ws_or).abs().max())
        print(f'[mapping depth {depth}] {sum(1 for k in ref_sd if "backbone.mapping.fc" in k and k.endswith("weight") and k.startswith("backbone"))} FC layers; max-abs(reference - oracle) ws = {err:.2e}')
        ok &= err <= 1e-6
        out[f'ws_depth{depth}'] = ws_ref.numpy()
    np.savez_compressed(os.path.join(GOLDEN, 'mapping_depth.npz'), **out)
    print('PIN mapping depth', 'OK' if ok else 'FAILED')
    return 0 if ok else 1


ORBIT_FRAMES = (0, 60, 119)


def orbit():
    """--orbit: gen_videos_next3d.py's render loop (:96-158) with the argument values the script itself computes — grid 1 x 1, one seed = one keyframe,
    w_frames = 120 (120-frame camera orbit: BASELINE.json configs[2]), --trunc 0.7, --sample_mult 2 (96 + 96 samples, :288-289), the scipy cubic
    interpolation of the tiled keyframe latents (float64, :112-120, :140-141), the per-frame LookAtPoseSampler pose (:131-137) and
    `G.synthesis(ws=w.unsqueeze(0), c=c[0:1], v=verts[0:1], noise_mode='const')` (:153) — run on the REAL reference for frames 0 / 60 / 119, the renderer's
    random draws injected as everywhere else.  Stores the frames' inputs and the reference's outputs in tests/golden/orbit_frames.npz
    (tests/test_generator_gpu.py::test_orbit_frames_match_reference_golden replays the same expressions through next3d_amd)."""
    import scipy.interpolate
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    uv_mask = n3d_mesh.synthetic_uv_face_mask()
    ref_shims.install(uv_mask[0, 0].numpy())
    import camera_utils as ref_cam
    from camera_utils import LookAtPoseSampler
    G = ref_shims.build_reference_generator(RENDERING_KWARGS)
    sd = n3d_spec.synthetic_state_dict(seed=0)
    verts_, faces, uvs, uvfaces = n3d_mesh.parse_obj(os.path.join(ref_shims.REF, 'data/demo/demo.obj'))
    sd.update(n3d_mesh.mesh_buffers(faces, uvs, uvfaces))
    G.load_state_dict(sd, strict=True)
    v_demo = n3d_mesh.parse_obj_vertices(os.path.join(ref_shims.REF, 'data/demo/demo.obj'))
    lms = n3d_mesh.parse_landmarks(os.path.join(ref_shims.REF, 'data/demo/demo_kpt2d.txt'))
    device = torch.device('cpu')
    # ---- the script's own lines from here on (gen_videos_next3d.py), device = cpu
    seeds, grid_w, grid_h, w_frames, wraps, kind, psi, truncation_cutoff, sampling_multiplier = [0], 1, 1, 120, 2, 'cubic', 0.7, 14, 2
    G.rendering_kwargs['depth_resolution'] = int(G.rendering_kwargs['depth_resolution'] * sampling_multiplier)                        # :288
    G.rendering_kwargs['depth_resolution_importance'] = int(G.rendering_kwargs['depth_resolution_importance'] * sampling_multiplier)  # :289
    verts = torch.cat((v_demo, lms), 1)                                              # :291-316 (v from the .obj, landmarks appended)
    num_keyframes = len(seeds) // (grid_w * grid_h)
    all_seeds = np.zeros(num_keyframes * grid_h * grid_w, dtype=np.int64)
    for idx in range(num_keyframes * grid_h * grid_w):
        all_seeds[idx] = seeds[idx % len(seeds)]
    camera_lookat_point = torch.tensor(G.rendering_kwargs['avg_camera_pivot'], device=device)
    zs = torch.from_numpy(np.stack([np.random.RandomState(seed).randn(G.z_dim) for seed in all_seeds])).to(device)
    cam2world_pose = LookAtPoseSampler.sample(3.14 / 2, 3.14 / 2, camera_lookat_point, radius=G.rendering_kwargs['avg_camera_radius'], device=device)
    focal_length = 4.2647
    intrinsics = torch.tensor([[focal_length, 0, 0.5], [0, focal_length, 0.5], [0, 0, 1]], device=device)
    c = torch.cat([cam2world_pose.reshape(-1, 16), intrinsics.reshape(-1, 9)], 1)
    c = c.repeat(len(zs), 1)
    verts = verts.repeat(len(zs), 1, 1)
    ws = G.mapping(z=zs, c=c, truncation_psi=psi, truncation_cutoff=truncation_cutoff)
    ws_map, c_cond = ws.clone(), c.clone()
    ws = ws.reshape(grid_h, grid_w, num_keyframes, *ws.shape[1:])
    x = np.arange(-num_keyframes * wraps, num_keyframes * (wraps + 1))
    y = np.tile(ws[0][0].cpu().numpy(), [wraps * 2 + 1, 1, 1])
    interp = scipy.interpolate.interp1d(x, y, kind=kind, axis=0)
    R, Sc, Sf = G.neural_rendering_resolution, G.rendering_kwargs['depth_resolution'], G.rendering_kwargs['depth_resolution_importance']
    rk = dict(RENDERING_KWARGS, depth_resolution=Sc, depth_resolution_importance=Sf)
    jitter, u = cases.rng_inputs(1, R, Sc, Sf)
    out, ok = {}, True
    for frame_idx in ORBIT_FRAMES:
        pitch_range = 0.25
        yaw_range = 0.35
        cam2world_pose = LookAtPoseSampler.sample(3.14 / 2 + yaw_range * np.sin(2 * 3.14 * frame_idx / (num_keyframes * w_frames // 2)),
                                                  3.14 / 2 - 0.05 + pitch_range * np.cos(2 * 3.14 * frame_idx / (num_keyframes * w_frames // 2)),
                                                  camera_lookat_point, radius=G.rendering_kwargs['avg_camera_radius'], device=device)
        c = torch.cat([cam2world_pose.reshape(-1, 16), intrinsics.reshape(-1, 9)], 1)
        w = torch.from_numpy(interp(frame_idx / w_frames)).to(device)
        orig_rand, orig_rand_like = torch.rand, torch.rand_like
        torch.rand_like = lambda t, *a, **k: jitter.clone() if tuple(t.shape) == tuple(jitter.shape) else orig_rand_like(t, *a, **k)
        torch.rand = lambda *a, **k: u.clone() if (tuple(a) == tuple(u.shape) or (len(a) == 1 and tuple(a[0]) == tuple(u.shape))) else orig_rand(*a, **k)
        try:
            t0 = time.time()
            o = G.synthesis(ws=w.unsqueeze(0), c=c[0:1], v=verts[0:1], noise_mode='const')
            t_ref = time.time() - t0
        finally:
            torch.rand, torch.rand_like = orig_rand, orig_rand_like
        o_or = ogen.synthesis(sd, w.unsqueeze(0).to(torch.float32), c[0:1], verts[0:1], uv_mask, rk, jitter, u, neural_rendering_resolution=R)
        d = {k: float((o[k] - o_or[k]).abs().max()) for k in ('image', 'image_raw', 'image_depth')}
        print(f'[orbit frame {frame_idx}] w dtype {w.dtype}, reference {t_ref:.1f}s, max-abs(ref - oracle): ' + ' '.join(f'{k}={x_:.2e}' for k, x_ in d.items()))
        ok &= all(x_ <= 1e-4 for x_ in d.values())
        out.update({f'c_{frame_idx}': c.numpy(), f'image_{frame_idx}_sub2': sub(o['image'], 2), f'image_raw_{frame_idx}': o['image_raw'].numpy(),
                    f'image_depth_{frame_idx}': o['image_depth'].numpy(), f'image_mean_{frame_idx}': o['image'].mean(dim=(2, 3)).numpy()})
    np.savez_compressed(os.path.join(GOLDEN, 'orbit_frames.npz'), frames=np.array(ORBIT_FRAMES), seed=seeds[0], w_frames=w_frames, wraps=wraps, psi=psi,
                        cutoff=truncation_cutoff, z=zs.numpy(), c_cond=c_cond.numpy(), ws=ws_map.numpy(), v=verts.numpy(), R=R, Sc=Sc, Sf=Sf, **out)
    print('PIN', 'OK' if ok else 'FAILED')
    return 0 if ok else 1


def main():
    if '--orbit' in sys.argv:
        return orbit()
    if '--mapping-depth' in sys.argv:
        return mapping_depth()
    if '--sr-modules' in sys.argv:
        return sr_modules()
    if '--channel-widths' in sys.argv:
        return sr_modules(VARIANTS_WIDTH, 'channel widths')
    if '--sr-noresize' in sys.argv:
        return sr_modules(VARIANTS_SR_NORESIZE, 'sr modules without a resize (in-place image_raw)')
    if '--fp16-backbones' in sys.argv:
        return fp16_backbones()
    if '--fp16-blocks' in sys.argv:
        return fp16_blocks()
    only = sys.argv[sys.argv.index('--only') + 1] if '--only' in sys.argv else None
    do_fp16 = '--fp16' in sys.argv
    timings = {}
    torch.manual_seed(0)
    torch.set_num_threads(os.cpu_count())
    uv_mask = n3d_mesh.synthetic_uv_face_mask()
    ref_shims.install(uv_mask[0, 0].numpy())
    import camera_utils as ref_cam                                                   # reference module
    G = ref_shims.build_reference_generator(RENDERING_KWARGS)
    ref_sd = G.state_dict()

    # --- state-dict inventory (spec fixture)
    if only is None:
        with open(os.path.join(GOLDEN, 'ref_state_dict_spec.txt'), 'w') as fh:
            for k in sorted(ref_sd):
                fh.write(f'{k} {tuple(ref_sd[k].shape)} {ref_sd[k].dtype}\n')

    # --- synthetic weights -> reference
    sd = n3d_spec.synthetic_state_dict(seed=0)
    verts, faces, uvs, uvfaces = n3d_mesh.parse_obj(os.path.join(ref_shims.REF, 'data/demo/demo.obj'))
    mb = n3d_mesh.mesh_buffers(faces, uvs, uvfaces)
    for k, v in mb.items():
        assert torch.equal(v, ref_sd[k]), f'mesh buffer {k} differs from the reference constructor'
    sd.update(mb)
    missing = set(ref_sd) - set(sd)
    extra = set(sd) - set(ref_sd)
    assert not missing and not extra, (missing, extra)
    G.load_state_dict(sd, strict=True)
    assert torch.equal(G.uv_face_mask, uv_mask), 'uv_face_mask path differs'

    v_demo = n3d_mesh.parse_obj_vertices(os.path.join(ref_shims.REF, 'data/demo/demo.obj'))
    lms = n3d_mesh.parse_landmarks(os.path.join(ref_shims.REF, 'data/demo/demo_kpt2d.txt'))
    if only is None:
        np.savez_compressed(os.path.join(GOLDEN, 'demo_inputs.npz'), verts=v_demo[0].numpy(), landmarks=lms[0].numpy(),
                            faces=faces.numpy().astype(np.int32), uvs=uvs.numpy(), uvfaces=uvfaces.numpy().astype(np.int32))

    stages = {}
    hooks = []

    def cap(name):
        def fn(mod, inp, out):
            stages[name] = out if isinstance(out, torch.Tensor) else out
        return fn
    for name, mod in [('textures', G.texture_backbone.synthesis), ('mouths_plane', G.mouth_backbone.synthesis),
                      ('rendering_stitch', G.neural_blending.synthesis), ('static_plane', G.backbone.synthesis)]:
        hooks.append(mod.register_forward_hook(cap(name)))
    hooks.append(G.renderer.register_forward_hook(lambda m, i, o: stages.__setitem__('renderer', (i[0], o))))
    hooks.append(G.superresolution.register_forward_hook(lambda m, i, o: stages.__setitem__('sr_in', tuple(t.clone() for t in i[:3]))))

    overall_ok = True
    loaded = (0, 'unit')
    for cname, cfg in CASES.items():
        if only is not None and cname != only:
            continue
        if cfg.get('weights', (0, 'unit')) != loaded:                      # another weight draw: both sides reload it
            loaded = cfg.get('weights', (0, 'unit'))
            sd = n3d_spec.synthetic_state_dict(seed=loaded[0], profile=loaded[1])
            sd.update(mb)
            G.load_state_dict(sd, strict=True)
        N = len(cfg['seeds'])
        R, Sc, Sf = cfg['R'], cfg['Sc'], cfg['Sf']
        G.rendering_kwargs['depth_resolution'] = Sc
        G.rendering_kwargs['depth_resolution_importance'] = Sf
        rk = dict(RENDERING_KWARGS, depth_resolution=Sc, depth_resolution_importance=Sf)

        z = torch.from_numpy(np.concatenate([np.random.RandomState(s).randn(1, 512) for s in cfg['seeds']], 0))
        pivot = torch.tensor(rk['avg_camera_pivot'])
        K = ref_cam.FOV_to_intrinsics(18.837)
        cams, conds = [], []
        for yaw in cfg['yaws']:
            c2w = ref_cam.LookAtPoseSampler.sample(np.pi / 2 + yaw, np.pi / 2 - 0.2, pivot, radius=2.7)
            cnd = ref_cam.LookAtPoseSampler.sample(np.pi / 2, np.pi / 2, pivot, radius=2.7)
            cams.append(torch.cat([c2w.reshape(-1, 16), K.reshape(-1, 9)], 1))
            conds.append(torch.cat([cnd.reshape(-1, 16), K.reshape(-1, 9)], 1))
        c, c_cond = torch.cat(cams, 0), torch.cat(conds, 0)
        v = torch.cat((v_demo, lms), 1).repeat(N, 1, 1)
        if N > 1:   # perturb the second mesh a little so the batch is not degenerate
            g = torch.Generator().manual_seed(1234)
            v[1] = v[1] + cfg.get('mesh_jitter', 0.0005) * torch.randn(v[1].shape, generator=g)
            if cfg.get('jitter_all', False):
                v[0] = v[0] + cfg.get('mesh_jitter', 0.0005) * torch.randn(v[0].shape, generator=g)

        jitter, u = cases.rng_inputs(N, R, Sc, Sf)

        # reference run with injected randomness
        orig_rand, orig_rand_like = torch.rand, torch.rand_like
        torch.rand_like = lambda t, *a, **k: jitter.clone() if tuple(t.shape) == tuple(jitter.shape) else orig_rand_like(t, *a, **k)
        torch.rand = lambda *a, **k: u.clone() if (tuple(a) == tuple(u.shape) or (len(a) == 1 and tuple(a[0]) == tuple(u.shape))) else orig_rand(*a, **k)
        try:
            t0 = time.time()
            ws_ref = G.mapping(z, c_cond, truncation_psi=cfg['psi'], truncation_cutoff=14)
            out_ref = G.synthesis(ws_ref, c, v, neural_rendering_resolution=R, noise_mode='const')
            t_ref = time.time() - t0
        finally:
            torch.rand, torch.rand_like = orig_rand, orig_rand_like

        # oracle run
        t0 = time.time()
        ws_or = ogen.mapping(sd, z, c_cond, rk, truncation_psi=cfg['psi'], truncation_cutoff=14)
        out_or, st = ogen.synthesis(sd, ws_or, c, v, uv_mask, rk, jitter, u, neural_rendering_resolution=R,
                                    return_stages=True)
        t_or = time.time() - t0

        def md(a, b):
            return float((a - b).abs().max())
        rep = {
            'ws': md(ws_ref, ws_or),
            'textures': md(stages['textures'], st['textures']),
            'mouths_plane': md(stages['mouths_plane'], st['mouths_plane']),
            'rendering_stitch': md(stages['rendering_stitch'], st['rendering_stitch']),
            'static_plane': md(stages['static_plane'], st['static_plane'].reshape(stages['static_plane'].shape)),
            'blended_planes': md(stages['renderer'][0], st['blended_planes']),
            'image_raw': md(out_ref['image_raw'], out_or['image_raw']),
            'image_depth': md(out_ref['image_depth'], out_or['image_depth']),
            'image': md(out_ref['image'], out_or['image']),
        }
        # point queries (G.sample_mixed, triplane_next3d.py:278): the shape-extraction entry point, same planes + decoder
        gq = torch.Generator().manual_seed(777)
        coords = torch.rand(N, 4096, 3, generator=gq) * 1.1 - 0.55          # a little outside the box on purpose (zero padding)
        smp_ref = G.sample_mixed(coords, torch.zeros_like(coords), ws_ref, v, noise_mode='const')
        smp_or = ogen.run_model(sd, st['blended_planes'], coords, rk)
        rep['sample_rgb'] = md(smp_ref['rgb'], smp_or['rgb'])
        rep['sample_sigma'] = md(smp_ref['sigma'], smp_or['sigma'])
        timings[cname] = {'batch': N, 'R': R, 'samples': [Sc, Sf], 'reference_seconds': round(t_ref, 2), 'oracle_seconds': round(t_or, 2),
                          'reference_frames_per_s': round(N / t_ref, 4), 'oracle_frames_per_s': round(N / t_or, 4)}
        print(f'[{cname}] reference {t_ref:.1f}s oracle {t_or:.1f}s  max-abs(ref-oracle): ' +
              ' '.join(f'{k}={v:.2e}' for k, v in rep.items()))
        ok = all(v <= 1e-4 for v in rep.values())
        overall_ok &= ok
        planes = stages['renderer'][0]
        if cfg.get('lean', False):
            np.savez_compressed(
                os.path.join(GOLDEN, f'{cname}.npz'),
                z=z.numpy(), c=c.numpy(), c_cond=c_cond.numpy(), v=v.numpy(), R=R, Sc=Sc, Sf=Sf, psi=cfg['psi'], cutoff=14,
                ws=ws_ref.numpy(), image_raw=out_ref['image_raw'].numpy(), image_depth=out_ref['image_depth'].numpy(),
                image_sub4=sub(out_ref['image'], 4), image_mean=out_ref['image'].mean(dim=(2, 3)).numpy(),
                textures_sub8=sub(stages['textures'], 8), static_plane_sub8=sub(stages['static_plane'], 8),
                alpha=(st['alpha'].numpy() * 255).round().astype(np.uint8), mouth_mask=st['mouth_mask'].numpy(),
                weights_seed=loaded[0], weights_profile=loaded[1],
                stage_absmax=np.array([float(stages[k].abs().max()) for k in ('textures', 'mouths_plane', 'rendering_stitch', 'static_plane')] +
                                      [float(out_ref['image'].abs().max())]))
            continue
        np.savez_compressed(
            os.path.join(GOLDEN, f'{cname}.npz'),
            # inputs
            z=z.numpy(), c=c.numpy(), c_cond=c_cond.numpy(), v=v.numpy(), R=R, Sc=Sc, Sf=Sf, psi=cfg['psi'], cutoff=14,
            # (jitter, u) are regenerated by oracle.cases.rng_inputs(N, R, Sc, Sf)
            # reference outputs
            ws=ws_ref.numpy(), image_raw=out_ref['image_raw'].numpy(), image_depth=out_ref['image_depth'].numpy(),
            image_sub4=sub(out_ref['image'], 4), image_mean=out_ref['image'].mean(dim=(2, 3)).numpy(),
            image_absmean=out_ref['image'].abs().mean(dim=(2, 3)).numpy(),
            textures_sub8=sub(stages['textures'], 8), mouths_plane_sub8=sub(stages['mouths_plane'], 8),
            rendering_stitch_sub8=sub(stages['rendering_stitch'], 8), static_plane_sub8=sub(stages['static_plane'], 8),
            blended_planes_sub8=sub(planes, 8), alpha=(st['alpha'].numpy() * 255).round().astype(np.uint8),
            mouth_mask=st['mouth_mask'].numpy(),
            sample_coords=coords.numpy(), sample_rgb=smp_ref['rgb'].numpy(), sample_sigma=smp_ref['sigma'].numpy(),
            stage_absmean=np.array([float(stages[k].abs().mean()) for k in
                                    ('textures', 'mouths_plane', 'rendering_stitch', 'static_plane')]),
        )
        if cname == 'case_r64_s48_b4':       # the benched configuration: every pixel of the reference image
            np.savez_compressed(os.path.join(GOLDEN, f'{cname}_image.npz'), image=out_ref['image'].numpy())
        if do_fp16 and cname in FP16_CASES:
            # the reference's DEFAULT super-resolution route (sr_num_fp16_res = 4, no force_fp32): its own float16 blocks, on the CPU
            rgb_in, feat_in, ws_in = stages['sr_in']
            orig_forward = enable_reference_fp16_on_cpu()
            import training.networks_stylegan2 as ref_sg2
            try:
                t0 = time.time()
                img16 = G.superresolution(rgb_in, feat_in, ws_in, noise_mode=RENDERING_KWARGS['superresolution_noise_mode'])
                t16 = time.time() - t0
            finally:
                ref_sg2.SynthesisBlock.forward = orig_forward
            assert img16.dtype == torch.float32
            from oracle import networks as onet
            or16 = onet.superresolution(sd, 'superresolution', rgb_in, feat_in, ws_in, force_fp32=False)
            d16 = (img16 - or16).abs()
            d32 = (img16 - out_ref['image']).abs()
            print(f'[{cname}] reference float16 SR on CPU {t16:.1f}s: vs oracle fp16 emulation max-abs {float(d16.max()):.3e} mean {float(d16.mean()):.3e}; '
                  f'vs the float32 route max-abs {float(d32.max()):.3e} mean {float(d32.mean()):.3e}')
            timings[cname]['reference_fp16_sr_seconds'] = round(t16, 2)
            timings[cname]['oracle_fp16_vs_reference_fp16_max_abs'] = float(d16.max())
            timings[cname]['reference_fp16_vs_fp32_max_abs'] = float(d32.max())
            step = 2 if N > 2 else 1
            np.savez_compressed(os.path.join(GOLDEN, f'{cname}_fp16sr.npz'), image=img16[..., ::step, ::step].contiguous().numpy(), image_step=step,
                                rgb_in=rgb_in.numpy(), feat_in=feat_in.numpy(), ws_in=ws_in.numpy(), oracle_max_abs=float(d16.max()))
    for h in hooks:
        h.remove()
    if timings:
        path = os.path.join(REPO, 'profiles', 'r03_cpu_reference.json')
        prev = json.load(open(path)) if os.path.exists(path) else {}
        prev.setdefault('cases', {}).update(timings)
        prev.update({'cores': os.cpu_count(), 'torch_threads': torch.get_num_threads(), 'host': 'build container (no GPU)',
                     'what': "seconds for mapping + synthesis: the reference's own Python (ops -> its _ref implementations, fp32 forced off-GPU; "
                             'rasteriser / flood fill = oracle/raster_ref.c stand-ins) and the oracle (kind "port") on the same inputs',
                     'script': 'oracle/pin_against_reference.py'})
        json.dump(prev, open(path, 'w'), indent=1)
    print('PIN', 'OK' if overall_ok else 'FAILED')
    return 0 if overall_ok else 1


if __name__ == '__main__':
    sys.exit(main())
