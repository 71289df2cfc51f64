"""oracle/b1_route.py — the reference's generator CODE PATTERN on an injected operator layer (TEST INFRASTRUCTURE).

What an un-reloaded `next3d_ffhq_512.pkl` executes (gen_samples_next3d.py:119,150-151: `--reload_modules False` is the default) is
the reference's own Python — training_avatar_texture/{networks_stylegan2, networks_stylegan2_styleunet, superresolution,
triplane_next3d}.py and volumetric_rendering/*.py — with only `torch_utils.ops.*`, `pytorch3d` and `cv2` resolved to next3d_amd
(`next3d_amd.install_dropin()`).  /root/reference cannot travel to the GPU box, so this module re-creates exactly that situation
from the oracle's restatement of the same code: oracle/{ops,networks,renderer,generator}.py are RE-INSTANTIATED (the module source
executed a second time under another name) with

  * the operator entry points — bias_act, upfirdn2d (+ upsample2d / downsample2d), conv2d_resample — bound to the operator layer
    under test (`next3d_amd.torch_utils.ops`): the FUSED modulated convolution reaches it as conv2d_resample(groups = batch) on
    per-sample weights [N*O, I, k, k], the way training_avatar_texture/networks_stylegan2.py:82-88 calls it;
  * the two third-party calls — pytorch3d rasterize_meshes (vr/renderer.py:415-424) and cv2.floodFill (:593) — bound to
    `next3d_amd.shims`;
  * everything else — weight modulation / demodulation, addmm, grid_sample, the renderer's tensor program, the `.cpu().numpy()`
    round trips of fill_mouth and gen_mouth_mask — plain torch on the tensors' device, as the reference does it;
  * the float16 super-resolution blocks as REAL float16 tensors (`synthesis_block_fp16` below restates SynthesisBlock.forward with
    use_fp16 and not force_fp32, training/networks_stylegan2.py:417-452, and modulated_conv2d's float16 pre-normalisation :55-59).

Used by tests/test_b1_route_gpu.py (parity of this route with the reference goldens) and bench.py's `b1_route` extra (its
throughput).  Nothing under next3d_amd/ imports this file.
"""
import importlib
import importlib.util

import numpy as np
import torch
import torch.nn.functional as F

_SQRT2 = float(np.sqrt(2))


def _clone(name):
    """A second instance of oracle/<name>.py (fresh module object, same source)."""
    src = importlib.import_module(f'oracle.{name}')
    spec = importlib.util.spec_from_file_location(f'oracle._b1_{name}', src.__file__)
    mod = importlib.util.module_from_spec(spec)
    mod.__package__ = 'oracle'
    spec.loader.exec_module(mod)
    return mod


class Route:
    """`Route(device)`: .mapping(P, z, c, rk, ...) / .synthesis(P, ws, c, v, uv_face_mask, rk, jitter, u, ...) with the signatures
    of oracle/generator.py, every tensor on `device`, operators = next3d_amd.torch_utils.ops, third party = next3d_amd.shims."""

    def __init__(self, device, op_pkg=None, shim_pkg=None):
        self.device = torch.device(device)
        T = op_pkg if op_pkg is not None else importlib.import_module('next3d_amd.torch_utils.ops')
        for m in ('bias_act', 'upfirdn2d', 'conv2d_resample'):
            importlib.import_module(f'{T.__name__}.{m}')
        S = shim_pkg if shim_pkg is not None else importlib.import_module('next3d_amd.shims')
        rast = importlib.import_module(f'{S.__name__}.pytorch3d.renderer.mesh')
        structures = importlib.import_module(f'{S.__name__}.pytorch3d.structures')
        cv2 = importlib.import_module(f'{S.__name__}.cv2')
        dev = self.device
        filters = {}

        def on_dev(f):
            if f is None or f.device == dev:
                return f
            if id(f) not in filters:
                filters[id(f)] = (f, f.to(dev))          # (the CPU tensor is kept so that its id stays taken)
            return filters[id(f)][1]

        # ---- operator layer -------------------------------------------------------------------------------------------------
        ops = _clone('ops')

        def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None):
            return T.bias_act.bias_act(x, b, dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp)

        def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1.0):
            return T.upfirdn2d.upfirdn2d(x, on_dev(f), up=up, down=down, padding=padding, flip_filter=flip_filter, gain=gain)

        def upsample2d(x, f, up=2, gain=1.0):
            return T.upfirdn2d.upsample2d(x, on_dev(f), up=up, gain=gain)

        def downsample2d(x, f, down=2, gain=1.0):
            return T.upfirdn2d.downsample2d(x, on_dev(f), down=down, gain=gain)

        def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True, quant=None):
            assert quant is None
            return T.conv2d_resample.conv2d_resample(x=x, w=w, f=on_dev(f), up=up, down=down, padding=padding, groups=groups, flip_weight=flip_weight)

        ops.__dict__.update(bias_act=bias_act, upfirdn2d=upfirdn2d, upsample2d=upsample2d, downsample2d=downsample2d, conv2d_resample=conv2d_resample)
        self.ops = ops

        # ---- networks ------------------------------------------------------------------------------------------------------
        net = _clone('networks')
        net.ops = ops

        def modconv16(weight, x, styles, up, demodulate):
            """modulated_conv2d, fused branch, x.dtype == float16 (training_avatar_texture/networks_stylegan2.py:51-91)."""
            n = x.shape[0]
            o, i, kh, kw = weight.shape
            if demodulate:                                                          # :55-59 pre-normalisation against float16 overflow
                weight = weight * (1 / np.sqrt(i * kh * kw) / weight.norm(float('inf'), dim=[1, 2, 3], keepdim=True))
                styles = styles / styles.norm(float('inf'), dim=1, keepdim=True)
            w = weight.unsqueeze(0) * styles.reshape(n, 1, -1, 1, 1)
            if demodulate:
                w = w * (w.square().sum(dim=[2, 3, 4]) + 1e-8).rsqrt().reshape(n, -1, 1, 1, 1)
            y = ops.conv2d_resample(x.reshape(1, -1, *x.shape[2:]), w.reshape(-1, i, kh, kw).to(x.dtype), f=net.FIR, up=up, padding=kh // 2,
                                    groups=n, flip_weight=(up == 1))
            return y.reshape(n, -1, *y.shape[2:])

        def synthesis_block_fp16(P, prefix, x, img, ws, conv_clamp=256, cpu_rounding=False, noise_mode='none'):
            """SynthesisBlock.forward with use_fp16 and not force_fp32 (training/networks_stylegan2.py:417-452)."""
            w0, w1, w2 = ws.unbind(dim=1)
            aff = lambda k, w: ops.fully_connected(w, P[f'{prefix}.{k}.affine.weight'], P[f'{prefix}.{k}.affine.bias'])
            x = x.to(torch.float16)
            for k, w, up in (('conv0', w0, 2), ('conv1', w1, 1)):
                x = modconv16(P[f'{prefix}.{k}.weight'], x, aff(k, w), up, True)
                if noise_mode == 'const':
                    x = x.add_(P[f'{prefix}.{k}.noise_const'] * P[f'{prefix}.{k}.noise_strength'])
                x = ops.bias_act(x, P[f'{prefix}.{k}.bias'].to(x.dtype), act='lrelu', gain=_SQRT2, clamp=conv_clamp)
            wt = P[f'{prefix}.torgb.weight']
            y = modconv16(wt, x, aff('torgb', w2) * (1.0 / np.sqrt(wt.shape[1] * wt.shape[2] ** 2)), 1, False)
            y = ops.bias_act(y, P[f'{prefix}.torgb.bias'].to(x.dtype), clamp=conv_clamp)
            if img is None:
                return x, y.to(torch.float32)
            img = ops.upsample2d(img, net.FIR)
            return x, img.add_(y.to(torch.float32))

        net.synthesis_block_fp16 = synthesis_block_fp16
        self.networks = net

        # ---- renderer: the reference's tensor program, on the device --------------------------------------------------------------
        ren = _clone('renderer')
        ren.ops = ops
        ren.PLANE_AXES = ren.PLANE_AXES.to(dev)
        self.renderer = ren

        # ---- rasterisation: the reference's own code around the two third-party calls ----------------------------------------------
        ras = _clone('raster')

        def orth_project(pts, tform, orth_shift, orth_scale):
            return ras_orth(pts, tform.to(pts.device), orth_shift.to(pts.device), orth_scale.to(pts.device))
        ras_orth = ras.orth_project

        def pytorch3d_rasterizer(vertices, faces, face_attrs, image_size=256):
            """vr/renderer.py:401-440 (Pytorch3dRasterizer.forward) on the rasterize_meshes shim."""
            fixed = vertices.clone()
            fixed[..., :2] = -fixed[..., :2]
            n = fixed.shape[0]
            meshes = structures.Meshes(verts=fixed.float(), faces=faces.long()[None].expand(n, -1, -1))
            p2f, _, bary, _ = rast.rasterize_meshes(meshes, image_size=image_size, blur_radius=0.0, faces_per_pixel=1, bin_size=None,
                                                    max_faces_per_bin=None, perspective_correct=False, cull_backfaces=True)
            vis = (p2f > -1).float()
            N, H, W, K, _ = bary.shape
            D = face_attrs.shape[-1]
            attrs = face_attrs.expand(N, -1, -1, -1).reshape(-1, 3, D)
            mask = p2f == -1
            idx = p2f.clone()
            idx[mask] = 0
            vals = attrs.gather(0, idx.view(N * H * W * K, 1, 1).expand(N * H * W * K, 3, D)).view(N, H, W, K, 3, D)
            pix = (bary[..., None] * vals).sum(dim=-2)
            pix[mask] = 0
            pix = pix[:, :, :, 0].permute(0, 3, 1, 2)
            return torch.cat([pix, vis[:, :, :, 0][:, None]], dim=1)

        def fill_mouth(images):
            """vr/renderer.py:583-602 (fill_mouth), host round trip per image included."""
            out = []
            for image in images:
                img = image[0].cpu().numpy() * 255.
                cp = np.ascontiguousarray(img.copy(), dtype=np.float32)
                h, w = cp.shape[:2]
                cv2.floodFill(cp, np.zeros([h + 2, w + 2], np.uint8), (0, 0), (255, 255, 255), (0, 0, 0), (254, 254, 254), cv2.FLOODFILL_FIXED_RANGE)
                out.append((torch.tensor(cp).to(images.device).to(torch.float32) / 127.5 - 1).unsqueeze(0))
            mm = torch.stack(out, 0)
            mm = ((mm * 2. - 1.) * -1. + 1.) / 2.
            return (images + mm).clip(0, 1)

        ras.__dict__.update(orth_project=orth_project, pytorch3d_rasterizer=pytorch3d_rasterizer, fill_mouth=fill_mouth)
        self.raster = ras

        gen = _clone('generator')
        gen.networks, gen.raster, gen.renderer = net, ras, ren
        self.generator = gen

    def to_device(self, P):
        return {k: v.to(self.device) for k, v in P.items()}

    def mapping(self, P, z, c, rendering_kwargs, **kw):
        with torch.device(self.device):
            return self.generator.mapping(P, z.to(self.device), c.to(self.device), rendering_kwargs, **kw)

    def synthesis(self, P, ws, c, v, uv_face_mask, rendering_kwargs, jitter, u, **kw):
        d = self.device
        with torch.device(d):                    # factory calls of the restated code (arange, ones, linspace, tensor) land on the device
            return self.generator.synthesis(P, ws.to(d), c.to(d), v.to(d), uv_face_mask.to(d), rendering_kwargs, jitter.to(d), u.to(d), **kw)
