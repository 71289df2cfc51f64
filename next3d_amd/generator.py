"""TriPlaneGenerator — the B2 (model-level) drop-in boundary.

Same constructor / `mapping` / `synthesis` / `forward` signatures, state-dict names and attributes as the
reference class (training_avatar_texture/triplane_next3d.py:40-344), so `gen_samples_next3d.py`,
`gen_videos_next3d.py` and `reenact_avatar_next3d.py` can construct it with `--reload_modules=True`
(`TriPlaneGenerator(*G.init_args, **G.init_kwargs)` + `misc.copy_params_and_buffers`).  The forward pass
itself is a different program: every arithmetic step runs in libn3d.so HIP kernels, there are no
device->host round trips (the reference has 4N+1 per frame: fill_mouth and gen_mouth_mask) and no
dynamic shapes (the mouth box lives in device memory).
"""
import os

import numpy as np
import torch

from . import _lib, layers, mesh, networks, spec

_BUFFER_LEAVES = ('noise_const', 'resample_filter', 'w_avg', 'dense_faces', 'faces', 'raw_uvcoords', 'uvcoords', 'uvfaces',
                  'face_uvcoords')
RENDER_DECODER_SPLIT = True        # render: with layers.PRECISION == 'bf16x3' the OSGDecoder runs on split-bf16 MFMAs too (n3d_render_opts.decoder_split_bf16; False: float32-input MFMAs)
RASTER_ON_SIDE_STREAM = 1          # _planes: batch sizes up to which the mesh rasterisation runs in front of the static backbone on the side stream (0 = never)
RENDERING_VIEWS = [[0, 0, 0], [0, 90, 0], [0, -90, 0], [90, 0, 0]]        # reference triplane_next3d.py:140-145


def angle2matrix(angles_deg):
    """[3] XYZ angles in degrees -> [1,3,3] rotation matrix (reference volumetric_rendering/renderer.py:518-547)."""
    a = torch.tensor(angles_deg, dtype=torch.float32).reshape(1, 3) * (np.pi) / 180.
    s, c = torch.sin(a), torch.cos(a)
    cx, cy, cz = c[:, 0], c[:, 1], c[:, 2]
    sx, sy, sz = s[:, 0], s[:, 1], s[:, 2]
    R = torch.stack([cz * cy, cz * sy * sx - sz * cx, cz * sy * cx + sz * sx,
                     sz * cy, sz * sy * sx + cz * cx, sz * sy * cx - cz * sx,
                     -sy, cy * sx, cy * cx], dim=0)
    return torch.reshape(R, (-1, 3, 3))


def _require_hip(dev):
    if dev.type != 'cuda':
        raise RuntimeError('TriPlaneGenerator runs on a HIP device only: move it with .to("cuda") '
                           '(no CPU fallback; the CPU restatement is oracle/, test infrastructure)')


_STRUCT_GEN = [0]        # bumped whenever a parameter / buffer OBJECT is (re)assigned anywhere in a generator's module tree


class _Tracked(torch.nn.Module):
    """Module whose parameter / buffer assignments bump `_STRUCT_GEN` (O(1) check per forward in `_check_params`: a parameter replaced by
    assignment or a re-registered buffer invalidates the prepared weights, caches and captured graphs on the NEXT call)."""

    def __setattr__(self, name, value):
        # (a Parameter value registers a parameter; a name already in _parameters / _buffers replaces one; a plain tensor under a new name is an ordinary attribute)
        if isinstance(value, torch.nn.Parameter) or name in self.__dict__.get('_parameters', ()) or name in self.__dict__.get('_buffers', ()):
            _STRUCT_GEN[0] += 1
        super().__setattr__(name, value)

    def __delattr__(self, name):
        if name in self.__dict__.get('_parameters', ()) or name in self.__dict__.get('_buffers', ()):
            _STRUCT_GEN[0] += 1
        super().__delattr__(name)

    def register_parameter(self, name, param):
        _STRUCT_GEN[0] += 1
        super().register_parameter(name, param)

    def register_buffer(self, name, tensor, persistent=True):
        _STRUCT_GEN[0] += 1
        super().register_buffer(name, tensor, persistent=persistent)


class _Node(_Tracked):
    """Anonymous container used to reproduce the reference's dotted parameter names."""


def _attach(root, dotted, tensor, is_buffer):
    parts = dotted.split('.')
    mod = root
    for p in parts[:-1]:
        if not hasattr(mod, p):
            mod.add_module(p, _Node())
        mod = getattr(mod, p)
    if is_buffer:
        mod.register_buffer(parts[-1], tensor)
    else:
        mod.register_parameter(parts[-1], torch.nn.Parameter(tensor, requires_grad=False))


class TriPlaneGenerator(_Tracked):
    def __init__(self, z_dim, c_dim, w_dim, img_resolution, img_channels, topology_path, sr_num_fp16_res=0,
                 mapping_kwargs={}, rendering_kwargs={}, sr_kwargs={}, uv_face_mask=None, **synthesis_kwargs):
        super().__init__()
        # the super-resolution module decides the output resolution (tat/triplane_next3d.py:66: construct_class_by_name(superresolution_module, img_resolution=...);
        # every module asserts its own img_resolution): 8XDC (ffhq-512) / 8X -> 512, 4X -> 256, 2X -> 128 (spec.SR_MODULES)
        self.sr_class, (sr_res, _, _, _, _) = spec.sr_module(rendering_kwargs.get('superresolution_module'))
        if (z_dim, c_dim, w_dim, img_channels) != (512, 25, 512, 3) or img_resolution != sr_res:
            raise RuntimeError(f'this build implements the next3d configuration z = w = 512, c = 25, 3 colours at the resolution of the super-resolution module '
                               f'({self.sr_class}: {sr_res} x {sr_res}); got z={z_dim} c={c_dim} w={w_dim} {img_resolution}x{img_resolution}x{img_channels}')
        # mapping depth: train_next3d.py passes map_depth = 2; without the key the reference's MappingNetwork builds its own default, 8 (tat/networks_stylegan2.py:207)
        self.mapping_layers = int(mapping_kwargs.get('num_layers', 8))
        if not 1 <= self.mapping_layers <= 16:
            raise RuntimeError(f'mapping_kwargs.num_layers = {self.mapping_layers}: 1..16 expected')
        # ... and the other options MappingNetwork honours (tat/networks_stylegan2.py:196-209) only at the values `mapping()` implements: a pickle built with another
        # one would produce other ws — refuse it instead of ignoring the key; unknown keys raise as the reference's constructor would
        mk_fixed = dict(embed_features=None, layer_features=None, activation='lrelu', lr_multiplier=0.01, w_avg_beta=0.998)
        for key, val in mapping_kwargs.items():
            if key == 'num_layers':
                continue
            if key not in mk_fixed:
                raise TypeError(f'TriPlaneGenerator: unexpected mapping keyword argument {key!r} (MappingNetwork takes num_layers, ' + ', '.join(mk_fixed) + ')')
            if val is not None and val != mk_fixed[key] and not (key in ('embed_features', 'layer_features') and val == 512):
                raise RuntimeError(f'mapping_kwargs[{key!r}] = {val!r}: this build implements the reference default ({mk_fixed[key]!r}) only')
        # the backbones' widths: channels_dict of `channel_base` / `channel_max` (tat/networks_stylegan2.py:614; ffhq-512: 32768 / 512) — RuntimeError for
        # widths the matrix-core kernels do not tile (spec.check_channels)
        self.channel_base, self.channel_max = int(synthesis_kwargs.get('channel_base', 32768)), int(synthesis_kwargs.get('channel_max', 512))
        spec.check_channels(self.channel_base, self.channel_max)
        # float16 blocks in the four StyleGAN2 / StyleUNet backbones: num_fp16_res > 0 — what legacy.load_network_pkl(force_fp16=True)
        # sets (legacy.py:49-59: num_fp16_res = 4, conv_clamp = 256; the ffhq-512 pickle itself has 0).  The blocks of resolution >=
        # fp16_resolution run on the f16 matrix-core kernels unless force_fp32 is passed (tat/networks_stylegan2.py:615-621, :548-562)
        # the block options this build implements are the next3d ones; anything else must not be ignored silently (the reference would build another network,
        # or raise TypeError for a key its blocks do not take)
        fixed = dict(architecture='skip', use_noise=True, activation='lrelu', resample_filter=[1, 3, 3, 1], kernel_size=3)     # (kernel_size: SynthesisLayer's, through layer_kwargs)
        for key, want in fixed.items():
            got = synthesis_kwargs.get(key, want)
            if (list(got) if key == 'resample_filter' else got) != want:
                raise RuntimeError(f'synthesis_kwargs[{key!r}] = {got!r}: this build implements {want!r} only')
        unknown = set(synthesis_kwargs) - set(fixed) - {'channel_base', 'channel_max', 'num_fp16_res', 'conv_clamp', 'fused_modconv_default', 'fp16_channels_last'}
        if unknown:
            raise TypeError(f'TriPlaneGenerator: unexpected synthesis keyword arguments {sorted(unknown)} (the reference\'s SynthesisBlock takes none of them)')
        # (absent keys take the reference classes' own defaults: SynthesisNetwork num_fp16_res = 4, SynthesisBlock conv_clamp = 256, tat/networks_stylegan2.py:603,376;
        #  train_next3d.py writes num_fp16_res = 0 / conv_clamp = None into every next3d pickle's init_kwargs)
        nfp16 = int(synthesis_kwargs.get('num_fp16_res', 4) or 0)
        self.backbone_fp16_resolution = max(2 ** (8 + 1 - nfp16), 8) if nfp16 > 0 else None            # the backbones are 256 x 256 networks
        self.backbone_conv_clamp = synthesis_kwargs.get('conv_clamp', 256)
        self.init_args = (z_dim, c_dim, w_dim, img_resolution, img_channels, topology_path)
        self.init_kwargs = dict(sr_num_fp16_res=sr_num_fp16_res, mapping_kwargs=mapping_kwargs,
                                rendering_kwargs=rendering_kwargs, sr_kwargs=sr_kwargs, **synthesis_kwargs)
        self.z_dim, self.c_dim, self.w_dim = z_dim, c_dim, w_dim
        self.img_resolution, self.img_channels = img_resolution, img_channels
        self.topology_path = topology_path
        self.neural_rendering_resolution = 64
        self.rendering_kwargs = rendering_kwargs
        self.load_lms = True
        self.uv_resolution = 256
        self.fill_mouth = True
        self.orth_scale = torch.tensor([[5.0]])
        self.orth_shift = torch.tensor([[0, -0.01, -0.01]])
        self.overlap_static = os.environ.get('N3D_OVERLAP_STATIC', '1') != '0'

        # parameters / buffers under the reference's names, reference init distributions (randn, affine bias 1, zeros)
        mb = mesh.mesh_buffers_from_obj(topology_path) if isinstance(topology_path, str) else mesh.mesh_buffers(*topology_path)
        for name, (shape, kind) in spec.build_spec(self.sr_class, self.channel_base, self.channel_max, self.mapping_layers).items():
            leaf = name.rsplit('.', 1)[-1]
            if kind == 'mesh':
                t = mb[name]
            elif kind == 'fir':
                t = spec._fir()
            elif kind in ('randn',):
                t = torch.randn(shape)
            elif kind == 'randn_lr':
                t = torch.randn(shape) / 0.01
            elif kind == 'affine_bias':
                t = torch.ones(shape)
            else:
                t = torch.zeros(shape)
            _attach(self, name, t, is_buffer=(leaf in _BUFFER_LEAVES))

        # plain attributes callers of the reference read on the sub-networks (viz/renderer.py:277,321: G.backbone.num_ws,
        # G.backbone.mapping.w_avg; triplane_next3d.py:65: mapping_ws = 2 x texture_backbone.num_ws)
        for net, (res, ch) in (('texture_backbone', (256, 32)), ('backbone', (256, 96)), ('mouth_backbone', (256, 32)),
                               ('neural_blending', (256, 32))):
            node = getattr(self, net)
            node.z_dim, node.c_dim, node.w_dim, node.img_resolution, node.img_channels, node.num_ws = z_dim, c_dim, w_dim, res, ch, 14
            node.synthesis.num_ws, node.synthesis.img_resolution, node.synthesis.img_channels = 14, res, ch
            if hasattr(node, 'mapping'):
                node.mapping.num_ws = 28 if net == 'backbone' else 14
                node.mapping.z_dim, node.mapping.c_dim, node.mapping.w_dim, node.mapping.num_layers = z_dim, c_dim, w_dim, self.mapping_layers
        self.superresolution.input_resolution = spec.SR_MODULES[self.sr_class][1]

        if uv_face_mask is None:      # reference: cv2.imread('data/ffhq/uv_face_eye_mask.png') (triplane_next3d.py:91)
            uv_face_mask = self._load_uv_mask('data/ffhq/uv_face_eye_mask.png')
        self.uv_face_mask = torch.nn.functional.interpolate(uv_face_mask.float(), [256, 256])
        # super-resolution settings the reference derives in its constructors (superresolution.py:264-278, triplane_next3d.py:181-183)
        self.sr_conv_clamp = 256 if sr_num_fp16_res > 0 else None        # (256 if use_fp16 else None); kept under force_fp32
        self.sr_use_fp16 = sr_num_fp16_res > 0                           # superresolution.py:269: the SR blocks run in fp16 unless force_fp32
        self._drop_derived()

    def _drop_derived(self):
        """Forget everything derived from the parameters: prepared weights AND the cross-call caches (blended planes,
        identity networks) — they are stale after a weight reload / device move / in-place update."""
        self._prepared = None
        self._last_planes = None
        self._identity_cache = None
        self._cache_gen = getattr(self, '_cache_gen', 0) + 1      # generation of the two caches above (synthesis_graph keys on it)
        self._param_stamp = None
        self._ptensors = None           # (cached parameter / buffer list of _check_params)
        self._graphs = None             # captured HIP graphs (synthesis_graph): they hold pointers into the prepared weights and caches
        self._tree_stamp = None         # (cheap structure stamp of _check_params)
        from .torch_utils.ops import conv2d_gradfix
        conv2d_gradfix.clear_prep_cache()     # the operator boundary's per-tensor cache cannot see `.data` updates: refresh() covers them

    def _set_cache(self, name, value):
        """(Re)assign a cross-call cache (`_last_planes` / `_identity_cache`).  Captured graphs bake in the device pointers of the
        cache they were captured with: a new cache gets a new generation number — never an id(), which CPython re-uses — and the
        graphs captured against the old one are dropped with their private memory pools (they could only ever replay stale data)."""
        setattr(self, name, value)
        self._cache_gen += 1
        if self._graphs:
            self._graphs = {sig: e for sig, e in self._graphs.items() if not e[3]}       # e[3]: the graph reads a cache

    # ------------------------------------------------------------------ plumbing
    @staticmethod
    def _load_uv_mask(path):
        """The reference reads this cwd-relative file with cv2 and keeps channel 0 of the BGR image, i.e. BLUE
        (triplane_next3d.py:91); a missing file crashes it.  Same here: no silent stand-in — pass `uv_face_mask=` explicitly
        (as the demo / tests do with mesh.synthetic_uv_face_mask()) to run without the asset."""
        if not os.path.exists(path):
            raise RuntimeError(f'{path} not found (relative to the current directory, as in the reference); run from the '
                               'repository root that holds data/ffhq/, or pass uv_face_mask= to the constructor')
        from PIL import Image
        m = np.asarray(Image.open(path).convert('RGB'), dtype=np.float32)[:, :, 2] / 255.       # cv2 BGR channel 0 == RGB channel 2
        return torch.from_numpy(m)[None, None].contiguous()

    def _apply(self, fn, *a, **k):
        self._drop_derived()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._drop_derived()
        return super().load_state_dict(*a, **k)

    @property
    def device(self):
        return next(self.parameters()).device

    def refresh(self):
        """Drop derived constants and caches (in-place parameter updates are also detected by `_check_params`)."""
        self._drop_derived()

    def _check_params(self):
        """In-place updates (misc.copy_params_and_buffers after a first forward, an optimizer step) bump the tensors' version
        counters: a changed sum invalidates the prepared weights and the caches.  Called once per mapping / synthesis call.
        The list of tensors is cached (walking the 674-entry module tree costs ~0.4 ms, twice per frame: at batch 1 the eager call is
        bound by the host).  A parameter / buffer OBJECT replaced by assignment bumps `_STRUCT_GEN` (`_Tracked`), which is compared on
        every call; every 256th call the tree is re-walked anyway (assignments that bypass the module API, `m._parameters[k] = ...`).
        `p.data.copy_()` bypasses the version counter: call refresh()."""
        ts = self.__dict__.get('_ptensors')
        self._check_calls = self.__dict__.get('_check_calls', 0) + 1
        if ts is None or self.__dict__.get('_tree_stamp') != _STRUCT_GEN[0] or (self._check_calls & 255) == 0:
            fresh = list(self.parameters()) + list(self.buffers())
            if ts is not None and (len(fresh) != len(ts) or any(a is not b for a, b in zip(fresh, ts))):
                self._drop_derived()
            ts = self._ptensors = fresh
            self._tree_stamp = _STRUCT_GEN[0]
        stamp = 0
        try:
            for t in ts:
                stamp += t._version
        except RuntimeError:            # inference tensors (a model built or loaded under torch.inference_mode()) track no version: they cannot be
            stamp = 0                   # updated in place either — only the tensors that do track one are summed
            for t in ts:
                if not t.is_inference():
                    stamp += t._version
        if self._param_stamp is not None and stamp != self._param_stamp:
            self._drop_derived()
            self._ptensors, self._tree_stamp = ts, _STRUCT_GEN[0]
        self._param_stamp = stamp

    def _prep(self):
        """Derived per-model constants: K-major conv weights, squared-weight sums, scaled decoder weights, mesh tables."""
        if self._prepared is not None:
            return self._prepared
        dev = self.device
        _require_hip(dev)
        _lib.lib()
        P = {k: v.detach() for k, v in self.state_dict().items()}
        S = type('Prepared', (), {})()
        S.P = P
        bk = dict(fp16_resolution=self.backbone_fp16_resolution, conv_clamp=self.backbone_conv_clamp)
        S.texture = networks.SynthesisNet(P, 'texture_backbone.synthesis', **bk)
        S.static = networks.SynthesisNet(P, 'backbone.synthesis', **bk)
        S.mouth = networks.StyleUNet(P, 'mouth_backbone.synthesis', in_size=64, final_size=4, num_cond_res=64, **bk)
        S.blend = networks.StyleUNet(P, 'neural_blending.synthesis', in_size=256, final_size=32, num_cond_res=256, **bk)
        S.sr = networks.SuperRes8XDC(P, 'superresolution', conv_clamp=self.sr_conv_clamp, sr_class=self.sr_class)
        # every style affine and demodulation coefficient of the five networks in TWO launches per forward (layers.StyleBank over the
        # full [N, 28, 512] latents: the texture backbone reads slots 14-27, the others 0-13, the super-resolution slot 13)
        nw = S.texture.num_ws
        ent = [(l, slot + nw, k) for (l, slot, k) in S.texture.bank.entries]
        for net in (S.static, S.mouth, S.blend):
            ent += list(net.bank.entries)
        ent += S.sr.bank_entries(nw - 1)
        S.all_bank = layers.StyleBank(ent, dev)
        S.one_bank = True           # False: every network computes its own styles (10 instead of 2 affine launches; tests)
        lr = float(self.rendering_kwargs.get('decoder_lr_mul', 1))
        S.dec_w1 = (P['decoder.net.0.weight'] * (lr / np.sqrt(32))).contiguous()
        S.dec_b1 = (P['decoder.net.0.bias'] * lr).contiguous() if lr != 1 else P['decoder.net.0.bias']
        w2 = P['decoder.net.2.weight'] * (lr / np.sqrt(64))                     # [33, 64]
        S.dec_w2 = torch.cat([w2.t(), torch.zeros(64, 1, device=w2.device)], dim=1).contiguous()   # [64, 34]: n3d_render_rays' w2t
        S.dec_b2 = (P['decoder.net.2.bias'] * lr).contiguous() if lr != 1 else P['decoder.net.2.bias']
        S.faces = P['faces'][0][:, [0, 2, 1]].to(torch.int32).contiguous()                 # triplane_next3d.py:207
        S.face_uv = P['face_uvcoords'][0][:, [0, 2, 1]].contiguous()                       # :208
        S.rot = torch.cat([angle2matrix(a) for a in RENDERING_VIEWS], 0).to(dev).contiguous()
        S.uv_mask = self.uv_face_mask.to(dev)[0, 0].contiguous()
        S.side_streams = {}         # launch stream -> its side stream (callers may pipeline calls on several streams)
        S.alpha_views = torch.tensor([0, 1, 3], dtype=torch.int64, device=dev)
        S.tlin = {}
        S.zero_c = {}               # batch size -> zeros [N, 25] (mapping with c_gen_conditioning_zero)
        self._prepared = S
        return S

    # ------------------------------------------------------------------ reference API
    def mapping(self, z, c, truncation_psi=1, truncation_cutoff=None, update_emas=False):
        """reference triplane_next3d.py:111-115 + MappingNetwork.forward (networks_stylegan2.py:233-268)."""
        self._check_params()
        S = self._prep()
        P = S.P
        # the scripts' z is float64 (np.random.RandomState(seed).randn): the float32 conversion of :239 happens inside the normalisation kernel
        z = z.to(device=self.device) if z.dtype == torch.float64 else z.to(device=self.device, dtype=torch.float32)
        n = z.shape[0]
        scale = float(self.rendering_kwargs.get('c_scale', 0))
        if self.rendering_kwargs['c_gen_conditioning_zero'] or scale == 0.0:
            # zeros_like(c)[:, :25] * c_scale (:112-114): one cached zero tensor per batch size instead of a fill + a multiply launch per call
            c = S.zero_c.get(n)
            if c is None:
                c = S.zero_c[n] = torch.zeros(n, 25, dtype=torch.float32, device=self.device)
            scale = 1.0
        else:
            c = c[:, :25].to(device=self.device, dtype=torch.float32)
        pre = 'backbone.mapping'
        L = _lib.lib()
        x = torch.empty(n, 1024, dtype=torch.float32, device=self.device)          # cat([norm(z), norm(embed(c))], 1)
        z = z.contiguous()              # (never pass a temporary to _lib.ptr: it is freed before the launch and its block can be re-used)
        _lib.check((L.n3d_normalize_2nd_moment_f64 if z.dtype == torch.float64 else L.n3d_normalize_2nd_moment)(_lib.ptr(z), _lib.ptr(x), n, 512, 1024, 1e-8, _lib.stream()))
        if scale != 1.0:                 # (c * c_scale is exact for c_scale = 1, the ffhq configuration; any other scale: libn3d.so's fma, not a torch multiply)
            from .torch_utils.ops import fma as _fma
            c = _fma.fma(c.contiguous(), torch.full((1, 1), scale, dtype=torch.float32, device=self.device).expand(n, 25).contiguous(), torch.zeros(n, 25, dtype=torch.float32, device=self.device))
        y = layers.fc(c.contiguous(), P[f'{pre}.embed.weight'], P[f'{pre}.embed.bias'], wgain=1 / np.sqrt(25))
        _lib.check(L.n3d_normalize_2nd_moment(_lib.ptr(y), _lib.c_void_p(x.data_ptr() + 512 * 4), n, 512, 1024, 1e-8, _lib.stream()))
        for i in range(self.mapping_layers):
            w = P[f'{pre}.fc{i}.weight']
            x = layers.fc(x, w, P[f'{pre}.fc{i}.bias'], wgain=0.01 / np.sqrt(w.shape[1]), bgain=0.01, act='lrelu')
        num_ws = 2 * S.texture.num_ws
        ws = torch.empty(n, num_ws, 512, dtype=torch.float32, device=self.device)
        trunc = truncation_psi != 1
        cutoff = num_ws if truncation_cutoff is None else min(int(truncation_cutoff), num_ws)
        _lib.check(L.n3d_truncate_ws(_lib.ptr(x), _lib.ptr(P[f'{pre}.w_avg']) if trunc else None, _lib.ptr(ws), n, num_ws, 512,
                                     cutoff if trunc else 0, float(truncation_psi), _lib.stream()))
        return ws

    def raster_geometry(self, v, lms, _all_views=False):
        """The texture-independent half of `rasterize` (reference triplane_next3d.py:190-222): z-buffer the four orthographic
        views of the mesh -> (uv sampling grid [N*4,256,256,2], alpha [N,3,256,256], mouth box [N,4] int32).  It depends only
        on the vertices, so `_planes` runs it first.  `_all_views`: alpha as the kernels leave it, [N,4,256,256] (n3d_blend_planes_views picks
        views 0 / 1 / 3 itself: no index_select copy on the forward path)."""
        S = self._prep()
        dev, N, V, Lm, F = v.device, v.shape[0], v.shape[1], lms.shape[1], S.faces.shape[0]
        views, H, W = len(RENDERING_VIEWS), 256, 256
        f32 = dict(dtype=torch.float32, device=dev)
        tv = torch.empty(N * views * V * 3, **f32)
        zbuf = torch.empty(N * views * H * W, dtype=torch.int64, device=dev)
        grid = torch.empty(N * views, H, W, 2, **f32)
        alpha4 = torch.empty(N, views, H, W, **f32)
        lm2d = torch.empty(N, Lm, 2, **f32)
        sh = self.orth_shift.reshape(-1).tolist()
        L = _lib.lib()
        # contiguous copies must stay referenced until the launch: a temporary passed straight to _lib.ptr() is freed at once and
        # the NEXT temporary (the landmarks) may be carved out of the same block, overwriting the vertices before the kernel runs
        v, lms = v.contiguous(), lms.contiguous()
        _lib.check(L.n3d_rasterize_views(_lib.ptr(v), _lib.ptr(lms), _lib.ptr(S.rot), _lib.ptr(S.faces),
                                         _lib.ptr(S.face_uv), _lib.ptr(S.uv_mask), S.uv_mask.shape[0], S.uv_mask.shape[1],
                                         _lib.ptr(tv), _lib.ptr(zbuf), _lib.ptr(grid), _lib.ptr(alpha4), _lib.ptr(lm2d), N, V, Lm, F,
                                         views, H, W, sh[0], sh[1], sh[2], float(self.orth_scale.item()),
                                         1 if self.fill_mouth else 0, 1, _lib.stream()))
        bbox = torch.empty(N, 4, dtype=torch.int32, device=dev)
        _lib.check(L.n3d_mouth_bbox(_lib.ptr(lm2d), _lib.ptr(bbox), N, Lm, _lib.stream()))
        if _all_views:
            return grid, alpha4, bbox
        alpha = alpha4.index_select(1, S.alpha_views)          # views 0 (front), 1 (side; view 2's alpha is unused, :226), 3 (top)
        return grid, alpha, bbox

    def project_textures(self, textures, grid):
        """reference triplane_next3d.py:223-230: sample the neural texture through the rasterised uv grids -> [front, side, top]."""
        N, views, H, W = textures.shape[0], len(RENDERING_VIEWS), 256, 256
        import ctypes
        cfg = ((0, -1), (1, 2), (3, -1))                     # front, side (views 1 + 2), top
        planes = [torch.empty(N, textures.shape[1], H, W, dtype=torch.float32, device=textures.device) for _ in cfg]
        outs = (ctypes.c_void_p * 3)(*[p.data_ptr() for p in planes])
        va = (ctypes.c_int * 3)(*[a for a, _ in cfg])
        vb = (ctypes.c_int * 3)(*[b for _, b in cfg])
        _lib.check(_lib.lib().n3d_texture_project_planes(_lib.ptr(textures), _lib.ptr(grid), outs, va, vb, 3, N, textures.shape[1],
                                                         textures.shape[2], textures.shape[3], H, W, views, _lib.stream()))
        return planes

    def rasterize(self, v, lms, textures):
        """reference triplane_next3d.py:190-230 -> ([front, side, top] each [N,32,256,256], alpha [N,3,256,256], bbox [N,4] int32)."""
        grid, alpha, bbox = self.raster_geometry(v, lms)
        return self.project_textures(textures, grid), alpha, bbox

    def _unpack(self, v, c):
        """v [N, 5023 + 68, 3], c [N, 25] (device float32, unit inner strides) -> dense (verts, lms, cam2world [N,16], intrinsics [N,9]) in ONE launch
        (n3d_unpack_inputs) instead of four torch copies."""
        n = v.shape[0]
        f32 = dict(dtype=torch.float32, device=v.device)
        verts, lms = torch.empty(n, 5023, 3, **f32), torch.empty(n, v.shape[1] - 5023, 3, **f32)
        cam, intr = torch.empty(n, 16, **f32), torch.empty(n, 9, **f32)
        _lib.check(_lib.lib().n3d_unpack_inputs(_lib.ptr(v), v.stride(0), _lib.ptr(c), c.stride(0), _lib.ptr(verts), _lib.ptr(lms), _lib.ptr(cam), _lib.ptr(intr),
                                                n, 5023, v.shape[1] - 5023, _lib.stream()))
        return verts, lms, cam, intr

    def _planes(self, ws, v, noise_mode, cache_identity=False, use_cached_identity=False, bank=None, force_fp32=False, _unpacked=None):
        """Everything up to the blended tri-planes (channels-last [N,3,256,256,32]).  `cache_identity` keeps the two
        latent-only results (neural texture, static tri-planes); `use_cached_identity` re-uses them for a new mesh `v` (the
        reenactment loop, reenact_avatar_next3d.py:139-160: one identity, one mesh per frame)."""
        S = self._prep()
        L = _lib.lib()
        if not self.load_lms:
            raise RuntimeError('load_lms=False is not supported: the mouth branch needs the 68 landmarks')
        if _unpacked is not None:
            v, lms = _unpacked
        else:
            v = v.to(device=self.device, dtype=torch.float32)
            v, lms = v[:, :5023], v[:, 5023:]
        N = ws.shape[0]
        nw = S.texture.num_ws
        eg3d_ws, texture_ws = ws[:, :nw], ws[:, nw:]
        # The static tri-plane backbone depends only on the latents: it runs on a second HIP stream so that its low-resolution layers (a handful
        # of workgroups each) overlap the texture -> mouth -> blending chain.  The mesh rasterisation depends only on the vertices: for single-frame
        # calls (the scripts' pattern) it runs on that side stream too, in front of the static backbone, and the main stream waits for it only
        # before the texture projection — ~0.3 ms of latency-bound launches off the critical path (batch 1: 4.53 -> 4.37 ms per frame from the graph).
        # Not for larger batches: with several steps in flight on the chip's four hardware queues the early cross-stream wait costs more than it
        # hides (batch 4, three lanes: 396 -> 387 frames/s; one stream: 352 -> 356) — profiles/r04_raster_side_stream_ab.txt.
        # (Round 1 saw the rasteriser's results change from run to run while 8-wave split-bf16 convolution workgroups of ANOTHER stream shared the
        # CUs; round 2 bisected that to vector-L1-served gather loads of the vertex / face tables and the kernels now read those tables with
        # agent-scope loads: DESIGN.md 3.3, tests/test_path_kernels_gpu.py::test_rasteriser_reproducible_under_coresident_convolutions.)
        cur = torch.cuda.current_stream()
        ident = self._identity_cache if use_cached_identity else None
        static = None
        if ident is not None:                       # reenactment: same latents, new mesh -> only the mesh-dependent half runs
            grid, alpha, bbox = self.raster_geometry(v, lms, _all_views=True)
            textures, static = ident
        elif self.overlap_static:
            sstream = S.side_streams.get(cur.cuda_stream)
            if sstream is None:
                sstream = S.side_streams[cur.cuda_stream] = torch.cuda.Stream(device=ws.device)
            sstream.wait_stream(cur)
            side_raster = N <= RASTER_ON_SIDE_STREAM
            if not side_raster:
                grid, alpha, bbox = self.raster_geometry(v, lms, _all_views=True)
                sstream.wait_stream(cur)
            with torch.cuda.stream(sstream):
                if side_raster:
                    grid, alpha, bbox = self.raster_geometry(v, lms, _all_views=True)
                    raster_done = sstream.record_event()
                static = S.static(eg3d_ws, noise_mode, bank=bank, force_fp32=force_fp32)
            static.record_stream(cur)
            textures = S.texture(texture_ws, noise_mode, bank=bank, force_fp32=force_fp32)
            if side_raster:
                for t in (grid, alpha, bbox):
                    t.record_stream(cur)
                cur.wait_event(raster_done)
        else:
            grid, alpha, bbox = self.raster_geometry(v, lms, _all_views=True)
            textures = S.texture(texture_ws, noise_mode, bank=bank, force_fp32=force_fp32)
        front, side, top = self.project_textures(textures, grid)
        f32 = dict(dtype=torch.float32, device=ws.device)
        crop = torch.empty(N, 32, 64, 64, **f32)
        _lib.check(L.n3d_resize_aa(_lib.ptr(front), _lib.ptr(crop), _lib.ptr(bbox), None, N, 32, 256, 256, 64, 64, 0, _lib.stream()))
        mouths = S.mouth(crop, eg3d_ws, noise_mode, bank=bank, force_fp32=force_fp32)
        # the mouth is pasted into the front plane in place (the reference copies it first, :158-160); a copy is kept only
        # when the stage tensors are requested for inspection
        stitch_in = front.clone() if getattr(self, 'keep_stages', False) else front
        _lib.check(L.n3d_resize_aa(_lib.ptr(mouths), _lib.ptr(stitch_in), None, _lib.ptr(bbox), N, 32, 256, 256, 256, 256, 1, _lib.stream()))
        stitch = S.blend(stitch_in, eg3d_ws, noise_mode, bank=bank, force_fp32=force_fp32)
        if ident is None and self.overlap_static:
            cur.wait_stream(sstream)
        elif static is None:
            static = S.static(eg3d_ws, noise_mode, bank=bank, force_fp32=force_fp32)
        if cache_identity:
            self._set_cache('_identity_cache', (textures, static))
        planes = torch.empty(N, 3, 256, 256, 32, **f32)
        # alpha [N, 4 views, 256, 256]: front / side / top = views 0 / 1 / 3 (view 2's alpha is unused, :226)
        _lib.check(L.n3d_blend_planes_views(_lib.ptr(stitch), _lib.ptr(side), _lib.ptr(top), _lib.ptr(static), _lib.ptr(alpha),
                                            _lib.ptr(planes), N, 256, 256, alpha.shape[1], 0, 1, 3, _lib.stream()))
        if getattr(self, 'keep_stages', False):
            alpha = alpha.index_select(1, S.alpha_views)
        self._debug = dict(textures=textures, front=front, side=side, top=top, alpha=alpha, bbox=bbox, grid=grid, crop=crop, mouths=mouths,
                           stitch_in=stitch_in, stitch=stitch, static=static) if getattr(self, 'keep_stages', False) else None
        return planes, eg3d_ws

    def render(self, planes_cl, c, neural_rendering_resolution, depth_jitter=None, importance_u=None, density_noise_draws=None, _camera=None):
        """RaySampler + ImportanceRenderer on channels-last planes -> (feature_image [N,32,R,R], depth_image [N,1,R,R]).
        rendering_kwargs as the reference reads them (vr/renderer.py:95-147, vr/ray_marcher.py:27-66): fixed `ray_start` / `ray_end` or
        both 'auto' (per-ray box limits), `disparity_space_sampling`, `white_back`, `density_noise` (the normal draws come from the
        device RNG like the reference's torch.randn_like, or from `density_noise_draws = (coarse [N,R²,Sc], fine [N,R²,Sf])`);
        `clamp_mode` must be 'softplus' (the reference asserts the same, ray_marcher.py:37-40)."""
        S = self._prep()
        rk = self.rendering_kwargs
        dev, N, R = planes_cl.device, planes_cl.shape[0], int(neural_rendering_resolution)
        Sc, Sf = int(rk['depth_resolution']), int(rk['depth_resolution_importance'])
        t0, t1 = rk['ray_start'], rk['ray_end']
        if rk.get('clamp_mode', 'softplus') != 'softplus':
            raise RuntimeError("MipRayMarcher only supports `clamp_mode`=`softplus`!")                 # ray_marcher.py:40
        auto = t0 == 'auto' and t1 == 'auto'                                   # renderer.py:98: both, or the fixed branch
        if not auto and (isinstance(t0, str) or isinstance(t1, str)):
            raise RuntimeError("ray_start / ray_end: two numbers, or both 'auto'")
        disparity = bool(rk.get('disparity_space_sampling', False))
        if auto and disparity:
            raise RuntimeError("ray_start = ray_end = 'auto' with disparity_space_sampling: the reference's broadcast of per-ray bounds against "
                               '[N, M, S, 1] depths (renderer.py:193) does not run either')
        noise_amp = float(rk.get('density_noise', 0) or 0)
        opts = None
        split = RENDER_DECODER_SPLIT and layers.PRECISION == 'bf16x3'          # the decoder in the convolutions' arithmetic (fp32 route: float32-input MFMAs)
        if auto or disparity or rk.get('white_back', False) or noise_amp > 0 or split:
            opts = _lib.RenderOpts()
            opts.decoder_split_bf16 = 1 if split else 0
            opts.white_back = 1 if rk.get('white_back', False) else 0
            opts.disparity_space_sampling = 1 if disparity else 0
            opts.auto_bounds = 1 if auto else 0
            opts.box_side = float(rk['box_warp'])
            keep = []
            if auto:
                keep.append(torch.empty(N * R * R * 2, dtype=torch.float32, device=dev))
                opts.ray_bounds_ws = _lib.ptr(keep[-1])
            else:
                opts.ray_start, opts.ray_end = float(t0), float(t1)
            if noise_amp > 0:
                nc, nf = density_noise_draws if density_noise_draws is not None else (torch.randn(N, R * R, Sc, device=dev), torch.randn(N, R * R, max(Sf, 1), device=dev))
                keep += [nc.to(dev).contiguous(), nf.to(dev).contiguous()]
                opts.density_noise, opts.density_noise_coarse, opts.density_noise_fine = noise_amp, _lib.ptr(keep[-2]), _lib.ptr(keep[-1])
            opts._keep = keep
        # coarse sample positions: linspace(ray_start, ray_end) (fixed) or linspace(0, 1) (disparity); unused with 'auto'
        lo, hi = (0.0, 1.0) if (disparity or auto) else (float(t0), float(t1))
        key = (Sc, lo, hi)
        if key not in S.tlin:
            S.tlin[key] = torch.linspace(lo, hi, Sc).to(dev)
        t0, t1 = lo, hi
        if _camera is not None:              # (synthesis: already dense, n3d_unpack_inputs)
            cam2world, intrinsics = _camera
        else:
            c = c.to(device=dev, dtype=torch.float32)
            cam2world = c[:, :16].contiguous()
            intrinsics = c[:, 16:25].contiguous()
        jitter = (torch.rand((N, R * R, Sc, 1), device=dev) if depth_jitter is None else depth_jitter.to(dev)).contiguous()
        u = (torch.rand((N * R * R, Sf), device=dev) if importance_u is None else importance_u.to(dev)).contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        feat = torch.empty(N, 32, R, R, **f32)
        depth = torch.empty(N, 1, R, R, **f32)
        bounds = torch.empty(2, **f32)              # scratch of this call (calls may be in flight on several streams)
        _lib.check(_lib.lib().n3d_render_rays_ex(
            _lib.ptr(planes_cl), _lib.ptr(cam2world), _lib.ptr(intrinsics), _lib.ptr(S.tlin[key]), _lib.ptr(jitter),
            _lib.ptr(u), _lib.ptr(S.dec_w1), _lib.ptr(S.dec_b1), _lib.ptr(S.dec_w2), _lib.ptr(S.dec_b2), _lib.ptr(feat),
            _lib.ptr(depth), None, _lib.ptr(bounds), N, R, Sc, Sf, planes_cl.shape[2], planes_cl.shape[3],
            float((t1 - t0) / (Sc - 1)), float(2 / rk['box_warp']), opts, _lib.stream()))
        return feat, depth

    def synthesis(self, ws, c, v, neural_rendering_resolution=None, update_emas=False, cache_backbone=False,
                  use_cached_backbone=False, depth_jitter=None, importance_u=None, cache_identity=False,
                  use_cached_identity=False, **synthesis_kwargs):
        """reference triplane_next3d.py:117-188.  Extra keyword inputs `depth_jitter` [N,R²,Sc,1] / `importance_u`
        [N·R²,Sf] replace the device RNG draws of the renderer (tests feed the oracle's tensors).  `cache_backbone` /
        `use_cached_backbone` (reference signature; semantics of upstream training/triplane.py:67-72) keep the blended planes
        across calls (camera orbits); `cache_identity` / `use_cached_identity` keep only the latent-dependent networks
        (texture + static backbone) so that a new mesh per frame re-runs just raster -> mouth -> blending."""
        noise_mode = synthesis_kwargs.get('noise_mode', 'random')       # the reference's default (networks_stylegan2.py:311)
        if noise_mode not in ('random', 'const', 'none'):
            raise RuntimeError(f'noise_mode {noise_mode!r}: random / const / none')
        if neural_rendering_resolution is None:
            neural_rendering_resolution = self.neural_rendering_resolution
        else:
            self.neural_rendering_resolution = neural_rendering_resolution
        self._check_params()
        S = self._prep()
        ws = ws.to(device=self.device, dtype=torch.float32)
        eg3d_ws = ws[:, :S.texture.num_ws]             # always the CURRENT latents (upstream training/triplane.py:67-72 caches planes only)
        bank = None
        if S.one_bank and ws.shape[1] == 2 * S.texture.num_ws and ws.shape[2] == 512:
            if not (ws.stride(2) == 1 and ws.stride(1) == 512):
                ws = ws.contiguous()
            bank = S.all_bank.compute(ws)
        # the call's two small inputs -> the dense tensors the kernels read, in one launch (v: vertices + landmarks, c: camera label)
        vd, cd = v.to(device=self.device, dtype=torch.float32), c.to(device=self.device, dtype=torch.float32)
        unpacked = camera = None
        if vd.dim() == 3 and vd.shape[1] > 5023 and vd.shape[2] == 3 and vd.stride()[1:] == (3, 1) and cd.dim() == 2 and cd.shape[1] >= 25 and cd.stride(1) == 1 and cd.shape[0] == vd.shape[0]:
            verts, lms, cam, intr = self._unpack(vd, cd)
            unpacked, camera = (verts, lms), (cam, intr)
        if use_cached_backbone and self._last_planes is not None:
            planes = self._last_planes
        else:
            planes, _ = self._planes(ws, vd, noise_mode, cache_identity, use_cached_identity, bank=bank,
                                     force_fp32=bool(synthesis_kwargs.get('force_fp32', False)), _unpacked=unpacked)
        if cache_backbone:
            self._set_cache('_last_planes', planes)
        feature_image, depth_image = self.render(planes, cd, neural_rendering_resolution, depth_jitter, importance_u,
                                                 synthesis_kwargs.get('density_noise_draws'), _camera=camera)
        rgb_image = feature_image[:, :3]
        sr_noise = self.rendering_kwargs.get('superresolution_noise_mode', 'none')     # triplane_next3d.py:182
        # the reference's default: fp16 super-resolution blocks (no inference script passes force_fp32); `force_fp32=True` is
        # the float32 path its CPU run takes (networks_stylegan2.py:548) and the one the golden fixtures pin
        sr_fp16 = self.sr_use_fp16 and not synthesis_kwargs.get('force_fp32', False)
        # rendering_kwargs['sr_antialias'] (train_next3d.py:326: True in every next3d configuration) selects the antialiased resize of superresolution.py:282-286.
        # False: an UP-scaling is the same two taps and weights either way (differences of a float32 ulp); a DOWN-scaling (render above the module's input
        # resolution) would be plain bilinear, which this build has no kernel for — refused, not replaced silently
        if not self.rendering_kwargs.get('sr_antialias', True) and feature_image.shape[-1] > S.sr.input_resolution:
            raise RuntimeError(f"sr_antialias=False with a {feature_image.shape[-1]} x {feature_image.shape[-1]} render above the super-resolution input "
                               f'({S.sr.input_resolution}): the non-antialiased down-scaling is not implemented')
        sr_image = S.sr(rgb_image, feature_image, eg3d_ws, _resize_aa, noise_mode=sr_noise, fp16=sr_fp16, bank=bank)
        if getattr(S.sr, 'aliased_raw', None) is not None:       # SuperresolutionHybrid4X / 2X without a resize: the reference's in-place update of its own view (networks.SuperRes8XDC)
            rgb_image = S.sr.aliased_raw
        return {'image': sr_image, 'image_raw': rgb_image, 'image_depth': depth_image}

    # ------------------------------------------------------------------ HIP-graph replay of the steady-state loops
    def synthesis_graph(self, ws, c, v, **synthesis_kwargs):
        """`synthesis(ws, c, v, **kw)` through a captured HIP graph: the first call with a given signature — tensor shapes, render
        resolution, sample counts, cache flags, precision switches — runs the forward eagerly twice (prepared weights, job tables,
        caches), captures it once with static input / scratch / output buffers, and every call (the first included) copies
        (ws, c, v) [+ depth_jitter / importance_u when given] into the static inputs and REPLAYS the graph: one host call instead of
        ~155 ctypes launches per frame.  Meant for the loops that call synthesis with a fixed signature — the camera orbit of
        gen_videos_next3d.py:126-158 (`use_cached_backbone=True`), the reenactment loop of reenact_avatar_next3d.py:139-164
        (`use_cached_identity=True`), single-frame latency at batch 1.  The returned tensors are the graph's static outputs: consume
        (or clone) them before the next replay.  `cache_backbone` / `cache_identity` (WRITE flags) are refused — fill the caches with
        an eager call first; a parameter update drops the graphs (`_drop_derived`)."""
        if synthesis_kwargs.get('cache_backbone') or synthesis_kwargs.get('cache_identity'):
            raise RuntimeError('synthesis_graph: fill the caches with an eager synthesis(..., cache_backbone / cache_identity=True) call first')
        self._check_params()
        self._prep()
        rk = self.rendering_kwargs
        tensors = {k: synthesis_kwargs[k] for k in ('depth_jitter', 'importance_u') if synthesis_kwargs.get(k) is not None}
        draws = synthesis_kwargs.get('density_noise_draws')
        if draws is not None:             # (coarse, fine) normal draws: static copies like the other random inputs, never the caller's pointers
            tensors['density_noise_coarse'], tensors['density_noise_fine'] = draws
        plain = {k: val for k, val in synthesis_kwargs.items() if k not in tensors and k not in ('graph_slot', 'density_noise_draws')}
        # graph_slot: independent instances of the same signature (own static buffers): a serving loop replays slot k on stream k so
        # that several single-frame requests are in flight at once (bench.py config1b)
        slot = int(synthesis_kwargs.get('graph_slot', 0))
        sig = (tuple(ws.shape), tuple(c.shape), tuple(v.shape), tuple((k, tuple(t.shape)) for k, t in sorted(tensors.items())),
               tuple(sorted((k, repr(val)) for k, val in plain.items())), plain.get('neural_rendering_resolution') or self.neural_rendering_resolution,
               rk['depth_resolution'], rk['depth_resolution_importance'], rk.get('superresolution_noise_mode', 'none'),
               # everything render() bakes into the launch arguments (a change between calls must not replay the old graph)
               tuple(repr(rk.get(k)) for k in ('ray_start', 'ray_end', 'white_back', 'density_noise', 'disparity_space_sampling', 'box_warp',
                                                'clamp_mode', 'decoder_lr_mul', 'c_gen_conditioning_zero', 'c_scale')),
               # ... and every module switch that selects kernels
               layers.PRECISION, layers.PRESPLIT, layers.S2_PRESPLIT, layers.UP_PRESPLIT, layers.TORGB_SIDE, layers.FUSED_TORGB, layers.FUSED_TORGB_MAX, layers.FUSED_TORGB_MID,
               layers.SK_S2, layers.SK_S2_MAX_IN, layers.DIRECT_SPLIT8,
               layers.UP_PS_NCHW, layers.NCHW_FIR_SPLIT8, layers.CONVERT_MAX_BYTES, layers.F16_REF_CPU_ROUNDING, layers.uf.FIR_SEP, RASTER_ON_SIDE_STREAM,
               self.overlap_static, slot)
        uses_cache = bool((plain.get('use_cached_backbone') and self._last_planes is not None) or
                          (plain.get('use_cached_identity') and self._identity_cache is not None))
        if uses_cache:
            sig = sig + (self._cache_gen,)
        if self._graphs is None:
            self._graphs = {}
        entry = self._graphs.get(sig)
        if entry is None:
            dev = self.device
            _require_hip(dev)
            st = dict(ws=ws.to(dev, torch.float32).clone(), c=c.to(dev, torch.float32).clone(), v=v.to(dev, torch.float32).clone())
            st.update({k: t.to(dev).clone() for k, t in tensors.items()})
            def call():
                kw = {k: st[k] for k in tensors if not k.startswith('density_noise_')}
                if 'density_noise_coarse' in tensors:
                    kw['density_noise_draws'] = (st['density_noise_coarse'], st['density_noise_fine'])
                return self.synthesis(st['ws'], st['c'], st['v'], **plain, **kw)
            cur = torch.cuda.current_stream()
            warm = torch.cuda.Stream(device=dev)
            warm.wait_stream(cur)
            with torch.cuda.stream(warm):                  # eager warm-up off the default stream, as graph capture wants it
                call(); call()
            cur.wait_stream(warm)
            torch.cuda.synchronize(dev)
            graph = torch.cuda.CUDAGraph()
            # the graph's own arrival counters (the few-pixel layers' in-launch split-K reduce, _lib.tickets): allocated and zeroed HERE, eagerly —
            # a replay may run beside eager launches of the streams it was captured on, so it must not share their pools
            pools = _lib.ticket_pools(_lib.new_ticket_pools(dev))
            with torch.cuda.graph(graph), pools:
                out = call()
            st['_ticket_pools'] = pools.all
            # the caches the graph reads stay referenced by its entry: they outlive the graph whatever the caller does next
            entry = self._graphs[sig] = (graph, st, out, uses_cache, (self._last_planes, self._identity_cache) if uses_cache else None)
        graph, st, out = entry[:3]
        st['ws'].copy_(ws); st['c'].copy_(c); st['v'].copy_(v)
        for k, t in tensors.items():
            st[k].copy_(t)
        graph.replay()
        return out

    def sample_mixed(self, coordinates, directions, ws, v, truncation_psi=1, truncation_cutoff=None, update_emas=False,
                     cache_backbone=False, use_cached_backbone=False, **synthesis_kwargs):
        """reference triplane_next3d.py:278-322: RGB features and density at arbitrary 3-D points (shape extraction,
        gen_samples_next3d.py:208-246) -> {'rgb' [N,M,32], 'sigma' [N,M,1]}.  `directions` is accepted and ignored exactly as
        OSGDecoder ignores it (:359).  The reference rebuilds all planes for every chunk of points; `cache_backbone` /
        `use_cached_backbone` (same flags as `synthesis`) keep them across calls."""
        noise_mode = synthesis_kwargs.get('noise_mode', 'random')       # the reference's default (networks_stylegan2.py:311): per-sample noise draws, as in `synthesis`
        if noise_mode not in ('random', 'const', 'none'):
            raise RuntimeError(f'noise_mode {noise_mode!r}: random / const / none')
        self._check_params()
        S = self._prep()
        ws = ws.to(device=self.device, dtype=torch.float32)
        if use_cached_backbone and self._last_planes is not None:
            planes = self._last_planes
        else:
            planes, _ = self._planes(ws, v, noise_mode, force_fp32=bool(synthesis_kwargs.get('force_fp32', False)))
            if cache_backbone:
                self._set_cache('_last_planes', planes)
        coords = coordinates.to(device=self.device, dtype=torch.float32).contiguous()
        N, M = coords.shape[0], coords.shape[1]
        if N != planes.shape[0] or coords.shape[2] != 3:
            raise RuntimeError(f'sample: coordinates must be [N={planes.shape[0]}, M, 3], got {tuple(coords.shape)}')
        rgb = torch.empty(N, M, 32, dtype=torch.float32, device=self.device)
        sigma = torch.empty(N, M, 1, dtype=torch.float32, device=self.device)
        _lib.check(_lib.lib().n3d_sample_points(_lib.ptr(planes), _lib.ptr(coords), _lib.ptr(S.dec_w1), _lib.ptr(S.dec_b1), _lib.ptr(S.dec_w2),
                                                _lib.ptr(S.dec_b2), _lib.ptr(rgb), _lib.ptr(sigma), N, M, planes.shape[2], planes.shape[3],
                                                2.0 / float(self.rendering_kwargs['box_warp']), _lib.stream()))
        return {'rgb': rgb, 'sigma': sigma}

    def sample(self, coordinates, directions, z, c, v, truncation_psi=1, truncation_cutoff=None, update_emas=False, **synthesis_kwargs):
        """reference triplane_next3d.py:232-276: `mapping` + `sample_mixed`."""
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff, update_emas=update_emas)
        return self.sample_mixed(coordinates, directions, ws, v, update_emas=update_emas, **synthesis_kwargs)

    def forward(self, z, c, v, truncation_psi=1, truncation_cutoff=None, neural_rendering_resolution=None, update_emas=False,
                cache_backbone=False, use_cached_backbone=False, **synthesis_kwargs):
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff, update_emas=update_emas)
        return self.synthesis(ws, c, v, update_emas=update_emas, neural_rendering_resolution=neural_rendering_resolution,
                              cache_backbone=cache_backbone, use_cached_backbone=use_cached_backbone, **synthesis_kwargs)


class _DecoderFC(torch.nn.Module):
    """weight [O,I] / bias [O] with the reference's equalised-learning-rate scaling (FullyConnectedLayer, tat/networks_stylegan2.py:95-127)."""

    def __init__(self, in_features, out_features, lr_multiplier=1.0):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.randn(out_features, in_features) / lr_multiplier)
        self.bias = torch.nn.Parameter(torch.zeros(out_features))
        self.weight_gain, self.bias_gain = lr_multiplier / np.sqrt(in_features), lr_multiplier

    def forward(self, x):
        return torch.addmm((self.bias * self.bias_gain).unsqueeze(0), x, (self.weight * self.weight_gain).t())


class OSGDecoder(torch.nn.Module):
    """The un-pickling target `training_avatar_texture.triplane_next3d.OSGDecoder`.  The reference's decoder is NOT a persistent class
    (tat/triplane_next3d.py:348-371 has no @persistence.persistent_class), so a network pickle stores `G.decoder` BY REFERENCE to that module path —
    the path `install_dropin(model=True)` aliases to this module: without this class `legacy.load_network_pkl` fails before `--reload_modules`
    is even looked at (found by tests/_e2e_scripts.py, round 5).  An un-pickled instance carries the pickle's `net` (the reference's own
    FullyConnectedLayer modules, which ARE persistent); with `--reload_modules=True` it only donates its parameters to
    misc.copy_params_and_buffers, without it (boundary B1: the pickled reference generator keeps running) `forward` is what that generator calls.
    This package's own generator never instantiates it: its decoder runs inside n3d_render_rays."""

    def __init__(self, n_features, options):
        super().__init__()
        self.hidden_dim = 64
        lr = options['decoder_lr_mul']
        self.net = torch.nn.Sequential(_DecoderFC(n_features, self.hidden_dim, lr), torch.nn.Softplus(),
                                       _DecoderFC(self.hidden_dim, 1 + options['decoder_output_dim'], lr))

    def forward(self, sampled_features, ray_directions, sampled_embeddings=None):
        feats = sampled_features.mean(dim=1)                              # [N, planes, M, C] -> [N, M, C]: the three planes' features averaged
        batch, points = feats.shape[:2]
        raw = self.net(feats.reshape(batch * points, -1)).reshape(batch, points, -1)
        return {'rgb': torch.sigmoid(raw[..., 1:]) * 1.002 - 0.001,         # MipNeRF's widened sigmoid (:369)
                'sigma': raw[..., :1]}


def _resize_aa(x, size):
    """F.interpolate(x, (size,size), mode='bilinear', align_corners=False, antialias=True) on libn3d.so."""
    n, c, h, w = x.shape
    y = torch.empty(n, c, size, size, dtype=torch.float32, device=x.device)
    if x.stride()[1:] != (h * w, w, 1) or x.stride(0) < c * h * w:        # dense planes with any batch stride (a channel-slice view) need no copy
        x = x.contiguous()
    _lib.check(_lib.lib().n3d_resize_aa_strided(_lib.ptr(x), x.stride(0), _lib.ptr(y), None, None, n, c, h, w, size, size, 0, _lib.stream()))
    return y
