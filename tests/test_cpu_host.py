"""CPU (-m "not gpu") tests: parameter inventory, C-ABI surface, host logic, loud failure without a device."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(REPO, 'tests', 'golden')


@pytest.fixture(scope='module')
def built_lib():
    from next3d_amd import build
    return build.build(verbose=False)


def test_spec_matches_reference_state_dict():
    """next3d_amd.spec reproduces every name/shape the reference constructors register (fixture written by
    oracle/pin_against_reference.py from the real reference)."""
    from next3d_amd import spec
    ref = {}
    for line in open(os.path.join(GOLDEN, 'ref_state_dict_spec.txt')):
        name, rest = line.split(' ', 1)
        shape = tuple(int(v) for v in re.findall(r'\d+', rest.rsplit(')', 1)[0]))
        ref[name] = shape
    mine = {k: tuple(v[0]) for k, v in spec.build_spec().items()}
    assert set(mine) == set(ref), (sorted(set(mine) ^ set(ref))[:10])
    bad = {k: (mine[k], ref[k]) for k in ref if mine[k] != ref[k]}
    assert not bad, bad
    assert len(ref) == 674


def test_generator_module_state_dict_names():
    from next3d_amd import demo, spec
    d = demo.demo_arrays()
    from next3d_amd.generator import TriPlaneGenerator
    from next3d_amd import mesh
    topo = (d['faces'], d['uvs'], d['uvfaces'])
    with pytest.raises(RuntimeError, match='uv_face_eye_mask.png not found'):     # as the reference (cv2.imread -> None -> crash): no silent stand-in
        TriPlaneGenerator(512, 25, 512, 512, 3, topo, rendering_kwargs=dict(demo.RENDERING_KWARGS))
    with pytest.raises(RuntimeError, match='superresolution_module'):            # (the reference cannot construct this one either: sr_antialias -> SynthesisLayer TypeError)
        TriPlaneGenerator(512, 25, 512, 256, 3, topo, uv_face_mask=mesh.synthetic_uv_face_mask(),
                          rendering_kwargs=dict(demo.RENDERING_KWARGS, superresolution_module='training_avatar_texture.superresolution.SuperresolutionHybridDeepfp32'))
    with pytest.raises(RuntimeError, match='256 x 256'):                          # every module has its own output resolution (the reference's modules assert it)
        TriPlaneGenerator(512, 25, 512, 512, 3, topo, uv_face_mask=mesh.synthetic_uv_face_mask(),
                          rendering_kwargs=dict(demo.RENDERING_KWARGS, superresolution_module='training_avatar_texture.superresolution.SuperresolutionHybrid4X'))
    for cls, res in (('SuperresolutionHybrid8X', 512), ('SuperresolutionHybrid4X', 256), ('SuperresolutionHybrid2X', 128)):      # round 5: the other modules
        g = TriPlaneGenerator(512, 25, 512, res, 3, topo, uv_face_mask=mesh.synthetic_uv_face_mask(), mapping_kwargs=dict(num_layers=2),
                              rendering_kwargs=dict(demo.RENDERING_KWARGS, superresolution_module='training_avatar_texture.superresolution.' + cls))
        assert set(g.state_dict()) == set(spec.build_spec(cls)) and 'superresolution.resample_filter' in g.state_dict()
        assert tuple(g.state_dict()['superresolution.block1.conv1.weight'].shape) == (64, 64, 3, 3)
    # mapping depth: without mapping_kwargs.num_layers the reference's MappingNetwork builds its own default of 8 layers (tat/networks_stylegan2.py:207;
    # pinned: oracle/pin_against_reference.py --mapping-depth); train_next3d.py passes map_depth = 2
    G8 = TriPlaneGenerator(512, 25, 512, 512, 3, topo, rendering_kwargs=dict(demo.RENDERING_KWARGS), uv_face_mask=mesh.synthetic_uv_face_mask())
    assert set(G8.state_dict()) == set(spec.build_spec(mapping_layers=8)) and 'neural_blending.mapping.fc7.weight' in G8.state_dict() and G8.backbone.mapping.num_layers == 8
    with pytest.raises(RuntimeError, match='num_layers'):
        TriPlaneGenerator(512, 25, 512, 512, 3, topo, rendering_kwargs=dict(demo.RENDERING_KWARGS), uv_face_mask=mesh.synthetic_uv_face_mask(), mapping_kwargs=dict(num_layers=0))
    # the other MappingNetwork options: their reference defaults pass, another value is refused (mapping() hard-codes it), an unknown key raises
    TriPlaneGenerator(512, 25, 512, 512, 3, topo, rendering_kwargs=dict(demo.RENDERING_KWARGS), uv_face_mask=mesh.synthetic_uv_face_mask(),
                      mapping_kwargs=dict(num_layers=2, lr_multiplier=0.01, activation='lrelu', w_avg_beta=0.998, embed_features=None), kernel_size=3)
    for bad in (dict(lr_multiplier=0.1), dict(activation='relu'), dict(layer_features=256)):
        with pytest.raises(RuntimeError, match='implements the reference default'):
            TriPlaneGenerator(512, 25, 512, 512, 3, topo, rendering_kwargs=dict(demo.RENDERING_KWARGS), uv_face_mask=mesh.synthetic_uv_face_mask(), mapping_kwargs=dict(num_layers=2, **bad))
    with pytest.raises(TypeError, match='unexpected mapping'):
        TriPlaneGenerator(512, 25, 512, 512, 3, topo, rendering_kwargs=dict(demo.RENDERING_KWARGS), uv_face_mask=mesh.synthetic_uv_face_mask(), mapping_kwargs=dict(depth=2))
    G = TriPlaneGenerator(512, 25, 512, 512, 3, topo, rendering_kwargs=dict(demo.RENDERING_KWARGS), uv_face_mask=mesh.synthetic_uv_face_mask(), mapping_kwargs=dict(num_layers=2))
    assert G.sr_conv_clamp is None                                           # sr_num_fp16_res == 0 -> no clamp (superresolution.py:273)
    # absent synthesis kwargs take the reference classes' defaults: float16 in the 4 highest resolutions of the backbones (fp16_resolution 32), conv_clamp 256
    assert G.backbone_fp16_resolution == 32 and G.backbone_conv_clamp == 256
    G0 = TriPlaneGenerator(512, 25, 512, 512, 3, topo, rendering_kwargs=dict(demo.RENDERING_KWARGS), uv_face_mask=mesh.synthetic_uv_face_mask(), num_fp16_res=0, conv_clamp=None)
    for bad in (dict(architecture='resnet'), dict(use_noise=False), dict(resample_filter=[1, 2, 1])):               # block options other than next3d's: refused, not ignored
        with pytest.raises(RuntimeError, match='implements'):
            TriPlaneGenerator(512, 25, 512, 512, 3, topo, rendering_kwargs=dict(demo.RENDERING_KWARGS), uv_face_mask=mesh.synthetic_uv_face_mask(), **bad)
    with pytest.raises(TypeError, match='unexpected'):                                                              # (as the reference's SynthesisBlock would)
        TriPlaneGenerator(512, 25, 512, 512, 3, topo, rendering_kwargs=dict(demo.RENDERING_KWARGS), uv_face_mask=mesh.synthetic_uv_face_mask(), magnitude_ema_beta=0.9)
    assert G0.backbone_fp16_resolution is None and G0.backbone_conv_clamp is None
    assert set(G.state_dict()) == set(spec.build_spec())
    params = {n for n, _ in G.named_parameters()}
    assert 'backbone.synthesis.b4.conv1.noise_const' not in params and 'backbone.synthesis.b4.conv1.weight' in params
    assert sum(p.numel() for p in G.parameters()) == 172_815_807
    with pytest.raises(RuntimeError):                                      # no CPU fallback
        G.mapping(torch.zeros(1, 512), torch.zeros(1, 25))


def test_abi_header_symbols_exported(built_lib):
    """Every function include/n3d.h declares resolves in libn3d.so and is bound by the ctypes layer."""
    from next3d_amd import _lib
    hdr = open(os.path.join(REPO, 'include', 'n3d.h')).read()
    declared = sorted(set(re.findall(r'\b(n3d_[a-z0-9_]+)\s*\(', hdr)))
    assert declared == _lib.exported_symbols(), set(declared) ^ set(_lib.exported_symbols())
    h = ctypes.CDLL(built_lib)
    for name in declared:
        assert hasattr(h, name), name
    assert h.n3d_abi_version() == _lib.ABI_VERSION
    nm = subprocess.run(['nm', '-D', '--defined-only', built_lib], capture_output=True, text=True).stdout
    exported = set(re.findall(r' T (n3d_[a-z0-9_]+)', nm))
    assert set(declared) <= exported
    lib = _lib.lib()
    assert lib.n3d_last_error() is not None


def test_ops_fail_loudly_without_device(built_lib, monkeypatch):
    from next3d_amd import _lib
    from next3d_amd.torch_utils.ops import bias_act, conv2d_resample, upfirdn2d
    x = torch.zeros(1, 4, 8, 8)
    for fn in (lambda: bias_act.bias_act(x), lambda: upfirdn2d.upfirdn2d(x, None),
               lambda: conv2d_resample.conv2d_resample(x, torch.zeros(4, 4, 3, 3), padding=1),
               lambda: bias_act.bias_act(x, impl='ref')):
        with pytest.raises(RuntimeError):
            fn()
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', '/nonexistent/libn3d.so')
    with pytest.raises(RuntimeError, match='no CPU or PyTorch fallback'):
        _lib.lib()


def test_filters_padding_and_ksplit():
    from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
    from next3d_amd.torch_utils.ops import upfirdn2d as uf
    from oracle import ops as O
    assert torch.equal(uf.setup_filter([1, 3, 3, 1]), O.setup_filter((1, 3, 3, 1)))
    assert uf.setup_filter([1] * 8).ndim == 1 and uf.setup_filter(None).shape == (1, 1)
    assert uf._parse_padding(2) == (2, 2, 2, 2) and uf._parse_padding([1, 2]) == (1, 1, 2, 2)
    assert cg.out_shape(16, 16, 2) == (33, 33) and cg.out_shape(33, 33, 1) == (16, 16)
    assert cg.pick_ksplit(4, 512, 512, 4, 4, 3) == 16 and cg.pick_ksplit(4, 128, 128, 256, 256, 3) == 1


def test_few_pixel_kernel_eligibility_plan(built_lib):
    """n3d_conv2d_sk_eligible (host-only): which stride-1 3x3 layers run on the one-launch few-pixel kernel — the generator's 4x4 .. 16x16
    layers at batch 1 / 4 / 8 and 32x32 at batch 1 and 2; not the layers with more than 2048 output pixels, the deep fusion layer at 16x16
    (8 chunks per wave), channel counts the tile plan does not take, or image shapes whose rows do not tile 32 / 64 pixels."""
    from next3d_amd import _lib
    e = lambda *a: _lib.lib().n3d_conv2d_sk_eligible(*a)
    for n in (1, 2, 4, 8):
        for hw in (4, 8, 16):
            assert e(n, 512, 512, hw, hw) == 1, (n, hw)
    assert e(1, 512, 512, 32, 32) == 1 and e(2, 512, 512, 32, 32) == 1 and e(4, 512, 512, 32, 32) == 0     # <= 2048 pixels in the batch
    assert e(4, 1024, 512, 8, 8) == 1 and e(4, 1024, 512, 16, 16) == 0 and e(1, 1024, 512, 16, 16) == 1     # I > 512: <= 256 pixels
    assert e(4, 64, 64, 4, 4) == 0 and e(4, 2048, 512, 4, 4) == 0 and e(4, 512, 70, 8, 8) == 0              # I % 128, I <= 1024, O % 32
    assert e(4, 512, 512, 64, 64) == 0 and e(1, 128, 32, 6, 6) == 0 and e(1, 128, 32, 5, 40) == 0          # large images; rows that do not tile
    assert e(3, 128, 64, 4, 4) == 1 and e(2, 128, 64, 8, 16) == 1 and e(5, 128, 32, 4, 8) == 1             # odd batches, non-square


def test_camera_and_mesh_helpers():
    from next3d_amd import demo, mesh
    camera_utils = demo
    g = np.load(os.path.join(GOLDEN, 'case_r32_s24.npz'))
    c, c_cond = camera_utils.demo_camera_params(angle_y=0.4)
    assert np.abs(c.numpy() - g['c']).max() <= 1e-6 and np.abs(c_cond.numpy() - g['c_cond']).max() <= 1e-6
    d = demo.demo_arrays()
    mb = mesh.mesh_buffers(d['faces'], d['uvs'], d['uvfaces'])
    assert mb['face_uvcoords'].shape == (1, 9976, 3, 3) and mb['dense_faces'].shape == (1, 122990, 3)
    assert float(mb['uvcoords'][..., 2].min()) == 1.0
    z, cc, _, v = demo.demo_batch([0], yaws=[0.4])
    assert np.abs(z.numpy() - g['z']).max() == 0 and np.abs(v.numpy() - g['v']).max() == 0
    m = mesh.synthetic_uv_face_mask()
    assert m.shape == (1, 1, 256, 256) and float(m.min()) == 0.0 and float(m.max()) == 1.0


def test_obj_and_landmark_parsers(tmp_path):
    from next3d_amd import mesh
    p = tmp_path / 't.obj'
    p.write_text('mtllib x\nv 0 0 0\nv 1 0 0.5\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nf 1/1 2/2 3/3\n')
    v, f, vt, ft = mesh.parse_obj(str(p))
    assert v.shape == (3, 3) and f.tolist() == [[0, 1, 2]] and ft.tolist() == [[0, 1, 2]] and vt.shape == (3, 2)
    assert mesh.parse_obj_vertices(str(p)).shape == (1, 3, 3)
    q = tmp_path / 'k.txt'
    q.write_text('\n'.join('0.1 0.2 0.3' for _ in range(68)))
    assert mesh.parse_landmarks(str(q)).shape == (1, 68, 3)


def test_packed_mesh_sequence_roundtrip(tmp_path):
    """next3d_amd.meshio: .obj/_kpt2d.txt frames -> packed file -> batches identical to the per-frame text parse."""
    from next3d_amd import mesh, meshio
    d = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'demo_inputs.npz'))
    objs, kpts = [], []
    for t in range(5):
        o, k = tmp_path / f'f{t}.obj', tmp_path / f'f{t}_kpt2d.txt'
        with open(o, 'w') as fh:
            fh.write('# test\n')
            for x, y, z in d['verts'] + 0.001 * t:
                fh.write(f'v {x:.8f} {y:.8f} {z:.8f}\n')
            fh.write('f 1 2 3\n')
        np.savetxt(k, d['landmarks'] - 0.002 * t)
        objs.append(str(o)); kpts.append(str(k))
    out = str(tmp_path / 'seq.n3dmesh')
    arr = meshio.pack_sequence(objs, kpts, out)
    seq = meshio.MeshSequence(out)
    assert len(seq) == 5 and (seq.V, seq.L) == (5023, 68) and arr.shape == (5, 5091, 3)
    ref = [torch.cat([mesh.parse_obj_vertices(o), mesh.parse_landmarks(k)], 1)[0] for o, k in zip(objs, kpts)]
    got = list(seq.batches(2, 'cpu'))
    assert [g.shape[0] for g in got] == [2, 2, 1]
    assert torch.equal(torch.cat(got, 0), torch.stack(ref, 0))
    assert [g.shape[0] for g in seq.batches(2, 'cpu', drop_last=True)] == [2, 2]
    with open(out, 'r+b') as fh:
        fh.write(b'XXXXXXXX')
    with pytest.raises(ValueError):
        meshio.MeshSequence(out)


def test_abi_struct_layouts_match_ctypes(tmp_path):
    """n3d_epilogue / n3d_conv2d_desc / n3d_fc_job as gcc lays them out from include/n3d.h == the ctypes mirrors in
    next3d_amd/_lib.py (size and every field offset): a silent mismatch would hand the kernels garbage pointers."""
    import ctypes as C
    from next3d_amd import _lib
    structs = {'n3d_epilogue': _lib.Epilogue, 'n3d_conv2d_desc': _lib.Conv2dDesc, 'n3d_fc_job': _lib.FcJob}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(REPO, "include", "n3d.h")}"', 'int main(void) {']
    for cname, cls in structs.items():
        lines.append(f'  printf("{cname} size %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname} {fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src, exe = tmp_path / 'layout.c', tmp_path / 'layout'
    src.write_text('\n'.join(lines))
    subprocess.run(['gcc', '-std=c11', str(src), '-o', str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    got = {tuple(l.split()[:2]): int(l.split()[2]) for l in out.strip().splitlines()}
    for cname, cls in structs.items():
        assert got[(cname, 'size')] == C.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert got[(cname, fname)] == getattr(cls, fname).offset, (cname, fname)


def test_operator_api_matches_reference_signatures():
    """B1: every function of the reference's operator layer exists here with the SAME argument list (names, order,
    defaults) — `tests/golden/ref_ops_api.txt` is read off the reference's source by oracle/dump_ref_api.py.  Extra arguments
    are allowed only after the reference's and only as private (`_name`) keywords."""
    import ast
    ops_dir = os.path.join(REPO, 'next3d_amd', 'torch_utils', 'ops')
    checked = 0
    for line in open(os.path.join(GOLDEN, 'ref_ops_api.txt')):
        line = line.strip()
        if not line:
            continue
        qual, ref_args = line.split('(', 1)
        mod, name = qual.split('.')
        ref_args = ref_args[:-1]
        if mod == 'TriPlaneGenerator':        # B2: same leading arguments; extra keyword inputs only between them and **synthesis_kwargs
            tree = ast.parse(open(os.path.join(REPO, 'next3d_amd', 'generator.py')).read())
            cls = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == 'TriPlaneGenerator'][0]
            ours = {m.name: ast.unparse(m.args) for m in cls.body if isinstance(m, ast.FunctionDef)}[name]
            if not ref_args.endswith(', **synthesis_kwargs'):
                assert ours == ref_args, (name, ours)
                checked += 1
                continue
            lead = ref_args[:-len(', **synthesis_kwargs')]
            assert ours.startswith(lead) and ours.endswith('**synthesis_kwargs'), (name, ours)
            extra = ours[len(lead):-len('**synthesis_kwargs')].strip(', ')
            assert all('=' in a for a in extra.split(', ') if a), f'{name}: extra positional argument in ({ours})'
            checked += 1
            continue
        tree = ast.parse(open(os.path.join(ops_dir, mod + '.py')).read())
        fns = {n.name: n for n in tree.body if isinstance(n, ast.FunctionDef)}
        assert name in fns, f'{qual} is missing'
        ours = ast.unparse(fns[name].args)
        assert ours == ref_args or ours.startswith(ref_args + ', _'), f'{qual}: ({ours}) != reference ({ref_args})'
        checked += 1
    assert checked >= 23
    from next3d_amd.torch_utils.ops import bias_act
    # activation table: names, default alpha / gain and the plugin's activation index (bias_act.py:23-33) — read by the
    # reference's layer constructors (networks_stylegan2.py:159,301)
    want = {'linear': (0, 1, 1), 'relu': (0, np.sqrt(2), 2), 'lrelu': (0.2, np.sqrt(2), 3), 'tanh': (0, 1, 4), 'sigmoid': (0, 1, 5),
            'elu': (0, 1, 6), 'selu': (0, 1, 7), 'softplus': (0, 1, 8), 'swish': (0, np.sqrt(2), 9)}
    assert sorted(bias_act.activation_funcs) == sorted(want)
    for k, (a, g, idx) in want.items():
        spec_ = bias_act.activation_funcs[k]
        assert (spec_.def_alpha, float(spec_.def_gain), spec_.cuda_idx) == (a, float(g), idx), k


def test_install_dropin_aliases_reference_module_paths():
    """install_dropin(): `torch_utils.ops.*` (and with model=True the generator module) resolve to this package — what makes
    un-pickled reference network code call libn3d.so (torch_utils/persistence.py:218 resolves imports at load time)."""
    import subprocess
    code = ("import sys, next3d_amd; next3d_amd.install_dropin(model=True);"
            "import torch_utils.ops.bias_act as b, torch_utils.ops.upfirdn2d as u, torch_utils.ops.conv2d_resample as c;"
            "from torch_utils.ops import fma, filtered_lrelu, conv2d_gradfix;"
            "import training_avatar_texture.triplane_next3d as t;"
            "assert all(m.__name__.startswith('next3d_amd.') for m in (b, u, c, fma, filtered_lrelu, conv2d_gradfix, t)), "
            "[m.__name__ for m in (b, u, c, t)];"
            "assert hasattr(t, 'TriPlaneGenerator') and callable(u.setup_filter) and callable(c.conv2d_resample); print('ok')")
    r = subprocess.run([sys.executable, '-c', code], cwd=REPO, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith('ok'), r.stdout + r.stderr


@pytest.mark.skipif(not os.path.isdir('/root/reference/training_avatar_texture'), reason='needs the reference tree (build container only)')
def test_reference_network_code_binds_to_this_operator_layer():
    """With install_dropin() the REFERENCE's own layer classes (training_avatar_texture/networks_stylegan2.py) import this
    package's ops: the constructors read `bias_act.activation_funcs` / `upfirdn2d.setup_filter` from it, and a forward on CPU
    tensors ends in this package's loud "HIP device only" error instead of the reference's `_ref` fallback — i.e. reference
    network code un-pickled after install_dropin() runs on libn3d.so and nothing else."""
    code = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, '/root/reference')
import torch
from next3d_amd import mesh
from oracle import ref_shims
ref_shims.install(mesh.synthetic_uv_face_mask()[0, 0].numpy())
import next3d_amd
next3d_amd.install_dropin()
from training_avatar_texture import networks_stylegan2 as ns
assert ns.bias_act.__name__ == 'next3d_amd.torch_utils.ops.bias_act' and ns.conv2d_resample.__name__ == 'next3d_amd.torch_utils.ops.conv2d_resample'
layer = ns.SynthesisLayer(16, 16, w_dim=32, resolution=8, up=2)
rgb = ns.ToRGBLayer(16, 3, w_dim=32)
fc = ns.FullyConnectedLayer(32, 16, activation='lrelu')
assert tuple(layer.resample_filter.shape) == (4, 4) and abs(float(layer.resample_filter.sum()) - 1) < 1e-6
x, w = torch.randn(1, 16, 4, 4), torch.randn(1, 32)
for fn in (lambda: layer(x, w, noise_mode='const'), lambda: rgb(x, w), lambda: fc(w)):
    try:
        fn()
    except RuntimeError as e:
        assert 'HIP device' in str(e), e
    else:
        raise SystemExit('the reference layer ran on CPU: it did not go through libn3d.so')
print('ok')
""" % REPO
    r = subprocess.run([sys.executable, '-c', code], cwd='/tmp', capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith('ok'), r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.skipif(not os.path.isdir('/root/reference/training_avatar_texture'), reason='needs the reference tree (build container only)')
def test_reload_modules_path_against_the_real_reference():
    """B2, live: what `gen_samples_next3d.py --reload_modules=True` does (:151-157) — construct THIS TriPlaneGenerator
    from the reference generator's init_args / init_kwargs and `misc.copy_params_and_buffers(G, G_new, require_all=True)` — with
    the reference's own `misc` and a reference-built G; then every tensor must be equal and the attributes the scripts / viz read
    must agree."""
    code = r"""
import sys
sys.path.insert(0, %r); sys.path.insert(0, '/root/reference')
import torch
from next3d_amd import mesh
from oracle import ref_shims, pin_against_reference as pin
ref_shims.install(mesh.synthetic_uv_face_mask()[0, 0].numpy())
G = ref_shims.build_reference_generator(pin.RENDERING_KWARGS)
from torch_utils import misc                                            # the reference's
from next3d_amd.generator import TriPlaneGenerator
G_new = TriPlaneGenerator(*G.init_args, uv_face_mask=mesh.synthetic_uv_face_mask(), **G.init_kwargs).eval().requires_grad_(False)
misc.copy_params_and_buffers(G, G_new, require_all=True)
ref = dict(misc.named_params_and_buffers(G)); new = dict(misc.named_params_and_buffers(G_new))
assert set(ref) == set(new), (sorted(set(ref) - set(new))[:5], sorted(set(new) - set(ref))[:5])
for k, v in ref.items():
    assert v.shape == new[k].shape and v.dtype == new[k].dtype and torch.equal(v.detach(), new[k].detach()), k
for path in ('z_dim', 'c_dim', 'w_dim', 'img_resolution', 'img_channels', 'neural_rendering_resolution', 'backbone.num_ws',
             'backbone.mapping.num_ws', 'texture_backbone.num_ws', 'backbone.img_channels', 'texture_backbone.img_channels',
             'superresolution.input_resolution'):
    a, b = G, G_new
    for part in path.split('.'):
        a, b = getattr(a, part), getattr(b, part)
    assert a == b, (path, a, b)
assert G_new.rendering_kwargs == G.rendering_kwargs and torch.equal(G.backbone.mapping.w_avg, G_new.backbone.mapping.w_avg)
print('ok', len(ref))
""" % REPO
    r = subprocess.run([sys.executable, '-c', code], cwd='/tmp', capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().startswith('ok'), r.stdout[-2000:] + r.stderr[-4000:]


def test_layout_grid_tiling():
    """frames.layout_grid == the helper of gen_videos_next3d.py:35-49 (tiling part; the float -> uint8 conversion is a GPU kernel):
    frame b lands at grid row b // grid_w, column b % grid_w."""
    from next3d_amd import frames
    g = torch.Generator().manual_seed(0)
    img = torch.randint(0, 256, (6, 3, 4, 5), generator=g, dtype=torch.uint8)
    for grid_w, grid_h in ((3, 2), (2, 3), (6, 1), (None, 2)):
        out = frames.layout_grid(img, grid_w=grid_w, grid_h=grid_h, float_to_uint8=False)
        gw = grid_w if grid_w is not None else 6 // grid_h
        assert out.shape == (grid_h * 4, gw * 5, 3) and out.dtype == np.uint8
        for b in range(6):
            gy, gx = b // gw, b % gw
            assert np.array_equal(out[gy * 4:(gy + 1) * 4, gx * 5:(gx + 1) * 5], img[b].permute(1, 2, 0).numpy())
    chw = frames.layout_grid(img, grid_w=3, grid_h=2, float_to_uint8=False, chw_to_hwc=False, to_numpy=False)
    assert tuple(chw.shape) == (3, 8, 15) and torch.equal(chw[:, 4:8, 5:10], img[4])
    with pytest.raises(AssertionError):
        frames.layout_grid(img, grid_w=4, grid_h=2, float_to_uint8=False)
    with pytest.raises(RuntimeError):
        frames.layout_grid(img.float(), grid_w=3, grid_h=2)          # CPU tensor: no fallback for the conversion


def test_parameter_update_check_uses_a_cached_tensor_list():
    """TriPlaneGenerator._check_params (called by every mapping / synthesis): an in-place update of any parameter or buffer drops the prepared
    weights and caches — detected through version counters of a CACHED tensor list (walking the 674-entry module tree cost 0.4 ms per call, which
    bound the eager batch-1 call: docs/history/DESIGN_rounds1-4.md 3.1h); a parameter OBJECT replaced by assignment / a re-registered buffer is caught on the NEXT call
    (generator._Tracked bumps a structure counter: ADVICE r4), and the tree is re-walked every 256th call for assignments that bypass the module API."""
    import torch
    from next3d_amd import demo
    G, _ = demo.build_generator(torch.device('cpu'))
    G._check_params()
    assert len(G._ptensors) == 674
    marker = object()
    G._prepared = marker
    for _ in range(3):
        G._check_params()
    assert G._prepared is marker                                            # nothing changed: nothing dropped
    with torch.no_grad():
        G._ptensors[100].mul_(1.0)                                          # in-place update (misc.copy_params_and_buffers, an optimizer step)
    G._check_params()
    assert G._prepared is None
    G._prepared = marker
    name, old = next((n, p) for n, p in G.named_parameters() if n.endswith('decoder.net.0.bias'))
    mod = G
    for part in name.split('.')[:-1]:
        mod = getattr(mod, part)
    setattr(mod, name.split('.')[-1], torch.nn.Parameter(old.detach().clone(), requires_grad=False))     # a NEW object, same values, version 0
    G._check_params()                                                      # the very next call sees it
    assert G._prepared is None and any(t is getattr(mod, name.split('.')[-1]) for t in G._ptensors)
    G._prepared = marker
    G.backbone.mapping.register_buffer('w_avg', G.backbone.mapping.w_avg.detach().clone())        # a re-registered buffer
    G._check_params()
    assert G._prepared is None
    G._prepared = marker
    mod._parameters[name.split('.')[-1]] = torch.nn.Parameter(old.detach().clone(), requires_grad=False)    # bypasses the module API
    for _ in range(257):
        G._check_params()
    assert G._prepared is None


def test_prepared_weight_cache_skips_inference_tensors_and_can_be_cleared():
    """conv2d_gradfix._prepared: tensors created under torch.inference_mode() have no version counter (`._version` raises) — they
    are never cached and never crash the operator boundary; `.data` updates bypass the counter, `clear_prep_cache()` (called by
    generator.refresh()) is the documented way to drop the cache."""
    import torch
    from next3d_amd.torch_utils.ops import conv2d_gradfix as cg
    from next3d_amd.torch_utils.ops import upfirdn2d as uf
    with torch.inference_mode():
        w = torch.randn(4, 4, 3, 3) * 0.5
        f = uf.setup_filter([1, 3, 3, 1])
        assert cg._tensor_version(w) is None
        a = uf.fir_factor(f)                                                # must not raise
        # round 6: the FILTER of an inference-mode model IS cached — tied to the tensor object and its storage address — so that no forward (and no
        # HIP-graph capture) re-reads it from the device; a new tensor object never sees the entry
        assert a is not None and id(f) in uf._FIR1D and uf.fir_factor(f) is a
        f2 = f.clone()
        assert uf.fir_factor(f2) is not a
    p = torch.nn.Parameter(torch.randn(4, 4, 3, 3))
    assert cg._tensor_version(p) == p._version
    cg._PREP_CACHE[12345] = {}
    cg.clear_prep_cache()
    assert not cg._PREP_CACHE


def test_ptr_keepalive_window():
    """`_lib.ptr()` holds every tensor whose pointer crosses the C ABI until `check()` (the launch is enqueued by then)."""
    import weakref
    import torch
    from next3d_amd import _lib
    t = torch.zeros(8)
    r = weakref.ref(t)
    p = _lib.ptr(t)
    del t
    assert r() is not None and int(p) == r().data_ptr()            # a temporary survives until the call it was marshalled for
    _lib.check(0)
    assert r() is None
    with pytest.raises(RuntimeError):
        _lib.check(-1)


def test_fir_factor_cache_is_tied_to_the_tensor_object():
    """upfirdn2d.fir_factor: the separable factor is cached per filter TENSOR (weak reference + version counter), so a
    new tensor at a recycled address or an in-place update never sees a stale factor, and the entry dies with the tensor."""
    import gc
    import torch
    from next3d_amd.torch_utils.ops import upfirdn2d as uf
    f = uf.setup_filter([1, 3, 3, 1])
    a = uf.fir_factor(f)
    assert a is not None and torch.equal(torch.outer(a, a), f) and uf.fir_factor(f) is a
    f.mul_(2.0)                                                       # in-place update: re-derived, still separable
    b = uf.fir_factor(f)
    assert b is not a and torch.allclose(torch.outer(b, b), f)
    f[0, 0] = 5.0                                                     # no longer an outer product
    assert uf.fir_factor(f) is None
    key = id(f)
    assert key in uf._FIR1D
    del f
    gc.collect()
    assert key not in uf._FIR1D


def test_third_party_shims_host_side():
    """next3d_amd/shims: module aliasing, the call surface the reference uses, and loud failures for everything else."""
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
import next3d_amd
for k in [k for k in sys.modules if k == 'cv2' or k.startswith('pytorch3d')]:
    del sys.modules[k]
next3d_amd.install_dropin(third_party=True)
import cv2, pytorch3d
from pytorch3d.io import load_obj
from pytorch3d.structures import Meshes
from pytorch3d.renderer.mesh import rasterize_meshes
from cv2 import norm                                            # ray_marcher.py:16 imports it (and never calls it)
assert cv2.__name__ == 'next3d_amd.shims.cv2' and Meshes.__module__ == 'next3d_amd.shims.pytorch3d.structures'
assert cv2.imread('/nonexistent/uv_face_eye_mask.png') is None  # OpenCV's behaviour
for bad in (lambda: cv2.resize(None), lambda: norm(1), lambda: cv2.floodFill(np.zeros((4, 4), np.float32), None, (1, 1), 255, 0, 254, cv2.FLOODFILL_FIXED_RANGE),
            lambda: cv2.floodFill(np.zeros((4, 4), np.float64), None, (0, 0), 255, 0, 254, cv2.FLOODFILL_FIXED_RANGE)):
    try:
        bad()
    except RuntimeError:
        continue
    raise AssertionError('expected RuntimeError')
m = Meshes(verts=torch.zeros(2, 5, 3), faces=torch.zeros(1, 3, 3, dtype=torch.long).expand(2, -1, -1))
assert len(m) == 2 and m.faces_padded().stride(0) == 0 and tuple(m.faces_packed().shape) == (6, 3)
for kw in (dict(blur_radius=1e-4, faces_per_pixel=1), dict(blur_radius=0.0, faces_per_pixel=8), dict(blur_radius=0.0, faces_per_pixel=1, perspective_correct=True)):
    try:
        rasterize_meshes(m, image_size=64, **kw)
    except RuntimeError:
        continue
    raise AssertionError('expected RuntimeError')
try:
    rasterize_meshes(m, image_size=64, blur_radius=0.0, faces_per_pixel=1)      # CPU tensors: no fallback
except RuntimeError as e:
    assert 'HIP device' in str(e)
else:
    raise AssertionError('expected RuntimeError')
print('SHIMS_OK')
""" % REPO
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'SHIMS_OK' in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
    from next3d_amd import mesh
    from next3d_amd.shims.pytorch3d.io import load_obj
    obj = os.path.join(os.path.dirname(__file__), 'golden', '_tiny.obj')
    with open(obj, 'w') as fh:
        fh.write('v 0 0 0\nv 1 0 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 0 1\nf 1/1 2/2 3/3\n')
    try:
        v, f, aux = load_obj(obj)
        pv, pf, puv, pft = mesh.parse_obj(obj)
        assert torch.equal(v, pv) and torch.equal(f.verts_idx, pf) and torch.equal(f.textures_idx, pft) and torch.equal(aux.verts_uvs, puv)
    finally:
        os.remove(obj)


@pytest.mark.skipif(not os.path.isfile('/root/reference/gen_samples_next3d.py'), reason='needs the reference tree (build container only)')
@pytest.mark.parametrize('script', ['gen_samples_next3d.py', 'gen_videos_next3d.py', 'reenact_avatar_next3d.py'])
def test_launcher_runs_reference_scripts_unchanged(script):
    """`python -m next3d_amd.run <script> --help`: the reference's script, unedited, is executed as __main__ after install_dropin —
    its imports of torch_utils.ops.*, training_avatar_texture.triplane_next3d, pytorch3d and cv2 resolve to next3d_amd, the
    rest (dnnlib, legacy, camera_utils, click options) to the reference tree next to the script.  (`--help` ends the run before a
    pickle or a GPU is needed.  Packages this container lacks and the scripts import at the top — mrcfile, imageio, torchvision,
    pydantic.NoneStr, turtle — are stubbed by oracle/ref_shims.py as for the pin script; on a deployment they are installed.)"""
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
from oracle import ref_shims
ref_shims.install(np.zeros((4, 4), np.float32), third_party=False)          # import-time stubs only; cv2 / pytorch3d are left to install_dropin
sys.path.remove('/root/reference')                                          # the launcher must find the tree from the script path itself
from next3d_amd import run
try:
    run.main(['--third-party', 'shims', '/root/reference/%s', '--help'])
except SystemExit as e:
    assert e.code in (0, None), e.code
import torch_utils.ops.bias_act as ba, training_avatar_texture.triplane_next3d as tp, cv2, pytorch3d
assert ba.__name__ == 'next3d_amd.torch_utils.ops.bias_act' and tp.__name__ == 'next3d_amd.generator', (ba.__name__, tp.__name__)
assert cv2.__name__ == 'next3d_amd.shims.cv2' and pytorch3d.__name__ == 'next3d_amd.shims.pytorch3d'
import legacy, dnnlib
assert legacy.__file__.startswith('/root/reference/') and dnnlib.__file__.startswith('/root/reference/')
print('LAUNCHER_OK')
""" % (repo, script)
    r = subprocess.run([sys.executable, '-c', code], cwd='/tmp', capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'LAUNCHER_OK' in r.stdout and '--network' in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


@pytest.mark.skipif(not os.path.isfile('/root/reference/gen_samples_next3d.py'), reason='needs the reference tree (build container only)')
def test_reference_scripts_run_end_to_end_through_the_launcher(tmp_path_factory):
    """"drop in unchanged", once, end to end.  The reference's OWN generator is pickled by the reference's own persistence
    ({'G_ema': G}), then gen_samples_next3d.py (1 seed), gen_videos_next3d.py (2x2 grid, 2 frames) and reenact_avatar_next3d.py (2 frames of a
    synthetic driving sequence) are executed UNCHANGED by `next3d_amd.run` with --reload_modules=True against the recording stand-in of
    libn3d.so (tests/_e2e_scripts.py): legacy.load_network_pkl -> TriPlaneGenerator(*G.init_args, **G.init_kwargs) -> misc.copy_params_and_buffers
    -> the scripts' image loops -> PNG / video frames.  Every G.synthesis call lands in next3d_amd.generator, carries the pickle's weights and issues
    exactly the launch sequence of a direct call (156 launches on the scripts' default float16 super-resolution route)."""
    import json
    work = tmp_path_factory.mktemp('e2e')
    pkl = str(work / 'synthetic_next3d.pkl')
    harness = os.path.join(os.path.dirname(__file__), '_e2e_scripts.py')
    try:
        r = subprocess.run([sys.executable, harness, 'write-pickle', pkl], cwd='/tmp', capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and 'PICKLE_OK' in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
        r = subprocess.run([sys.executable, harness, 'run', pkl, str(work)], cwd='/tmp', capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0 and 'E2E_OK' in r.stdout, r.stdout[-1500:] + r.stderr[-4000:]
    finally:
        if os.path.exists(pkl):
            os.remove(pkl)                                            # 0.7 GB
    rep = json.loads(r.stdout[r.stdout.index('E2E_OK') + 7:].splitlines()[0])
    assert rep['gen_samples_next3d.py']['synthesis_calls'] == 3 and rep['gen_samples_next3d.py']['mapping_calls'] == 3          # three views of one seed
    assert rep['gen_videos_next3d.py']['synthesis_calls'] == 1 + 2 * 4 and rep['reenact_avatar_next3d.py']['synthesis_calls'] == 2
    assert all(v['launches_per_synthesis'] == rep['gen_samples_next3d.py']['launches_per_synthesis'] and v['sequences_checked'] >= 1 for v in rep.values())


def test_osg_decoder_unpickling_target_matches_oracle():
    """generator.OSGDecoder: the class a network pickle names by reference (tat/triplane_next3d.py:348 is not a persistent class) — constructor
    signature, parameter names (decoder.net.{0,2}.{weight,bias}) and forward against the oracle's restatement of the reference decoder."""
    import torch
    from next3d_amd import generator, spec
    from oracle import renderer as orend
    dec = generator.OSGDecoder(32, {'decoder_lr_mul': 1.0, 'decoder_output_dim': 32})
    assert sorted(k for k, _ in dec.named_parameters()) == ['net.0.bias', 'net.0.weight', 'net.2.bias', 'net.2.weight']
    sd = spec.synthetic_state_dict(0, only=lambda n: n.startswith('decoder.'))
    dec.load_state_dict({k[len('decoder.'):]: v for k, v in sd.items()}, strict=True)
    feats = torch.randn(2, 3, 50, 32, generator=torch.Generator().manual_seed(3))
    out = dec(feats, None)
    rgb, sigma = orend.osg_decoder(sd, 'decoder', feats)
    assert tuple(out['rgb'].shape) == (2, 50, 32) and tuple(out['sigma'].shape) == (2, 50, 1)
    assert float((out['rgb'] - rgb).abs().max()) <= 1e-6 and float((out['sigma'] - sigma).abs().max()) <= 1e-5


def test_profile_tools_count_every_split_bf16_3x3_kernel():
    """tools/traffic_summary.py selects "the 3x3 split-bf16 family" by kernel name: every such kernel defined in csrc/ must match (a new variant that
    did not once left `roofline.traffic` on a stale profile), and nothing else may (1x1, split-K reduce, fp32-MFMA and f16 kernels are other families)."""
    import glob
    import importlib.util
    import re
    spec_ = importlib.util.spec_from_file_location('traffic_summary', os.path.join(REPO, 'tools', 'traffic_summary.py'))
    src = open(spec_.origin).read()
    fam = re.compile(re.search(r"FAMILY = re\.compile\(r'([^']+)'\)", src).group(1))
    names = set()
    for f in glob.glob(os.path.join(REPO, 'next3d_amd', 'csrc', '*.hip')):
        names |= set(re.findall(r'__global__[^;{]*?void\s+(\w+_kernel)\s*\(', open(f).read()))
    family = {n for n in names if re.fullmatch(r'conv2d(_\w+)?_bf16x3(_pair)?_kernel', n)}
    assert len(family) >= 14 and 'conv2d_ps2_rgb32s_bf16x3_kernel' in family and 'conv2d_sk_bf16x3_kernel' in family
    for n in sorted(names):
        for shown in (n, 'void ' + n + '<1, 2>'):                       # (rocprofv3 prints template kernels as "void name<args>")
            assert bool(fam.match(shown)) == (n in family), shown
    import bench
    assert os.path.isfile(bench.TRAFFIC_PROFILE) and os.path.basename(bench.TRAFFIC_PROFILE).startswith('r05_')
