"""End-to-end harness of tests/test_cpu_host.py::test_reference_scripts_run_end_to_end_through_the_launcher (build container only: it runs the
REFERENCE's scripts from /root/reference).  TEST INFRASTRUCTURE.

    python tests/_e2e_scripts.py write-pickle <file.pkl>     the reference's OWN TriPlaneGenerator (oracle/ref_shims.build_reference_generator, seeded
                                                             synthetic weights) pickled as {'G_ema': G} by the reference's own torch_utils.persistence
    python tests/_e2e_scripts.py run <file.pkl> <workdir>    `python -m next3d_amd.run <script> ...` for the three scripts, UNCHANGED, in one process:
                                                             legacy.load_network_pkl -> --reload_modules=True -> TriPlaneGenerator(*G.init_args,
                                                             **G.init_kwargs) -> misc.copy_params_and_buffers -> the scripts' own image loops

There is no GPU here: libn3d.so is replaced by the recording stand-in of tests/_dryrun.py (every launch is marshalled and checked, nothing runs) and
a TorchFunctionMode maps the scripts' `torch.device('cuda')` / `.cuda()` / `.to(device)` to the CPU.  What is asserted: every `G.mapping` /
`G.synthesis` call the scripts make lands in next3d_amd.generator.TriPlaneGenerator with the reference's arguments, carries the pickle's
weights, and issues EXACTLY the launch sequence a direct call of that class issues (the B2 sequence of test_generator_launch_sequence_dry_run).
"""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'tests'))
REF = '/root/reference'


def write_pickle(path):
    import pickle
    import torch
    from next3d_amd import mesh as n3d_mesh, spec
    from oracle import pin_against_reference as pin, ref_shims
    uv = n3d_mesh.synthetic_uv_face_mask()
    ref_shims.install(uv[0, 0].numpy())
    G = ref_shims.build_reference_generator(pin.RENDERING_KWARGS)
    sd = spec.synthetic_state_dict(0)
    _, faces, uvs, uvfaces = n3d_mesh.parse_obj(os.path.join(REF, 'data/demo/demo.obj'))
    sd.update(n3d_mesh.mesh_buffers(faces, uvs, uvfaces))
    G.load_state_dict(sd, strict=True)
    with open(path, 'wb') as fh:
        pickle.dump({'G_ema': G, 'G': None, 'D': None, 'training_set_kwargs': None, 'augment_pipe': None}, fh)
    print('PICKLE_OK', os.path.getsize(path))


def _fake_cuda_mode():
    import torch
    from torch.overrides import TorchFunctionMode

    class FakeCuda(TorchFunctionMode):
        """The scripts ask for 'cuda'; this container has none: device arguments are mapped to the CPU (the launches go to the recorder anyway)."""

        def __torch_function__(self, func, types, args=(), kwargs=None):
            kwargs = kwargs or {}
            if func is torch.Tensor.cuda:
                return args[0]

            def fix(a):
                if isinstance(a, torch.device) and a.type == 'cuda':
                    return torch.device('cpu')
                if isinstance(a, str) and (a == 'cuda' or a.startswith('cuda:')):
                    return 'cpu'
                return a
            return func(*[fix(a) for a in args], **{k: fix(v) for k, v in kwargs.items()})
    return FakeCuda()


def run(pkl, work):
    import numpy as np
    import torch
    from PIL import Image
    os.environ['N3D_OVERLAP_STATIC'] = '0'                  # the side stream needs a real device (the dry-run tests switch it off the same way)
    from next3d_amd import mesh as n3d_mesh, spec
    from oracle import ref_shims
    uv = n3d_mesh.synthetic_uv_face_mask()
    ref_shims.install(uv[0, 0].numpy(), third_party=False)  # import-time stubs only (mrcfile, imageio, torchvision, turtle, pydantic.NoneStr)
    sys.path.remove(REF)                                    # the launcher must find the tree from the script path itself
    # the working directory a user would run in: the assets the reference opens by RELATIVE path (triplane_next3d.py:91 the uv mask, which the
    # reference tree does not ship; init_args' topology_path 'data/demo/demo.obj')
    os.makedirs(os.path.join(work, 'data/ffhq'), exist_ok=True)
    if not os.path.exists(os.path.join(work, 'data/demo')):
        os.symlink(os.path.join(REF, 'data/demo'), os.path.join(work, 'data/demo'))
    m = (uv[0, 0].numpy() * 255).round().astype(np.uint8)
    Image.fromarray(np.stack([m] * 3, -1)).save(os.path.join(work, 'data/ffhq/uv_face_eye_mask.png'))
    os.chdir(work)
    # a tiny driving sequence for reenact_avatar_next3d.py: 4 frames of the demo mesh, their landmarks and camera labels
    drive = os.path.join(work, 'drive')
    os.makedirs(drive, exist_ok=True)
    from next3d_amd import demo
    cam = demo.camera_label(0.1, -0.1)[0].tolist()
    labels = []
    for k in range(4):
        Image.fromarray(np.zeros((512, 512, 3), np.uint8)).save(os.path.join(drive, f'{k:05d}.png'))
        for src, dst in (('demo.obj', f'{k:05d}.obj'), ('demo_kpt2d.txt', f'{k:05d}_kpt2d.txt')):
            if not os.path.exists(os.path.join(drive, dst)):
                os.symlink(os.path.join(REF, 'data/demo', src), os.path.join(drive, dst))
        labels.append([f'{k:05d}.png', cam])
    json.dump({'labels': labels}, open(os.path.join(drive, 'dataset.json'), 'w'))

    import _dryrun
    patches, calls = _dryrun.patches()
    for obj, attr, val in patches:
        setattr(obj, attr, val)
    import imageio                                          # (ref_shims' empty stand-in: the video writer the two video scripts open)
    written = []

    class _Writer:
        def append_data(self, frame):
            written.append(tuple(frame.shape))

        def close(self):
            pass
    imageio.get_writer = lambda *a, **k: _Writer()

    from next3d_amd import generator, run as launcher
    log = []                                                # (method, instance, kwargs, first launch index, last launch index)
    for name in ('mapping', 'synthesis'):
        orig = getattr(generator.TriPlaneGenerator, name)

        def wrapped(self, *a, _orig=orig, _name=name, **k):
            start = len(calls)
            out = _orig(self, *a, **k)
            names = ('ws', 'c', 'v') if _name == 'synthesis' else ('z', 'c')
            full = dict(zip(names, a)); full.update(k)          # (gen_videos_next3d.py passes ws= / c= / v= by keyword)
            log.append((_name, self, {q: val for q, val in full.items() if q not in names}, tuple(tuple(full[q].shape) for q in names), start, len(calls)))
            return out
        setattr(generator.TriPlaneGenerator, name, wrapped)

    common = ['--network', pkl, '--reload_modules', 'True', '--trunc', '0.7', '--lms_cond', 'True']
    demo_args = ['--obj_path', 'data/demo/demo.obj', '--lms_path', 'data/demo/demo_kpt2d.txt']
    jobs = {
        'gen_samples_next3d.py': common + demo_args + ['--seeds', '0', '--outdir', os.path.join(work, 'out_samples')],
        'gen_videos_next3d.py': common + demo_args + ['--seeds', '10720,12374,13393,17099', '--grid', '2x2', '--w-frames', '2', '--outdir', os.path.join(work, 'out_videos')],
        'reenact_avatar_next3d.py': common + ['--seeds', '0', '--drive_root', drive, '--grid', '2x1', '--num_frames', '2', '--outdir', os.path.join(work, 'out_reenact')],
    }
    report = {}
    with _fake_cuda_mode():
        for script, args in jobs.items():
            log.clear(); calls.clear(); written.clear()
            try:
                launcher.main(['--third-party', 'shims', os.path.join(REF, script)] + args)
            except SystemExit as e:                         # click's standalone mode ends every command with sys.exit(0)
                assert e.code in (0, None), (script, e.code)
            import training_avatar_texture.triplane_next3d as tp
            assert tp.__name__ == 'next3d_amd.generator'
            syn = [e for e in log if e[0] == 'synthesis']
            mp = [e for e in log if e[0] == 'mapping']
            assert syn and mp, script
            inst = syn[-1][1]
            assert type(inst).__module__ == 'next3d_amd.generator' and all(e[1] is inst for e in log), script
            # the pickle's weights arrived (legacy.load_network_pkl -> misc.copy_params_and_buffers)
            want = spec.synthetic_state_dict(0, only=lambda n: n in ('decoder.net.0.weight', 'superresolution.block1.conv1.weight'))
            have = inst.state_dict()
            assert all(torch.equal(have[k], want[k]) for k in want), script
            # ... and every synthesis call issued the launch sequence of a DIRECT call with the same arguments (B2)
            seqs = {}
            for _, _, kw, shapes, a, b in syn[1:]:          # (the first call also prepares the weights)
                key = (json.dumps({k: (v if isinstance(v, (int, float, str, bool, type(None))) else str(type(v))) for k, v in sorted(kw.items())}), shapes)
                seqs.setdefault(key, []).append(tuple(calls[a:b]))
            checked = 0
            for (kwj, shapes), recorded in seqs.items():
                kw = json.loads(kwj)
                n = shapes[0][0]
                ws, c, v = torch.zeros(shapes[0]), torch.zeros(n, 25), torch.zeros(shapes[-1])
                c[:, :16] = demo.camera_label()[0, :16]; c[:, 16:] = demo.camera_label()[0, 16:]
                start = len(calls)
                inst.synthesis(ws, c, v, **kw)
                direct = tuple(calls[start:len(calls)])
                assert all(r == direct for r in recorded), (script, kwj, len(recorded[0]), len(direct))
                checked += len(recorded)
            report[script] = dict(mapping_calls=len(mp), synthesis_calls=len(syn), sequences_checked=checked, launches_per_synthesis=sorted({len(r) for rs in seqs.values() for r in rs}),
                                  synthesis_kwargs=sorted({k for e in syn for k in e[2]}), frames_written=list(written))
    # what the scripts wrote
    assert os.path.isfile(os.path.join(work, 'out_samples', 'seed0000.png'))
    assert report['gen_videos_next3d.py']['frames_written'] == [(1024, 1024, 3)] * 2, report['gen_videos_next3d.py']
    assert report['reenact_avatar_next3d.py']['frames_written'] == [(512, 1024, 3)] * 2, report['reenact_avatar_next3d.py']
    print('E2E_OK', json.dumps(report))


if __name__ == '__main__':
    if sys.argv[1] == 'write-pickle':
        write_pickle(sys.argv[2])
    else:
        run(sys.argv[2], sys.argv[3])
