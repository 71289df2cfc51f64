"""python -m next3d_amd.run [--boundary model|operator] [--third-party auto|shims|real] <script.py> [script arguments ...]

Runs one of the reference's scripts — gen_samples_next3d.py, gen_videos_next3d.py, reenact_avatar_next3d.py — UNCHANGED on
libn3d.so: `install_dropin` is applied before the script's first import, then the script is executed as `__main__` (runpy) with
its own argv.  Nothing in the reference tree is edited.

  --boundary model    (default) `torch_utils.ops.*` AND `training_avatar_texture.triplane_next3d` resolve to next3d_amd: with the script's
                      `--reload_modules True` the generator is next3d_amd.generator.TriPlaneGenerator (boundary B2, the fast route:
                      fused layers, no host round trips); without it the pickled reference modules run on next3d_amd's operators (B1).
  --boundary operator only the operator layer (+ third-party shims) is replaced, whatever --reload_modules says (B1).
  --third-party       pytorch3d / cv2: `auto` = next3d_amd.shims where the real package cannot be imported, `shims` = always,
                      `real` = never (gen_samples_next3d.py:119,150-157; next3d_amd.install_dropin).

The script path may be relative to the current directory or to the directory that holds the reference tree on sys.path[0]; like
`python script.py`, the script's own directory is put first on sys.path, so `import dnnlib`, `import legacy`, `from
training_avatar_texture...` resolve to the reference tree the script lives in.
"""
import os
import runpy
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    boundary, third = 'model', 'auto'
    while argv and argv[0].startswith('--'):
        opt = argv.pop(0)
        if opt in ('-h', '--help'):
            print(__doc__)
            return 0
        key, _, val = opt.partition('=')
        if not val:
            if not argv:
                raise SystemExit(f'next3d_amd.run: {key} needs a value')
            val = argv.pop(0)
        if key == '--boundary' and val in ('model', 'operator'):
            boundary = val
        elif key == '--third-party' and val in ('auto', 'shims', 'real'):
            third = val
        else:
            raise SystemExit(f'next3d_amd.run: unknown option {opt} {val}\n\n{__doc__}')
    if not argv:
        raise SystemExit(__doc__)
    script = argv[0]
    if not os.path.isfile(script):
        raise SystemExit(f'next3d_amd.run: script {script!r} not found')
    script = os.path.abspath(script)
    # what `python script.py` does: the script's directory first on sys.path, argv[0] = the script
    sys.path.insert(0, os.path.dirname(script))
    sys.argv = [script] + argv[1:]
    from . import install_dropin
    install_dropin(model=(boundary == 'model'), third_party={'auto': 'auto', 'shims': True, 'real': False}[third])
    runpy.run_path(script, run_name='__main__')
    return 0


if __name__ == '__main__':
    sys.exit(main())
