"""Ahead-of-time build of libn3d.so (HIP, gfx950 only) and of the oracle's C helper.

    python -m next3d_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting in-tree .so travels to the GPU box with the snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'build')
LIB = os.path.join(HERE, 'libn3d.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
ARCH = 'gfx950'
CFLAGS = ['-O3', '-std=c++17', '-fPIC', f'--offload-arch={ARCH}', '-Wall', '-Wno-unused-function', '-Wno-unused-variable']
# raster.hip mirrors the C oracle operation-for-operation: no FMA contraction there (bit-exact face selection)
PER_FILE = {'raster.hip': ['-ffp-contract=off']}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    headers.append(os.path.join(HERE, '..', 'include', 'n3d.h'))
    jobs = []
    for src in _sources():
        obj = os.path.join(OBJ, src.replace('.hip', '.o'))
        if force or _stale(obj, [os.path.join(CSRC, src)] + headers):
            jobs.append([HIPCC] + CFLAGS + PER_FILE.get(src, []) + ['-c', os.path.join(CSRC, src), '-o', obj])

    def run(cmd):
        if verbose:
            print('[n3d build]', ' '.join(cmd[-3:]), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed: {" ".join(cmd)}\n{r.stdout}\n{r.stderr}')
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    objs = [os.path.join(OBJ, s.replace('.hip', '.o')) for s in _sources()]
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, '-shared', '-fPIC', f'--offload-arch={ARCH}', '-Wl,--no-undefined', '-o', LIB] + objs)
        # a library that links but cannot be loaded (hipcc's host pass silently drops the launch stub of some __global__ templates:
        # an undefined __device_stub__ symbol) must fail HERE, on the build machine, not on the GPU box
        import ctypes
        try:
            import torch                               # noqa: F401  (torch's HIP runtime first: see __graft_entry__.build)
        except ImportError:
            pass
        ctypes.CDLL(LIB)
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
