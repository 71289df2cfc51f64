#!/usr/bin/env python3
"""oracle/dump_ref_api.py — writes tests/golden/ref_ops_api.txt: the argument lists of the reference's operator layer
(torch_utils/ops/*.py, SURVEY.md §8b "B1"), taken from its SOURCE with `ast` (decorators such as misc.profiled_function hide
the signature from `inspect`).  TEST INFRASTRUCTURE; runs only where /root/reference exists.  tests/test_cpu_host.py compares
next3d_amd/torch_utils/ops/*.py against the committed file."""
import ast
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get('N3D_REFERENCE', '/root/reference')
API = {
    'bias_act': ['bias_act'],
    'upfirdn2d': ['setup_filter', 'upfirdn2d', 'upsample2d', 'downsample2d', 'filter2d', '_parse_padding', '_get_filter_size', '_parse_scaling'],
    'conv2d_resample': ['conv2d_resample', '_get_weight_shape', '_conv2d_wrapper'],
    'conv2d_gradfix': ['conv2d', 'conv_transpose2d', 'no_weight_gradients'],
    'fma': ['fma'],
    'filtered_lrelu': ['filtered_lrelu'],
}


def signatures(path, names):
    tree = ast.parse(open(path).read())
    found = {n.name: ast.unparse(n.args) for n in tree.body if isinstance(n, ast.FunctionDef)}
    return {name: found.get(name) for name in names}


MODEL_METHODS = ['__init__', 'mapping', 'synthesis', 'sample', 'sample_mixed', 'forward']     # triplane_next3d.py:41-328 (B2)


def class_methods(path, cls, names):
    tree = ast.parse(open(path).read())
    for n in tree.body:
        if isinstance(n, ast.ClassDef) and n.name == cls:
            found = {m.name: ast.unparse(m.args) for m in n.body if isinstance(m, ast.FunctionDef)}
            return {name: found.get(name) for name in names}
    raise KeyError(cls)


def main():
    out = []
    for mod, names in API.items():
        for name, sig in signatures(os.path.join(REF, 'torch_utils', 'ops', mod + '.py'), names).items():
            assert sig is not None, (mod, name)
            out.append(f'{mod}.{name}({sig})')
    for name, sig in class_methods(os.path.join(REF, 'training_avatar_texture', 'triplane_next3d.py'), 'TriPlaneGenerator', MODEL_METHODS).items():
        assert sig is not None, name
        out.append(f'TriPlaneGenerator.{name}({sig})')
    with open(os.path.join(REPO, 'tests', 'golden', 'ref_ops_api.txt'), 'w') as fh:
        fh.write('\n'.join(out) + '\n')
    print('\n'.join(out))


if __name__ == '__main__':
    sys.exit(main())
