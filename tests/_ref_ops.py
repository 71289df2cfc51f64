"""Loader for tests/golden/ref_ops.npz — operator-level fixtures whose outputs come from the REFERENCE's own `_ref`
implementations (written by oracle/pin_ops_against_reference.py; see there for the list of cases)."""
import json
import os

import numpy as np
import torch

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_ops.npz')


def load():
    """[(index, op, kwargs, {name: tensor}, reference output)]"""
    d = np.load(PATH)
    out = []
    for i, c in enumerate(json.loads(str(d['index']))):
        t = {k: torch.from_numpy(d[f'c{i}_{k}']) for k in c['inputs']}
        out.append((i, c['op'], c['kw'], t, torch.from_numpy(d[f'c{i}_y'])))
    return out


def run(mod, op, kw, t, setup_filter, to=lambda v: v):
    """One case through `mod` (dict op-name -> callable: oracle.ops or the B1 operator layer); `to` moves tensors."""
    kw = {k: v for k, v in kw.items() if k != 'shape'}
    x, b = to(t['x']), (to(t['b']) if 'b' in t else None)
    if op == 'bias_act':
        return mod['bias_act'](x, b, **kw)
    if op == 'upfirdn2d':
        taps = kw.pop('taps')
        return mod['upfirdn2d'](x, None if taps is None else to(setup_filter(taps)), **kw)
    tu, td = kw.pop('taps_up'), kw.pop('taps_down')
    return mod['filtered_lrelu'](x, fu=None if tu is None else to(setup_filter(tu)), fd=None if td is None else to(setup_filter(td)),
                                 b=b, **kw)
