#!/usr/bin/env python3
"""GPU box: which Python lines of one eager batch-1 forward issue torch copies / fills (aten::copy_, aten::fill_, aten::cat ...)?  torch.profiler with stacks;
prints every aten op of the forward that is not a view / allocation, with the innermost next3d_amd frame.  [--batch N]"""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from next3d_amd import demo
B = int(sys.argv[sys.argv.index('--batch') + 1]) if '--batch' in sys.argv else 1
dev = torch.device('cuda', 0)
G, _ = demo.build_generator(dev)
z, c, c_cond, v = demo.demo_batch(list(range(B)), device=dev)
R, Sc, Sf = 64, 48, 48
jit, u = torch.rand((B, R * R, Sc, 1), device=dev), torch.rand((B * R * R, Sf), device=dev)


def frame():
    ws = G.mapping(z, c_cond, truncation_psi=0.7, truncation_cutoff=14)
    return G.synthesis(ws, c, v, neural_rendering_resolution=R, noise_mode='const', depth_jitter=jit, importance_u=u)['image']


for _ in range(3):
    frame()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    frame()
    torch.cuda.synchronize()
VIEWS = ('aten::empty', 'aten::view', 'aten::slice', 'aten::select', 'aten::as_strided', 'aten::reshape', 'aten::expand', 'aten::unsqueeze', 'aten::narrow', 'aten::permute',
         'aten::_unsafe_view', 'aten::alias', 'aten::detach', 'aten::empty_strided', 'aten::empty_like', 'aten::to', 'aten::_to_copy', 'aten::contiguous', 'aten::lift_fresh', 'aten::squeeze',
         'aten::transpose', 'aten::t', 'aten::result_type', 'aten::is_', 'aten::unbind', 'aten::item', 'aten::_local_scalar_dense', 'aten::zeros', 'aten::clone', 'aten::unflatten', 'aten::flatten')
cnt = collections.Counter()
for e in prof.events():
    if not e.name.startswith('aten::') or e.name in VIEWS:
        continue
    where = next((s for s in e.stack if 'next3d_amd' in s), (e.stack[0] if e.stack else '?'))
    cnt[(e.name, where.strip()[:150])] += 1
for (name, where), k in sorted(cnt.items(), key=lambda t: -t[1]):
    print(f'{k:3d} x {name:28s} {where}')
gpu = collections.Counter(e.name for e in prof.events() if e.device_type is not None and 'cuda' in str(e.device_type).lower())
print({k: n for k, n in gpu.items() if 'copy' in k.lower() or 'Memcpy' in k or 'Memset' in k})
