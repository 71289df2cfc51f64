import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from next3d_amd import demo
from oracle import cases
dev = torch.device('cuda', 0)
G, _ = demo.build_generator(dev)
G.keep_stages = True
d = np.load('tests/golden/case_r64_s48.npz')
N, R, Sc, Sf = 2, 32, 24, 24
G.rendering_kwargs['depth_resolution'], G.rendering_kwargs['depth_resolution_importance'] = Sc, Sf
jitter, u = cases.rng_inputs(N, R, Sc, Sf)
t = lambda k: torch.from_numpy(d[k]).to(dev)
ws = G.mapping(t('z'), t('c_cond'), truncation_psi=0.7, truncation_cutoff=14)
kw = dict(neural_rendering_resolution=R, noise_mode='const', depth_jitter=jitter, importance_u=u)
c, v = t('c'), t('v')
SYNC = len(sys.argv) > 1
def snap():
    if SYNC: torch.cuda.synchronize()
    s = {k: x.clone() for k, x in G._debug.items() if torch.is_tensor(x)}
    return s
def df(a, b): return {k: round(float((a[k].float() - b[k].float()).abs().max()), 5) for k in a if k in b}
outs = []
for i in range(4):
    o = G.synthesis(ws, c, v, cache_backbone=True, **kw); s = snap(); s['planes'] = G._last_planes[0].clone(); s.update({k: x.clone() for k, x in o.items()}); outs.append(s)
for i in range(1, 4):
    print(f'run0 vs run{i}', {k: x for k, x in df(outs[0], outs[i]).items() if x > 0})
