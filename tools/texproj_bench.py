#!/usr/bin/env python3
"""n3d_texture_project_planes alone at the benchmark's shape (batch 4, 32-channel 256 x 256 texture, four rasterised views): in-stream
time per launch, alone and with 8-wave convolution workgroups of another stream co-resident.  A/B of the agent-scope gathers:
N3D_LIB=tools/probe/libn3d_noagent.so (raster.hip built with -DTEXPROJ_AGENT=0).  Usage (GPU box): python tools/texproj_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from next3d_amd import demo
    dev = torch.device('cuda', 0)
    G, _ = demo.build_generator(dev)
    z, c, c_cond, v = demo.demo_batch([0, 1, 2, 3], device=dev)
    grid, alpha, bbox = G.raster_geometry(v[:, :5023].contiguous(), v[:, 5023:].contiguous())
    tex = torch.randn(4, 32, 256, 256, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        out = G.project_textures(tex, grid)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50):
        out = G.project_textures(tex, grid)
    e1.record(); torch.cuda.synchronize()
    print(f'{os.environ.get("N3D_LIB", "shipped library")}: texture_project_planes {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per launch (3 planes, batch 4); '
          f'checksum {float(sum(o.double().sum() for o in out)):.6f}')


if __name__ == '__main__':
    main()
