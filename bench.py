#!/usr/bin/env python3
"""bench.py — generator-forward frames/s of the Next3D hot path on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch: `G.mapping` + `G.synthesis` for B=4 seeds per GPU at
512² output, 64² neural render, 48 coarse + 48 importance depth samples, trunc 0.7, demo mesh (BASELINE.json
configs[1]); inputs are resident in HBM before the timed region.  With N GPUs every rank renders its own B seeds
(independent seeds shard with no data-path collective — weak scaling) and the finished uint8 frames are gathered to
rank 0 over RCCL inside the timed step (north_star: "RCCL over xGMI only for the final gather").
Rank 0 prints ONE JSON line; `roofline` is the dominant kernel family — the 3x3 split-bf16 convolutions on the bf16 matrix
cores (fp32-MFMA convolutions with N3D_PRECISION=fp32) — timed with HIP events on the launch stream; `cpu_baseline` is the
CPU oracle (a port of the reference's fp32 path) timed on the host cores.
"""
import argparse
import json
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_FP32_MFMA_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md (dense fp32 matrix peak)
PEAK_BF16_MFMA_TFLOPS = 2500.0         # dense bf16 matrix peak (same guide); bf16x3 spends 3 bf16 MFMAs per algorithmic MAC


def cpu_baseline(seconds_budget=25.0):
    """The oracle (kind='port': bit-exact restatement of the reference's CPU fp32 path, oracle/pin_against_reference.py)
    on the host cores: N=1 frames of the same workload (R=64, 48+48), bounded sample."""
    from next3d_amd import demo, mesh, spec
    from oracle import cases, generator as ogen
    threads = min(os.cpu_count() or 1, 64)
    torch.set_num_threads(threads)
    d = demo.demo_arrays()
    sd = spec.synthetic_state_dict(0)
    sd.update(mesh.mesh_buffers(d['faces'], d['uvs'], d['uvfaces']))
    rk = dict(demo.RENDERING_KWARGS)
    z, c, c_cond, v = demo.demo_batch([0], yaws=[0.0])
    jitter, u = cases.rng_inputs(1, 64, 48, 48)
    mask = mesh.synthetic_uv_face_mask()

    def frame():
        ws = ogen.mapping(sd, z, c_cond, rk, truncation_psi=0.7, truncation_cutoff=14)
        return ogen.synthesis(sd, ws, c, v, mask, rk, jitter, u, neural_rendering_resolution=64)
    frame()                                            # warm-up
    n, t0 = 0, time.time()
    while n < 2 or (time.time() - t0 < seconds_budget and n < 8):
        frame()
        n += 1
    dt = time.time() - t0
    return {'value': n / dt, 'unit': 'frames/s', 'cores': threads, 'kind': 'port',
            'sample': f'{n} frames, batch 1, 512²/64²/48+48, fp32, torch-CPU oracle with {threads} threads'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=4, help='seeds per GPU per step (BASELINE.json configs[1]: 4)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit('launch multi-GPU runs with torch.distributed.run (one process per GPU)')
        args.gpus = world
    assert torch.cuda.is_available(), 'bench.py needs a HIP device'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=dev)        # 'nccl' == RCCL on ROCm

    from next3d_amd import _lib, demo
    _lib.lib()
    G, _ = demo.build_generator(dev)
    B = args.batch
    seeds = [rank * B + i for i in range(B)]                  # seed-sharded: rank r owns seeds r*B .. r*B+B-1
    z, c, c_cond, v = demo.demo_batch(seeds, device=dev)
    R, Sc, Sf = 64, 48, 48
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    jitter = torch.rand((B, R * R, Sc, 1), device=dev, generator=g)
    u = torch.rand((B * R * R, Sf), device=dev, generator=g)

    from next3d_amd.sharding import AsyncFrameGather
    # asynchronous (host never blocks) gather of the finished frames, one per step (sharding.AsyncFrameGather)
    gatherer = AsyncFrameGather(torch.empty(B, 3, 512, 512, dtype=torch.uint8, device=dev), dst=0)

    def drain():
        gatherer.drain()

    def step():
        # the previous step's gather is joined BEFORE this step's kernels are enqueued: the rasteriser must not overlap another
        # stream's kernels, and an RCCL gather kernel is exactly that
        gatherer.drain()
        ws = G.mapping(z, c_cond, truncation_psi=0.7, truncation_cutoff=14)
        img = G.synthesis(ws, c, v, neural_rendering_resolution=R, noise_mode='const', depth_jitter=jitter, importance_u=u)['image']
        frames = torch.empty(img.shape, dtype=torch.uint8, device=dev)       # gen_samples_next3d.py:201 (NCHW kept)
        _lib.check(_lib.lib().n3d_to_uint8(_lib.ptr(img), _lib.ptr(frames), img.numel(), _lib.stream()))
        gatherer.submit(frames)
        return frames

    def sync():
        drain()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the inputs do not change between steps, so pipelined steps must return identical frames: a cheap guard against stream
    # races in exactly the configuration that was timed (tests/test_generator_gpu.py has the per-stage version)
    fa = step().clone(); fb = step().clone(); fc = step()
    torch.cuda.synchronize()
    reproducible = bool(torch.equal(fa, fb) and torch.equal(fb, fc))

    roofline = None
    if not args.no_roofline:
        # same K steps again with per-launch HIP events on the launch stream (kept out of the timed region above so
        # the event records cannot perturb `value`)
        # ... and with the static-backbone side stream switched off for this pass: two streams' kernels overlap in time, which
        # would stretch every family's event time; the roofline wants each kernel's own duration
        overlap, G.overlap_static = G.overlap_static, False
        step(); torch.cuda.synchronize()
        _lib.prof_reset()
        _lib.prof_enable(True)
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        _lib.prof_enable(False)
        G.overlap_static = overlap
        prof = _lib.prof_read()
        _lib.prof_reset()
        c16, c32 = prof['conv2d_bf16x3'], prof['conv2d']
        if c16['ms'] >= c32['ms']:       # dominant kernel family: split-bf16 conv on the bf16 matrix cores
            dom, name, peak = c16, ('3x3 split-bf16 conv family: conv2d_p_bf16x3_kernel (persistent) + conv2d_bf16x3_kernel + conv2d_up_bf16x3_kernel + conv2d_s2_bf16x3_kernel '
                                    '(all launches of the step)'), PEAK_BF16_MFMA_TFLOPS / 3.0
        else:                            # N3D_PRECISION=fp32: fp32-MFMA conv
            dom, name, peak = c32, 'conv2d_mfma_kernel (all launches of the step)', PEAK_FP32_MFMA_TFLOPS
        achieved = dom['flops'] / (dom['ms'] * 1e-3) / 1e12 if dom['ms'] > 0 else 0.0
        traffic = traffic_x2 = None   # HBM bytes per launch of the roofline family: PMC passes cannot run inside bench.py, so the
        tpath = os.path.join(REPO, 'profiles', 'r01_traffic_pmc.json')       # committed rocprofv3 --pmc result is attached
        if dom is c16 and os.path.exists(tpath):
            tj = json.load(open(tpath))
            # raw FETCH_SIZE + WRITE_SIZE; the guide's gfx950 x2 correction applies to 16-byte-per-lane reads only (here: the
            # weight slabs, which mostly hit L2), the activation patches are read 4 bytes per lane -> raw is the estimate, x2 the bound
            traffic, traffic_x2 = tj['traffic_bytes_per_launch_raw'], tj['traffic_bytes_per_launch_fetch_x2']
        roofline = {'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'frac': achieved / peak, 'traffic': traffic,
                    'traffic_fetch_x2_upper_bound': traffic_x2,
                    'algorithmic_bytes_per_launch': dom['bytes'] / max(dom['launches'], 1),
                    'kernel': name,
                    'note': 'achieved = algorithmic (fp32-equivalent) conv flops / HIP-event time of the family; for bf16x3 the '
                            'hardware executes 3 bf16 MFMA flops per algorithmic flop, so peak = 2500/3 TFLOP/s',
                    'launches_per_step': dom['launches'] / args.steps,
                    'algorithmic_gflop_per_step': dom['flops'] / args.steps / 1e9,
                    'avg_launch_ms': dom['ms'] / max(dom['launches'], 1),
                    'all_conv_tflops': (c16['flops'] + c32['flops'] + prof['conv1x1_bf16x3']['flops']) /
                                       ((c16['ms'] + c32['ms'] + prof['conv1x1_bf16x3']['ms']) * 1e-3) / 1e12,
                    'family_ms_per_step': {k: round(p['ms'] / args.steps, 4) for k, p in prof.items()}}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        from next3d_amd import layers
        precision_dtype = 'bf16x3 (split-bf16 operands, f32 accumulate; f32 elsewhere)' if layers.PRECISION == 'bf16x3' else 'f32'
        frames = args.steps * B * world
        print(json.dumps({
            'metric': 'generator fwd frames/sec at 512² (64³ vol, 96 samples)', 'value': frames / elapsed, 'unit': 'frames/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': precision_dtype, 'data': 'synthetic',
            'config': {'workload': f'BASELINE.json configs[1]: batch={B} seeds per GPU, 512² output, 64² neural render, '
                                   '48 coarse + 48 importance samples, trunc=0.7, demo.obj mesh, mapping+synthesis, '
                                   'seeded synthetic weights (172.8M params)', 'batch_per_gpu': B, 'seed_sharded': True,
                       'gather': 'RCCL gather of uint8 frames to rank 0' if world > 1 else 'none'},
            'frames_bitwise_reproducible': reproducible, 'roofline': roofline, 'cpu_baseline': cpu}))
        if not reproducible:
            print('bench.py: pipelined steps returned different frames', file=sys.stderr)
    drain()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
