#!/usr/bin/env python3
"""bench.py — generator-forward frames/s of the Next3D hot path on MI355X (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B]          (N > 1: spawns one process per GPU itself)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W                            (the driver's launch line; same result)

One "step" = one pass of the hot path over one batch: `G.mapping` + `G.synthesis` for B=4 seeds per GPU at
512² output, 64² neural render, 48 coarse + 48 importance depth samples, trunc 0.7, demo mesh (BASELINE.json
configs[1]); inputs are resident in HBM before the timed region.  With N GPUs every rank renders its own B seeds
(independent seeds shard with no data-path collective — weak scaling) and the finished uint8 frames are gathered to
rank 0 over RCCL (north_star: "RCCL over xGMI only for the final gather"); the gather of step k runs on RCCL's stream
while step k+1 computes and is joined before step k+2 is enqueued.
Rank 0 prints ONE JSON line.  Besides the contract's fields it carries (N = 1 only, all measured by this process):
  single_stream  the same K steps issued in order on ONE stream (`value` pipelines consecutive steps over --lanes streams)
  roofline       the dominant kernel family (3x3 split-bf16 convolutions, bf16 matrix cores) timed with HIP events on the
                 launch stream, against 2500/3 TFLOP/s; `traffic` from the committed rocprofv3 --pmc profile of this build
  roofline_fp32  the same K steps with N3D_PRECISION=fp32 arithmetic (v_mfma_f32_32x32x2_f32 everywhere): frames/s and the
                 conv family against 157.3 TFLOP/s
  config3        gen_videos_next3d.py's 2x2-grid, 120-frame camera orbit over a fixed mesh (BASELINE.json configs[2])
  sr_fp16_mode   the same K steps with the reference's default float16 super-resolution blocks (no force_fp32): frames/s, speed-up over
                 the float32 route, `roofline_f16` (the f16 3x3 kernels against the 2.5 PFLOP/s dense f16 peak)
  fp16_backbones_mode  the generator as legacy.load_network_pkl(force_fp16=True) builds it (float16 blocks in the four backbones as well)
  config1        BASELINE.json configs[0]'s shape as a latency figure (batch 1), eager launches vs HIP-graph replay
  config1b       the scripts' true call pattern (gen_samples / gen_videos call G.synthesis one frame at a time): batch 1 at the metric's
                 512² / 64² / 48+48 — single-frame latency eager and from a HIP graph, and frames/s with requests pipelined over the lanes
  b1_route       the OPERATOR-boundary route an un-reloaded pickle takes (oracle/b1_route.py: the reference's code pattern on
                 next3d_amd.torch_utils.ops + shims; fused modulated convolutions as groups = batch calls): frames/s on the same workload
  config5        reenact_avatar_next3d.py's loop: one identity, a new FLAME mesh per frame (configs[4], synthetic sequence)
  cpu_baseline   the CPU oracle (a port of the reference's fp32 path) timed on the host cores.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_L2_TBPS = 34.5                 # aggregate L2 bandwidth, /opt/skills/guides/MI355X_MICROARCH.md (L2 per XCD)
PEAK_FP32_MFMA_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md (dense fp32 matrix peak)
PEAK_BF16_MFMA_TFLOPS = 2500.0         # dense bf16 matrix peak (same guide); bf16x3 spends 3 bf16 MFMAs per algorithmic MAC
TRAFFIC_PROFILE = os.path.join(REPO, "profiles", "r06_traffic_pmc.json")
CPU_REFERENCE_PROFILE = os.path.join(REPO, 'profiles', 'r03_cpu_reference.json')
# what the bf16 matrix pipe sustains with the kernels' instruction mix and RANDOM operands (tools/mfma_peak.hip, profiles/r02_mfma_peak_probe.txt:
# 1850-1950 TFLOP/s bf16 = 617-650 fp32-equivalent; the chip power-limits to ~1.8 GHz under this load)
MEASURED_BF16X3_CEILING_TFLOPS = 633.0
MEASURED_F16_CEILING_TFLOPS = 1700.0          # profiles/r04_mfma_peak_f16_probe.txt: v_mfma_f32_32x32x16_f16, random operands, registers only (1560-1640 with the kernels' LDS reads)
CONV_FAMILY = ('3x3 split-bf16 conv family: every conv2d*_bf16x3 kernel launched by n3d_conv2d_bf16x3 with ksize 3 '
               '(stride 1 incl. the persistent and pre-split variants, transposed stride 2, stride 2; all launches of the step)')


def cpu_baseline(seconds_budget=25.0):
    """The oracle (kind='port': bit-exact restatement of the reference's CPU fp32 path, oracle/pin_against_reference.py)
    on the host cores: N=1 frames of the same workload (R=64, 48+48), bounded sample."""
    from next3d_amd import demo, mesh, spec
    from oracle import cases, generator as ogen
    host = os.cpu_count() or 1
    threads = min(host, 64)                 # torch's intra-op pool stops scaling well before that on this workload
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
    return {'value': n / dt, 'unit': 'frames/s', 'cores': threads, 'host_cores': host, 'kind': 'port',
            'sample': f'{n} frames, batch 1, 512²/64²/48+48, fp32, torch-CPU oracle with {threads} threads '
                      f'(host reports {host} logical cores)'}


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def self_spawn(argv, n):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one process per GPU (what
    train_next3d.py:100-103 does with torch.multiprocessing.spawn); rank 0's JSON line is the child's stdout."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    return subprocess.call(cmd, env=env)


def spawn_selftest():
    """--selftest-spawn (CPU, gloo): the launch / rendezvous / gather plumbing of the multi-GPU path with a tiny stand-in
    for the frames — tests/test_cpu_distributed.py runs `python bench.py --gpus 2 --selftest-spawn`."""
    import torch.distributed as dist
    from next3d_amd.sharding import AsyncFrameGather
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group('gloo')
    gat = AsyncFrameGather(torch.empty(2, 3, 4, 4, dtype=torch.uint8), dst=0)
    for step in range(3):
        gat.submit(torch.full((2, 3, 4, 4), 10 * rank + step, dtype=torch.uint8))
    gat.drain()
    dist.barrier()
    if rank == 0:
        got = [int(t[0, 0, 0, 0]) for t in gat.received]
        print(json.dumps({'selftest': 'spawn', 'n_gpus': world, 'gathered_last_step': got, 'ok': got == [10 * r + 2 for r in range(world)]}))
    dist.destroy_process_group()


def orbit_cameras(frames, device):
    """gen_videos_next3d.py:133-137: yaw / pitch walk of the camera over `frames` frames -> [frames, 25]."""
    from next3d_amd import demo
    out = []
    for k in range(frames):
        az = 3.14 / 2 + 0.35 * math.sin(2 * 3.14 * k / (frames // 2))                 # the script's own 3.14
        po = 3.14 / 2 - 0.05 + 0.25 * math.cos(2 * 3.14 * k / (frames // 2))
        out.append(demo.camera_label(az - math.pi / 2, po - math.pi / 2, focal=4.2647))       # gen_videos_next3d.py:97
    return torch.cat(out, 0).to(device)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=4, help='seeds per GPU per step (BASELINE.json configs[1]: 4)')
    ap.add_argument('--prewarm-seconds', type=float, default=2.0, help='un-counted steps before the warm-up (clock ramp)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-roofline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip roofline_fp32 / config3 / config5')
    ap.add_argument('--serial-gather', action='store_true', help='join the frame gather of step k before step k+1 is enqueued')
    ap.add_argument('--lanes', type=int, default=3, help='HIP streams the steps are issued on in turn (1 = one stream, in order): '
                    'consecutive steps are independent batches, so step k+1 may start while step k still runs')
    ap.add_argument('--sr-fp16', action='store_true', help="time the scripts' default route (float16 super-resolution blocks) in the main loop instead of "
                    'force_fp32=True (SURVEY 8d config 2) — for profiling that route; the headline stays the float32 route')
    ap.add_argument('--selftest-spawn', action='store_true', help=argparse.SUPPRESS)
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(self_spawn(sys.argv[1:], args.gpus))
    if args.selftest_spawn:
        return spawn_selftest()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    args.gpus = world
    if world > 1:
        # one process per GPU: pin each rank to its own slice of the host's cores (launch threads of N ranks otherwise migrate over all
        # sockets; the GPUs of a node hang off different NUMA domains, and a rank's ~600 launches per step are host-latency sensitive)
        from next3d_amd.sharding import pin_rank_to_cores
        pin_rank_to_cores(local_rank, int(os.environ.get('LOCAL_WORLD_SIZE', world)))
    assert torch.cuda.is_available(), 'bench.py needs a HIP device'
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    import torch.distributed as dist
    # N3D_BENCH_FORCE_DIST=1 (tests/test_generator_gpu.py::test_bench_multi_gpu_path_on_one_gpu): run the N > 1 code path — RCCL process
    # group, per-step asynchronous frame gather, barrier — with a single rank, so that it executes on hardware on a 1-GPU box
    force_dist = os.environ.get('N3D_BENCH_FORCE_DIST', '0') == '1' and 'RANK' in os.environ
    use_dist = world > 1 or force_dist
    if use_dist:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=dev)        # 'nccl' == RCCL on ROCm

    from next3d_amd import _lib, demo, layers
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
    gatherer = AsyncFrameGather(torch.empty(B, 3, 512, 512, dtype=torch.uint8, device=dev), dst=0, single_rank_collective=force_dist)

    def to_frames(img):
        frames = torch.empty(img.shape, dtype=torch.uint8, device=img.device)     # gen_samples_next3d.py:201 (NCHW kept)
        _lib.check(_lib.lib().n3d_to_uint8(_lib.ptr(img), _lib.ptr(frames), img.numel(), _lib.stream()))
        return frames

    # Consecutive steps are independent batches (different seeds in a real run): they are issued on `--lanes` HIP streams in
    # turn, so the latency-bound low-resolution layers of step k+1 (a handful of workgroups each) fill the chip while the
    # MFMA-bound layers of step k run, exactly as a serving loop would pipeline requests.  Every step's whole work is inside
    # the timed region (all lanes are joined before the clock stops).
    lanes = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.lanes))]
    for s_ in lanes:
        s_.wait_stream(torch.cuda.current_stream())
    counter, one_lane, sr_fp32, gen, draw_rng = [0], [False], [not args.sr_fp16], [G], [False]

    def step():
        if args.serial_gather:
            gatherer.drain()
        lane = lanes[counter[0] % (1 if one_lane[0] else len(lanes))]
        counter[0] += 1
        with torch.cuda.stream(lane):
            ws = gen[0].mapping(z, c_cond, truncation_psi=0.7, truncation_cutoff=14)
            # the renderer's two random inputs are INJECTED in the timed loop (the reference draws them per call with torch.rand, vr/renderer.py:205,252:
            # two launches); `rng_default_path` on the line times the same steps with the draws inside the call
            rnd = dict() if draw_rng[0] else dict(depth_jitter=jitter, importance_u=u)
            img = gen[0].synthesis(ws, c, v, neural_rendering_resolution=R, noise_mode='const', force_fp32=sr_fp32[0], **rnd)['image']   # the fp32 path (the one the goldens pin)
            frames = to_frames(img)
            gatherer.submit(frames)      # joins the PREVIOUS step's gather, then starts this one (runs under the next step's compute)
        return frames

    def sync():
        gatherer.drain()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(fn, steps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        torch.cuda.synchronize()
        return time.perf_counter() - t0

    # un-counted pre-warm: the first launches page the code objects in and the chip ramps its clocks; a 0.3 s timed region
    # right after a cold start measured 11 % low on a fresh box (round 1)
    t_pre = time.perf_counter()
    step()
    while time.perf_counter() - t_pre < args.prewarm_seconds:
        step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    per_rank = [args.steps * B / elapsed]
    rccl = None
    if use_dist:
        mine = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(every, mine)                               # each rank's own clock: per-rank frames/s on the line (did RCCL see N ranks?)
        per_rank = [args.steps * B / float(e.item()) for e in every]
        t = mine.clone()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        try:
            ver = '.'.join(str(x) for x in torch.cuda.nccl.version())
        except Exception:                                           # noqa: BLE001
            ver = None
        rccl = {'rccl_world_size': dist.get_world_size(), 'backend': dist.get_backend(), 'rccl_version': ver,
                'per_rank_frames_per_s': [round(x, 2) for x in per_rank]}

    # the inputs do not change between steps, so pipelined steps must return identical frames — and the frames of a step issued ALONE
    # on one stream with nothing else in flight: a cheap guard against stream races in exactly the configuration that was timed
    # (tests/test_generator_gpu.py has the per-stage version)
    # the SAME timed loop twice more inside this run: the boxes of the pool differ by +-4 % and a 0.2-0.3 s region has its own run-to-run
    # spread — `value` stays the first (contractual) loop, `ms_per_step_repeats` shows min / median of the three
    repeats = [1e3 * elapsed / args.steps]
    if world == 1 and not args.no_extras:
        for _ in range(2):
            sync()
            tr = time.perf_counter()
            for _ in range(args.steps):
                step()
            sync()
            repeats.append(1e3 * (time.perf_counter() - tr) / args.steps)
    fa, fb, fc = step(), step(), step()
    sync()
    one_lane[0] = True
    f1 = step()
    sync()
    one_lane[0] = False
    reproducible = bool(torch.equal(fa, fb) and torch.equal(fb, fc) and torch.equal(fc, f1))

    def conv_profile(steps):
        """`steps` steps with per-launch HIP events on the launch stream (outside the timed region; the static-backbone side
        stream is switched off for this pass so that every kernel's event time is its own duration) -> n3d_prof_read()."""
        overlap, G.overlap_static = G.overlap_static, False
        one_lane[0] = True                                      # ... and one launch stream
        torch.cuda.synchronize()
        step(); torch.cuda.synchronize()
        _lib.prof_reset()
        _lib.prof_enable(True)
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        _lib.prof_enable(False)
        G.overlap_static, one_lane[0] = overlap, False
        prof = _lib.prof_read()
        _lib.prof_reset()
        return prof

    single = world == 1
    single_stream = None
    if single and len(lanes) > 1 and not args.no_extras:          # the same K steps on ONE stream (in order), for reference
        one_lane[0] = True
        step(); torch.cuda.synchronize()
        t1 = timed(step, args.steps)
        one_lane[0] = False
        single_stream = {'value': args.steps * B / t1, 'unit': 'frames/s', 'ms_per_step': 1e3 * t1 / args.steps,
                         'note': 'steps issued in order on one HIP stream'}
    roofline = None
    if not args.no_roofline and single:                           # (N > 1: the timed steps only — no extra passes on any rank)
        prof = conv_profile(args.steps)
        dom, peak = prof['conv2d_bf16x3'], PEAK_BF16_MFMA_TFLOPS / 3.0
        name = CONV_FAMILY
        if layers.PRECISION != 'bf16x3':
            dom, peak, name = prof['conv2d'], PEAK_FP32_MFMA_TFLOPS, 'conv2d_mfma_kernel family (fp32 MFMA; all launches of the step)'
        achieved = dom['flops'] / (dom['ms'] * 1e-3) / 1e12 if dom['ms'] > 0 else 0.0
        traffic = traffic_raw = traffic_src = traffic_cal = None   # HBM bytes per launch: PMC passes cannot run inside bench.py -> the committed
        if layers.PRECISION == 'bf16x3' and os.path.exists(TRAFFIC_PROFILE):     # rocprofv3 --pmc result of THIS build is attached
            tj = json.load(open(TRAFFIC_PROFILE))
            traffic_raw = tj['traffic_bytes_per_launch_raw']
            traffic = tj.get('traffic_bytes_per_launch_calibrated', traffic_raw)        # FETCH_SIZE / WRITE_SIZE divided by the factors measured on known byte counts
            traffic_cal = {k: {'fetch_counter_per_byte': v['fetch_counter_per_byte'], 'write_counter_per_byte': v['write_counter_per_byte']}
                           for k, v in tj.get('calibration', {}).items()} or None
            traffic_src = 'profiles/' + os.path.basename(TRAFFIC_PROFILE)
        convs = ('conv2d_bf16x3', 'conv2d', 'conv1x1_bf16x3')
        roofline = {'bound': 'mfma', 'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'frac': achieved / peak,
                    'frac_vs_measured_ceiling': (achieved / MEASURED_BF16X3_CEILING_TFLOPS) if layers.PRECISION == 'bf16x3' else None,
                    'measured_ceiling': {'value': MEASURED_BF16X3_CEILING_TFLOPS, 'unit': 'TFLOP/s fp32-equivalent', 'source': 'profiles/r02_mfma_peak_probe.txt (tools/mfma_peak.hip: '
                                         'v_mfma_f32_32x32x16_bf16 with random operands and the kernels\' fragment-read mix sustains 1850-1950 TFLOP/s: the chip power-limits to ~1.8 GHz)'},
                    'traffic': traffic, 'traffic_raw_counters': traffic_raw, 'traffic_calibration': traffic_cal, 'traffic_source': traffic_src,
                    'algorithmic_bytes_per_launch': dom['bytes'] / max(dom['launches'], 1),
                    'kernel': name,
                    'note': 'achieved = algorithmic (fp32-equivalent) conv flops / HIP-event time of the family; for bf16x3 the '
                            'hardware executes 3 bf16 MFMA flops per algorithmic flop, so peak = 2500/3 TFLOP/s',
                    'launches_per_step': dom['launches'] / args.steps,
                    'algorithmic_gflop_per_step': dom['flops'] / args.steps / 1e9,
                    'avg_launch_ms': dom['ms'] / max(dom['launches'], 1),
                    'all_conv_tflops': sum(prof[k]['flops'] for k in convs) / (sum(prof[k]['ms'] for k in convs) * 1e-3) / 1e12,
                    'family_ms_per_step': {k: round(p['ms'] / args.steps, 4) for k, p in prof.items()}}
        # ---- the volume renderer (n3d_render_rays_ex: depth-bounds pre-pass + render_rays_kernel) against what bounds it: the texel gathers.
        # Algorithmic bytes = 12 texels x 128 B per sample point (3 planes x 4 bilinear taps x 32 float32 channels; nothing is shared between
        # points in the accounting) over the HIP-event time of the entry point.  The planes (25 MB per sample) are L2 / Infinity-Cache resident:
        # the bound is the L2 -> L1 -> register gather path, not HBM.  PMC of this kernel: profiles/r06_render_pmc_final.txt.
        rr = prof.get('render_rays')
        if rr and rr['ms'] > 0:
            tb = rr['bytes'] / (rr['ms'] * 1e-3) / 1e12
            roofline['render_rays'] = {'bound': 'l2 gather', 'achieved': tb, 'peak': PEAK_L2_TBPS, 'unit': 'TB/s', 'frac': tb / PEAK_L2_TBPS,
                                       'ms_per_launch': rr['ms'] / max(rr['launches'], 1), 'algorithmic_bytes_per_launch': rr['bytes'] / max(rr['launches'], 1),
                                       'decoder_tflops': rr['flops'] / (rr['ms'] * 1e-3) / 1e12,
                                       'note': 'achieved = 12 texels x 128 B per sample point / HIP-event time of n3d_render_rays_ex (batch of rays: N x 64 x 64 x (48 + 48)); '
                                               'peak = aggregate L2 bandwidth of /opt/skills/guides/MI355X_MICROARCH.md (34.5 TB/s); measured L1 / L2 request rates and the '
                                               'reason the kernel sits where it does: profiles/r06_render_pmc_final.txt, DESIGN.md 3.2'}

    extras = {}
    frames_total, elapsed_total = args.steps * B * world, elapsed
    precision0 = layers.PRECISION

    def run_extras():
        """Every figure below is optional: main() calls this under try / except — a failure is reported on the line ('extras_error'), it never costs the headline."""
        # ---- the same K steps with the renderer's random inputs drawn INSIDE the call (torch.rand on the device, as the reference does)
        draw_rng[0] = True
        step(); step(); torch.cuda.synchronize()
        t_rng = timed(step, args.steps)
        draw_rng[0] = False
        extras['rng_default_path'] = {'value': args.steps * B / t_rng, 'unit': 'frames/s', 'ms_per_step': 1e3 * t_rng / args.steps,
                                      'note': 'depth jitter / importance u drawn per call by torch.rand (vr/renderer.py:205,252) instead of injected: the headline injects them'}
        kw = dict(neural_rendering_resolution=R, noise_mode='const', depth_jitter=jitter, importance_u=u, force_fp32=True)
        # ---- strict-fp32 arithmetic (N3D_PRECISION=fp32): the same K steps on v_mfma_f32_32x32x2_f32
        if layers.PRECISION == 'bf16x3':
            layers.set_precision('fp32')
            for _ in range(len(lanes) + 1):       # every lane once (its allocator pool, this mode's kernels), then one more
                step()
            torch.cuda.synchronize()
            k32 = max(3, args.steps // 2)
            t32 = timed(step, k32)
            p32 = conv_profile(k32)['conv2d']
            a32 = p32['flops'] / (p32['ms'] * 1e-3) / 1e12 if p32['ms'] > 0 else 0.0
            extras['roofline_fp32'] = {'value': k32 * B / t32, 'unit': 'frames/s', 'ms_per_step': 1e3 * t32 / k32, 'steps': k32, 'dtype': 'f32',
                                       'bound': 'mfma', 'achieved': a32, 'peak': PEAK_FP32_MFMA_TFLOPS, 'frac': a32 / PEAK_FP32_MFMA_TFLOPS,
                                       'kernel': 'conv2d_mfma_kernel family (v_mfma_f32_32x32x2_f32, bit-equivalent to an fmaf chain)',
                                       'algorithmic_gflop_per_step': p32['flops'] / k32 / 1e9}
            layers.set_precision('bf16x3')
            step(); torch.cuda.synchronize()
        # ---- the reference's DEFAULT super-resolution mode (sr_num_fp16_res = 4, no force_fp32: float16 storage in the SR blocks,
        # superresolution.py:210-217) on the same workload
        sr_fp32[0] = False
        for _ in range(len(lanes) + 1):       # every lane once (its allocator pool, this mode's kernels), then one more
            step()
        torch.cuda.synchronize()
        k16 = max(3, args.steps // 2)
        t16 = timed(step, k16)
        p16 = conv_profile(k16)
        f16 = p16['conv2d_f16']
        a16 = f16['flops'] / (f16['ms'] * 1e-3) / 1e12 if f16['ms'] > 0 else 0.0
        extras['sr_fp16_mode'] = {'value': k16 * B / t16, 'unit': 'frames/s', 'ms_per_step': 1e3 * t16 / k16, 'steps': k16,
                                  'speedup_vs_fp32_route': (k16 * B / t16) / (frames_total / elapsed_total),
                                  'note': "synthesis(..., force_fp32=False): the scripts' default route — float16 super-resolution blocks on v_mfma_f32_32x32x16_f16 "
                                          '(float16 operands, float32 accumulation: the arithmetic of the reference\'s half convolutions)',
                                  'roofline_f16': {'bound': 'mfma', 'achieved': a16, 'peak': PEAK_BF16_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': a16 / PEAK_BF16_MFMA_TFLOPS,
                                                   'kernel': 'conv2d_h8_f16_kernel / conv2d_up_h8_f16_*_kernel (n3d_conv2d_f16: the four 3x3 convolutions of the two float16 blocks)',
                                                   'launches_per_step': f16['launches'] / k16, 'algorithmic_gflop_per_step': f16['flops'] / k16 / 1e9,
                                                   'avg_launch_ms': f16['ms'] / max(f16['launches'], 1),
                                                   'algorithmic_bytes_per_launch': f16['bytes'] / max(f16['launches'], 1),
                                                   'frac_vs_measured_ceiling': a16 / MEASURED_F16_CEILING_TFLOPS,
                                                   'measured_ceiling': {'value': MEASURED_F16_CEILING_TFLOPS, 'unit': 'TFLOP/s', 'source': 'profiles/r04_mfma_peak_f16_probe.txt '
                                                                        '(tools/mfma_peak_f16.hip: the f16 MFMA with random operands sustains 1700-1720 TFLOP/s from registers, 1560-1640 with the '
                                                                        "kernels' fragment reads: the chip clocks 1.5-1.64 GHz under this load)"}},
                                  'family_ms_per_step': {k: round(pv['ms'] / k16, 4) for k, pv in p16.items()}}
        # ---- legacy.load_network_pkl(force_fp16=True) (legacy.py:49-59): num_fp16_res = 4, conv_clamp = 256 in all four backbones too —
        # every block of resolution >= 32 of the five networks runs as a float16 block on the f16 matrix cores
        G16, _ = demo.build_generator(dev, force_fp16=True)
        gen[0] = G16
        for _ in range(len(lanes) + 1):       # every lane once (its allocator pool, this mode's kernels), then one more
            step()
        torch.cuda.synchronize()
        tbb = timed(step, k16)
        pbb = conv_profile(k16)
        fbb = pbb['conv2d_f16']
        abb = fbb['flops'] / (fbb['ms'] * 1e-3) / 1e12 if fbb['ms'] > 0 else 0.0
        extras['fp16_backbones_mode'] = {'value': k16 * B / tbb, 'unit': 'frames/s', 'ms_per_step': 1e3 * tbb / k16, 'steps': k16,
                                         'speedup_vs_fp32_route': (k16 * B / tbb) / (frames_total / elapsed_total),
                                         'note': 'the generator as legacy.load_network_pkl(force_fp16=True) builds it (num_fp16_res = 4, conv_clamp = 256 in the four '
                                                 'backbones, float16 super-resolution blocks): every block of resolution >= 32 on v_mfma_f32_32x32x16_f16',
                                         'roofline_f16': {'bound': 'mfma', 'achieved': abb, 'peak': PEAK_BF16_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': abb / PEAK_BF16_MFMA_TFLOPS,
                                                          'launches_per_step': fbb['launches'] / k16, 'algorithmic_gflop_per_step': fbb['flops'] / k16 / 1e9,
                                                          'avg_launch_ms': fbb['ms'] / max(fbb['launches'], 1), 'frac_vs_measured_ceiling': abb / MEASURED_F16_CEILING_TFLOPS},
                                         'family_ms_per_step': {k: round(pv['ms'] / k16, 4) for k, pv in pbb.items()}}
        gen[0] = G
        del G16
        sr_fp32[0] = True
        step(); torch.cuda.synchronize()
        # ---- configs[2]: 2x2 grid (batch 4 = one video frame), 120-frame orbit, fixed mesh (gen_videos_next3d.py:126-158)
        grid_seeds = [10720, 12374, 13393, 17099]                       # README.md:48
        zg, _, cg_cond, vg = demo.demo_batch(grid_seeds[:B], device=dev)
        wsg = G.mapping(zg, cg_cond, truncation_psi=0.7, truncation_cutoff=14)
        cams = orbit_cameras(120, dev)
        cam_of = lambda k: cams[k % 120:k % 120 + 1].expand(B, -1).contiguous()
        it = [0]

        def orbit(cached, kw_, graph=False):
            k = it[0]; it[0] += 1
            fn = G.synthesis_graph if graph else G.synthesis
            to_frames(fn(wsg, cam_of(k), vg, use_cached_backbone=cached, **kw_)['image'])
        G.synthesis(wsg, cam_of(0), vg, cache_backbone=True, **kw); torch.cuda.synchronize()
        t_c = timed(lambda: orbit(True, kw), 120)
        def graph_timed(fn, steps):
            """An optional figure: a failed capture must not cost the benchmark line."""
            try:
                fn(); torch.cuda.synchronize()                          # capture
                return timed(fn, steps)
            except Exception as e:                                      # noqa: BLE001
                print(f'bench.py: HIP-graph leg skipped ({type(e).__name__}: {e})', file=sys.stderr)
                torch.cuda.synchronize()
                return float('nan')
        t_cg = graph_timed(lambda: orbit(True, kw, graph=True), 120)    # the same 120 frames replayed from ONE captured HIP graph
        t_u = timed(lambda: orbit(False, kw), 24)
        G.rendering_kwargs['depth_resolution'], G.rendering_kwargs['depth_resolution_importance'] = 96, 96     # sampling_multiplier 2
        j2 = torch.rand((B, R * R, 96, 1), device=dev, generator=g)
        u2 = torch.rand((B * R * R, 96), device=dev, generator=g)
        kw2 = dict(kw, depth_jitter=j2, importance_u=u2)
        orbit(True, kw2); torch.cuda.synchronize()
        t_c2 = timed(lambda: orbit(True, kw2), 60)
        G.rendering_kwargs['depth_resolution'], G.rendering_kwargs['depth_resolution_importance'] = Sc, Sf
        extras['config3'] = {'workload': 'gen_videos_next3d.py: 2x2 grid (batch 4 = one video frame), 120-frame camera orbit, fixed FLAME '
                                         'mesh, uint8 frames; images/s = 4 x video frames/s',
                             'cached_planes_images_per_s': 120 * B / t_c, 'cached_planes_video_fps': 120 / t_c,
                             'cached_planes_hip_graph_images_per_s': 120 * B / t_cg,
                             'uncached_images_per_s': 24 * B / t_u, 'uncached_video_fps': 24 / t_u,
                             'cached_planes_96+96_images_per_s': 60 * B / t_c2, 'unit': 'frames/s', 'frames_timed': [120, 24, 60]}
        # ---- configs[0] as a latency figure: batch 1, 32x32 render, 24 + 24 samples, one frame at a time (eager launches vs graph replay)
        z1, c1, c1_cond, v1 = demo.demo_batch([0], device=dev)
        ws1 = G.mapping(z1, c1_cond, truncation_psi=0.7, truncation_cutoff=14)
        G.rendering_kwargs['depth_resolution'], G.rendering_kwargs['depth_resolution_importance'] = 24, 24
        j1 = torch.rand((1, 32 * 32, 24, 1), device=dev, generator=g)
        u1 = torch.rand((32 * 32, 24), device=dev, generator=g)
        kw1 = dict(neural_rendering_resolution=32, noise_mode='const', depth_jitter=j1, importance_u=u1, force_fp32=True)
        one = lambda fn: to_frames(fn(ws1, c1, v1, **kw1)['image'])
        one(G.synthesis); torch.cuda.synchronize()
        t_1e = timed(lambda: one(G.synthesis), 60)
        t_1g = graph_timed(lambda: one(G.synthesis_graph), 60)
        G.rendering_kwargs['depth_resolution'], G.rendering_kwargs['depth_resolution_importance'] = Sc, Sf
        extras['config1'] = {'workload': 'BASELINE.json configs[0] shape on the GPU: batch 1, 512² output, 32² neural render, 24 + 24 samples, frames issued back to back',
                             'config1': 'shape only — the product has no CPU fallback (BASELINE.json configs[0] is the reference\'s --force-fp32 CPU custom-ops plumbing run; '
                                        'here a missing library or a CPU tensor raises, and the only CPU restatement is the oracle, timed as cpu_baseline)',
                             'eager_ms_per_frame': 1e3 * t_1e / 60, 'hip_graph_ms_per_frame': 1e3 * t_1g / 60, 'frames_timed': 60}
        # ---- the scripts' TRUE call pattern: gen_samples_next3d.py:165-201 / gen_videos_next3d.py:131-158 call G.synthesis one frame at a
        # time (batch 1) — at the metric's 512² / 64² / 48 + 48: latency of one frame (eager, HIP graph) and frames/s with independent
        # requests pipelined over the lanes (a serving loop; each request = mapping + synthesis + uint8 conversion)
        j1b = torch.rand((1, R * R, Sc, 1), device=dev, generator=g)
        u1b = torch.rand((R * R, Sf), device=dev, generator=g)
        kw1b = dict(neural_rendering_resolution=R, noise_mode='const', depth_jitter=j1b, importance_u=u1b, force_fp32=True)
        one_b = lambda fn: to_frames(fn(ws1, c1, v1, **kw1b)['image'])
        one_b(G.synthesis); one_b(G.synthesis); torch.cuda.synchronize()
        t_1be = timed(lambda: one_b(G.synthesis), 60)
        t_1bg = graph_timed(lambda: one_b(G.synthesis_graph), 60)
        req = [0]

        def request():
            lane = lanes[req[0] % len(lanes)]; req[0] += 1
            with torch.cuda.stream(lane):
                w_ = G.mapping(z1, c1_cond, truncation_psi=0.7, truncation_cutoff=14)
                return to_frames(G.synthesis(w_, c1, v1, **kw1b)['image'])
        for s_ in lanes:
            s_.wait_stream(torch.cuda.current_stream())
        request(); request(); request(); torch.cuda.synchronize()
        t_1bp = timed(request, 120)
        # ... and the same requests replayed from one captured HIP graph PER LANE (synthesis_graph(graph_slot=k) on stream k): eager
        # issue costs ~4 ms of host time per frame (160 ctypes launches) — one Python thread cannot feed three lanes that way
        greq = [0]

        def request_graph():
            k = greq[0] % len(lanes); greq[0] += 1
            with torch.cuda.stream(lanes[k]):
                return to_frames(G.synthesis_graph(ws1, c1, v1, graph_slot=k, **kw1b)['image'])

        def lanes_graph_timed():
            try:
                for _ in range(2 * len(lanes)):
                    request_graph()
                torch.cuda.synchronize()
                return timed(request_graph, 240)
            except Exception as e:                                      # noqa: BLE001
                print(f'bench.py: pipelined HIP-graph leg skipped ({type(e).__name__}: {e})', file=sys.stderr)
                torch.cuda.synchronize()
                return float('nan')
        t_1bpg = lanes_graph_timed()
        # ... and on the route the scripts actually take (no force_fp32: float16 super-resolution blocks; gen_videos_next3d.py:155)
        kw1d = {k: val for k, val in kw1b.items() if k != 'force_fp32'}
        one_d = lambda fn: to_frames(fn(ws1, c1, v1, **kw1d)['image'])
        one_d(G.synthesis); one_d(G.synthesis); torch.cuda.synchronize()
        t_1de = timed(lambda: one_d(G.synthesis), 60)
        t_1dg = graph_timed(lambda: one_d(G.synthesis_graph), 60)

        def request_d():
            lane = lanes[req[0] % len(lanes)]; req[0] += 1
            with torch.cuda.stream(lane):
                w_ = G.mapping(z1, c1_cond, truncation_psi=0.7, truncation_cutoff=14)
                return to_frames(G.synthesis(w_, c1, v1, **kw1d)['image'])
        request_d(); request_d(); request_d(); torch.cuda.synchronize()
        t_1dp = timed(request_d, 120)
        extras['config1b'] = {
                              'default_route': {'eager_ms_per_frame': 1e3 * t_1de / 60, 'hip_graph_ms_per_frame': 1e3 * t_1dg / 60, 'pipelined_frames_per_s': 120 / t_1dp,
                                                'note': "no force_fp32: the scripts' own call (float16 super-resolution blocks on the f16 matrix cores)"},'workload': 'batch 1 (one G.synthesis call per frame, as gen_samples_next3d.py / gen_videos_next3d.py issue them), 512² output, '
                                          '64² neural render, 48 + 48 samples, force_fp32=True',
                              'eager_ms_per_frame': 1e3 * t_1be / 60, 'hip_graph_ms_per_frame': 1e3 * t_1bg / 60,
                              'pipelined_frames_per_s': 120 / t_1bp, 'pipelined_hip_graph_frames_per_s': 240 / t_1bpg, 'lanes': len(lanes),
                              'frames_timed': [60, 60, 120, 240], 'unit': 'ms / frames/s',
                              'note': 'pipelined: independent single-frame requests round-robin over the lanes; eager issue is bound by the one Python '
                                      'thread (~160 launches per frame), the HIP-graph form (one captured graph per lane) by the GPU'}
        # ---- the OPERATOR-boundary route (B1): what the scripts' default `--reload_modules False` executes — the reference's own network
        # code on next3d_amd.torch_utils.ops + shims.  /root/reference does not exist here: oracle/b1_route.py re-instantiates the
        # oracle's restatement of that code on the operator layer (test infrastructure standing in for the pickled modules)
        try:
            from oracle import b1_route
            from next3d_amd import mesh as _mesh, spec as _spec
            dd = demo.demo_arrays()
            Pb = _spec.synthetic_state_dict(0)
            Pb.update(_mesh.mesh_buffers(dd['faces'], dd['uvs'], dd['uvfaces']))
            route = b1_route.Route(dev)
            Pb = route.to_device(Pb)
            maskb = torch.nn.functional.interpolate(_mesh.synthetic_uv_face_mask().float(), [256, 256]).to(dev)
            rkb = dict(demo.RENDERING_KWARGS)
            zf = z.float()

            def b1_step(fp32):
                with torch.no_grad():
                    w_ = route.mapping(Pb, zf, c_cond, rkb, truncation_psi=0.7, truncation_cutoff=14)
                    return to_frames(route.synthesis(Pb, w_, c, v, maskb, rkb, jitter, u, neural_rendering_resolution=R, force_fp32=fp32)['image'])
            res = {}
            for name, fp32 in (('force_fp32', True), ('default_fp16_sr', False)):
                fb1 = b1_step(fp32); b1_step(fp32); torch.cuda.synchronize()
                kb = max(3, args.steps // 5)
                tb = timed(lambda: b1_step(fp32), kb)
                res[name] = {'value': kb * B / tb, 'unit': 'frames/s', 'ms_per_step': 1e3 * tb / kb, 'steps': kb}
                if fp32:
                    res[name]['max_abs_uint8_diff_vs_model_boundary'] = int((fb1.int() - fa.int()).abs().max())
            b2_one = single_stream['value'] if single_stream else frames_total / elapsed_total
            extras['b1_route'] = {'workload': 'the same batch (BASELINE.json configs[1]) through the operator boundary: reference code pattern (fused modulated '
                                              'convolutions as groups = batch calls, torch weight modulation, separate bias_act / upfirdn2d / noise ops, the '
                                              "reference's torch renderer, fill_mouth / gen_mouth_mask host round trips) on next3d_amd.torch_utils.ops + shims; one stream",
                                  **res, 'fraction_of_model_boundary_single_stream': res['force_fp32']['value'] / b2_one,
                                  'driver': 'oracle/b1_route.py (test infrastructure standing in for the un-reloaded pickle)',
                                  'caller': 'oracle restatement of the reference modules (the pickled reference code is not on the GPU box): the oracle CALLS the product\'s '
                                            'operator layer here, it is not what is measured'}
            del route, Pb
        except Exception as e:                                      # noqa: BLE001  (an optional figure must not cost the benchmark line)
            print(f'bench.py: b1_route leg skipped ({type(e).__name__}: {e})', file=sys.stderr)
            extras['b1_route'] = {'error': f'{type(e).__name__}: {e}'}
        # ---- configs[4]: reenactment — one identity (ws), a NEW mesh per frame (reenact_avatar_next3d.py:139-164); data/obama is
        # not in the tree: demo mesh + seeded smooth per-frame perturbation (sigma 1 mm), 4 consecutive frames per step
        zr, cr, cr_cond, vr = demo.demo_batch([0] * B, yaws=[0.0] * B, device=dev)
        wsr = G.mapping(zr, cr_cond, truncation_psi=0.7, truncation_cutoff=14)
        gq = torch.Generator(device=dev).manual_seed(7)
        F_ = 64
        drift = torch.cumsum(torch.randn(F_ + B, 1, 3, device=dev, generator=gq) * 2e-4, 0) + torch.randn(F_ + B, vr.shape[1], 3, device=dev, generator=gq) * 1e-4
        meshes = vr[:1] + drift
        G.synthesis(wsr, cr, meshes[:B].contiguous(), cache_identity=True, **kw); torch.cuda.synchronize()
        it[0] = 0

        def reenact(cached):
            k = it[0] % (F_ // B); it[0] += 1
            to_frames(G.synthesis(wsr, cr, meshes[k * B:(k + 1) * B].contiguous(), use_cached_identity=cached, **kw)['image'])
        t_rc = timed(lambda: reenact(True), 32)
        t_ru = timed(lambda: reenact(False), 16)

        def reenact_graph():
            k = it[0] % (F_ // B); it[0] += 1
            to_frames(G.synthesis_graph(wsr, cr, meshes[k * B:(k + 1) * B].contiguous(), use_cached_identity=True, **kw)['image'])
        t_rg = graph_timed(reenact_graph, 32)
        extras['config5'] = {'workload': 'reenact_avatar_next3d.py loop: one identity, a new FLAME mesh + landmarks per frame (synthetic '
                                         'smooth sequence; data/obama is not in the tree), 4 consecutive frames per step, camera fixed',
                             'cached_identity_frames_per_s': 32 * B / t_rc, 'cached_identity_hip_graph_frames_per_s': 32 * B / t_rg,
                             'uncached_frames_per_s': 16 * B / t_ru, 'unit': 'frames/s',
                             'frames_timed': [32 * B, 16 * B]}

    if single and not args.no_extras:
        try:
            run_extras()
        except Exception as e:                                          # noqa: BLE001
            print(f'bench.py: extras stopped ({type(e).__name__}: {e})', file=sys.stderr)
            extras['extras_error'] = f'{type(e).__name__}: {e}'
            try:
                torch.cuda.synchronize()
            except Exception:                                           # noqa: BLE001
                pass
        finally:                                                        # (the legs switch these; nothing below depends on them, but leave the process as it was found)
            draw_rng[0] = False
            sr_fp32[0] = not args.sr_fp16
            gen[0] = G
            if layers.PRECISION != precision0:
                layers.set_precision(precision0)

    cpu = None
    if rank == 0 and single and not args.no_cpu_baseline:
        cpu = cpu_baseline()
        if os.path.exists(CPU_REFERENCE_PROFILE):      # the reference's OWN Python beside the port, timed where /root/reference exists (the build container)
            cr = json.load(open(CPU_REFERENCE_PROFILE))
            case = cr['cases'].get('case_r64_s48_b4', {})
            cpu['reference_python'] = {'frames_per_s': case.get('reference_frames_per_s'), 'port_frames_per_s_same_host': case.get('oracle_frames_per_s'),
                                       'cores': cr.get('cores'), 'sample': 'batch 4, 512²/64²/48+48, fp32: mapping + synthesis of the reference\'s own modules (ops -> its _ref '
                                       'implementations; rasteriser / flood fill stand-ins) and of the port on the SAME inputs, same container',
                                       'source': 'profiles/' + os.path.basename(CPU_REFERENCE_PROFILE) + ' (oracle/pin_against_reference.py)'}

    if rank == 0:
        precision_dtype = 'bf16x3 (split-bf16 operands, f32 accumulate; f32 elsewhere)' if layers.PRECISION == 'bf16x3' else 'f32'
        frames = args.steps * B * world
        print(json.dumps({
            'metric': 'generator fwd frames/sec at 512² (64³ vol, 96 samples)', 'value': frames / elapsed, 'unit': 'frames/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
            'ms_per_step_repeats': {'values': [round(x, 4) for x in repeats], 'min': round(min(repeats), 4), 'median': round(sorted(repeats)[len(repeats) // 2], 4),
                                    'note': 'the timed loop of `steps` steps, three times in this run (the first is `value`); boxes of the pool differ by +-4 %'},
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': precision_dtype, 'data': 'synthetic',
            'config': {'route': 'default (float16 super-resolution blocks)' if args.sr_fp16 else 'force_fp32=True (SURVEY 8d config 2; the reference-generated goldens pin it)',
                       'workload': f'BASELINE.json configs[1]: batch={B} seeds per GPU, 512² output, 64² neural render, '
                                   '48 coarse + 48 importance samples, trunc=0.7, demo.obj mesh, mapping+synthesis, '
                                   'seeded synthetic weights (172.8M params)', 'batch_per_gpu': B, 'seed_sharded': True,
                       'streams': f'{len(lanes)} (consecutive steps alternate between them)',
                       'gather': 'RCCL gather of uint8 frames to rank 0, overlapped with the next step' if use_dist else 'none',
                       'prewarm_seconds': args.prewarm_seconds},
            'rng_inputs': 'depth_jitter / importance_u injected (resident tensors); rng_default_path = the same steps with the draws inside the call',
            'multi_gpu': rccl if rccl is not None else {'rccl_world_size': 1, 'per_rank_frames_per_s': [round(per_rank[0], 2)], 'note': 'single process, no process group'},
            'frames_bitwise_reproducible': reproducible, 'single_stream': single_stream, 'roofline': roofline, **extras,
            'cpu_baseline': cpu}))
        if not reproducible:
            print('bench.py: pipelined steps returned different frames', file=sys.stderr)
    gatherer.drain()
    if use_dist:
        if rank == 0 and gatherer.received is not None and args.no_roofline and args.no_extras:       # (the last step submitted was `fc`) what arrived over RCCL is what was rendered
            assert torch.equal(gatherer.received[0], fc), 'gathered frames differ from the rendered ones'
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
