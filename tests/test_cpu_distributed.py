"""world_size-2 `gloo` test of the N>1 path: seed/frame sharding + the final frame gather (the only collective)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from next3d_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    seeds = list(range(6))
    mine = sharding.shard_strided(seeds, rank, world)
    frames = torch.stack([torch.full((3, 4, 4), s, dtype=torch.uint8) for s in mine])      # "rendered" frame = its seed
    g = sharding.gather_frames(frames, dst=0)
    if rank == 0:
        ordered = sharding.unshard_strided(g, world)
        out.put(ordered[:, 0, 0, 0].tolist())
    else:
        assert g is None
    # the benchmark / video loop pattern: one asynchronous gather per step, overlapped with the next step
    ag = sharding.AsyncFrameGather(frames, dst=0)
    seen = []
    for step in range(4):
        ag.submit(frames + 10 * step)
        if rank == 0 and step > 0:
            pass                                            # (the previous step's frames are complete after this submit's drain)
    ag.drain()
    if rank == 0:
        assert [int(t[0, 0, 0, 0]) for t in ag.received] == [30 + r for r in range(world)]
    else:
        assert ag.received is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_helpers():
    assert sharding.shard_block(list(range(10)), 0, 4) == [0, 1, 2] and sharding.shard_block(list(range(10)), 3, 4) == [8, 9]
    assert sum((sharding.shard_block(list(range(10)), r, 4) for r in range(4)), []) == list(range(10))
    assert sharding.shard_strided(list(range(7)), 1, 3) == [1, 4]
    assert sharding.shard_block([], 0, 2) == []
    t = torch.zeros(2, 3, 4, 4, dtype=torch.uint8)
    assert sharding.gather_frames(t) is t                      # single process: no collective


@pytest.mark.timeout(120)
def test_two_rank_gather_gloo():
    ctx = mp.get_context('spawn')
    q = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(100)
        assert p.exitcode == 0
    assert q.get() == [0, 1, 2, 3, 4, 5]


@pytest.mark.timeout(180)
def test_bench_self_spawns_for_multi_gpu():
    """`python bench.py --gpus 2` (no launcher, the way the driver runs --gpus 1) must start one process per GPU itself
    (reference: train_next3d.py:100-103 spawns its own ranks).  The CPU stand-in (--selftest-spawn: gloo, tiny frames) goes
    through the same self-spawn -> torch.distributed.run -> rendezvous -> AsyncFrameGather path as the GPU benchmark."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(repo, 'bench.py'), '--gpus', '2', '--selftest-spawn'], capture_output=True, text=True,
                       env=env, timeout=170)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    out = json.loads(line)
    assert out['n_gpus'] == 2 and out['ok'] and out['gathered_last_step'] == [2, 12]
