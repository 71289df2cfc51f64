"""Seed / frame sharding across the GPUs of one node (one process per GPU).

Every (seed, camera, mesh-frame) triple of the generator forward is independent, so the path shards with
no data-path collective: rank r renders items r, r+W, r+2W, ... (configs 3/5: frames k mod W) or a contiguous block
(configs 2/4: seeds).  The only collective is the final gather of finished uint8 frames to rank 0 (RCCL over xGMI on the
GPU box — backend 'nccl'; 'gloo' in the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_block(items, rank, world):
    """Contiguous block partition (seed-sharded batches); sizes differ by at most one."""
    n = len(items)
    base, extra = divmod(n, world)
    start = rank * base + min(rank, extra)
    return list(items[start:start + base + (1 if rank < extra else 0)])


def shard_strided(items, rank, world):
    """Round-robin partition (frame k -> rank k mod world), the order video frames are produced in."""
    return list(items[rank::world])


def gather_frames(frames, dst=0, group=None):
    """Gather equally-shaped uint8 frame batches [B,3,H,W] to `dst`; returns [world*B,3,H,W] on dst, None elsewhere."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return frames
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    out = [torch.empty_like(frames) for _ in range(world)] if rank == dst else None
    dist.gather(frames, out, dst=dst, group=group)
    return torch.cat(out, 0) if rank == dst else None


class AsyncFrameGather:
    """The per-step frame gather of a video / benchmark loop, off the critical path: `submit(frames)` starts an asynchronous
    gather to `dst` (RCCL runs it on its own stream while the next step computes) after waiting for the previous one, so the
    receive buffers on `dst` are never the target of two collectives at once; `drain()` waits for the last one.
    `received` (on dst, after drain / the next submit) = list of the ranks' frame batches of the most recent completed step."""

    def __init__(self, like, dst=0, group=None, single_rank_collective=False):
        """single_rank_collective: with one rank, still issue the gather when a process group exists (a 1-rank RCCL gather is a
        valid collective): lets a 1-GPU box execute the N > 1 code path — process-group init, gather on RCCL's stream, drain."""
        self.dst, self.group = dst, group
        init = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if init else 1
        self.active = init and (self.world > 1 or single_rank_collective)
        self.rank = dist.get_rank(group) if self.active else 0
        self.received = [torch.empty_like(like) for _ in range(self.world)] if (self.active and self.rank == dst) else None
        self._pending = None

    def submit(self, frames):
        if not self.active:
            return
        self.drain()
        work = dist.gather(frames, self.received, dst=self.dst, group=self.group, async_op=True)
        self._pending = (work, frames)                     # `frames` stays referenced until the collective is done

    def drain(self):
        if self._pending is not None:
            self._pending[0].wait()
            self._pending = None


def unshard_strided(gathered, world):
    """Invert shard_strided on a gathered [world*B, ...] tensor: frame k = gathered[(k % world) * B + k // world]."""
    b = gathered.shape[0] // world
    idx = torch.arange(world * b)
    return gathered[(idx % world) * b + idx // world]


def pin_rank_to_cores(local_rank, local_world, cores=None):
    """One process per GPU: restrict rank `local_rank` (of `local_world` on this node) to its own contiguous slice of the cores
    this process may run on — block r of `local_world` equal blocks — and size torch's intra-op pool to it.  Contiguous core ids
    are one NUMA domain's on the usual numbering, so a rank's launch thread, its pinned staging buffers and its GPU stay together;
    an explicit `cores` list overrides the slice.  Returns the core list in effect (unchanged affinity on platforms without
    sched_setaffinity or when the slice would be empty)."""
    import os
    if not hasattr(os, 'sched_setaffinity'):
        return None
    avail = sorted(os.sched_getaffinity(0))
    if cores is None:
        per = len(avail) // max(1, local_world)
        if per < 1:
            return avail
        cores = avail[local_rank * per:(local_rank + 1) * per]
    try:
        os.sched_setaffinity(0, set(cores))
    except OSError:
        return avail
    torch.set_num_threads(max(1, min(len(cores), 16)))
    return list(cores)
