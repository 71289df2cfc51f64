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


def unshard_strided(gathered, world):
    """Invert shard_strided on a gathered [world*B, ...] tensor: frame k = gathered[(k % world) * B + k // world]."""
    b = gathered.shape[0] // world
    idx = torch.arange(world * b)
    return gathered[(idx % world) * b + idx // world]
