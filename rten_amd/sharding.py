"""Batch sharding across the GPUs of one node: one process per GPU, `torch.distributed`
with the "nccl" backend (= RCCL over xGMI on ROCm).

Every op on the path is per-sample independent, so the path shards along the batch with NO data-path
collective.  The only collective is a one-time broadcast of the prepacked weight arena from rank 0 at
load (ResNet-50: ~102 MB f32), so that only one rank pays the host->device upload and prepack.
DynamicQuantizeLinear statistics are per tensor INCLUDING the batch dimension (src/ops/quantize.rs:397-419),
so an int8 run sharded G ways equals G independent reference runs on the shards -- `shard_range` defines
exactly which images those are.
"""
from __future__ import annotations


def shard_range(global_batch: int, rank: int, world_size: int) -> range:
    """Contiguous batch slice [start, end) owned by `rank`; the first `global_batch % world_size` ranks get
    one extra item (ragged batches are allowed, an empty range is allowed)."""
    if world_size <= 0 or not (0 <= rank < world_size) or global_batch < 0:
        raise ValueError("invalid shard request")
    base, extra = divmod(global_batch, world_size)
    start = rank * base + min(rank, extra)
    return range(start, start + base + (1 if rank < extra else 0))


def broadcast_weight_arena(arena, src: int = 0):
    """Broadcast the weight arena (a 1-D uint8 torch tensor) from `src` to every rank, in place."""
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return arena
    dist.broadcast(arena, src=src)
    return arena


def gather_outputs(local_out, global_batch: int):
    """All-gather per-rank output slices (torch tensors, possibly ragged along dim 0) into the full batch
    on every rank.  Used by harnesses that want the whole batch's logits; not on the timed path."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local_out
    world = dist.get_world_size()
    sizes = [len(shard_range(global_batch, r, world)) for r in range(world)]
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(local_out.shape[1:]), dtype=local_out.dtype, device=local_out.device)
    pad[: local_out.shape[0]] = local_out
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[:n] for b, n in zip(bufs, sizes)], dim=0)
