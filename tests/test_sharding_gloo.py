"""World-size-2 `gloo` test (CPU) of the N>1 path: batch sharding + weight-arena broadcast + output gather.
The GPU kernels are not involved; this covers the distributed plumbing bench.py uses at --gpus N."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rten_amd.sharding import broadcast_weight_arena, gather_outputs, shard_range


def test_shard_range_partitions_batch():
    for gb in (0, 1, 7, 32, 256, 257):
        for world in (1, 2, 3, 4, 8):
            got = [i for r in range(world) for i in shard_range(gb, r, world)]
            assert got == list(range(gb))
            sizes = [len(shard_range(gb, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    assert list(shard_range(256, 3, 8)) == list(range(96, 128))  # BASELINE config 5: 8 shards x 32
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # rank 0 "prepacks" the arena, others receive it
        arena = torch.zeros(1 << 16, dtype=torch.uint8)
        if rank == 0:
            arena = torch.from_numpy(np.random.default_rng(0).integers(0, 255, 1 << 16, dtype=np.uint8))
        broadcast_weight_arena(arena, src=0)
        want = np.random.default_rng(0).integers(0, 255, 1 << 16, dtype=np.uint8)
        ok_arena = bool((arena.numpy() == want).all())
        # each rank "infers" its shard of a ragged global batch; outputs = f(global index)
        gb = 7
        mine = shard_range(gb, rank, world)
        local = torch.tensor([[float(i), float(i) * 2] for i in mine], dtype=torch.float32).reshape(len(mine), 2)
        full = gather_outputs(local, gb)
        ok_gather = full.shape == (gb, 2) and bool((full[:, 0] == torch.arange(gb, dtype=torch.float32)).all())
        # max-over-ranks timing reduction used by bench.py
        t = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        q.put((rank, ok_arena, ok_gather, float(t.item())))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_broadcast_and_gather():
    world, port = 2, _free_port()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_arena, ok_gather, tmax in res:
        assert ok_arena and ok_gather and tmax == 2.0
