#!/usr/bin/env python3
"""Two (or more) ranks sharing ONE GPU over gloo: checks the rank != 0 path of the batch-sharded deployment.

Rank 0 prepacks the weights into its arena and broadcasts it; every other rank only receives the arena.  All ranks then
run the SAME batch and must produce logits bit-identical to rank 0's.  (RCCL refuses two ranks per device, so this uses
gloo; the collective call -- one dist.broadcast of the arena -- is the same.)

    RTEN_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \\
        --master-port 29555 tools/check_multi_rank.py
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rten_amd import lib  # noqa: E402
from rten_amd.workloads import resnet50  # noqa: E402
from rten_amd.sharding import broadcast_weight_arena  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev)
    dist.init_process_group(backend=os.environ.get("RTEN_DIST_BACKEND", "gloo"), rank=rank, world_size=world)
    batch = 4
    ctx = lib.Context(dev)
    weights = resnet50.make_weights()
    probe = resnet50.ResNet50(ctx, batch, weights)
    nbytes = probe.arena_bytes
    del probe
    arena_t = torch.zeros(nbytes, dtype=torch.uint8, device=f"cuda:{dev}")
    net = resnet50.ResNet50(ctx, batch, weights, arena_ptr=arena_t.data_ptr(), arena_keepalive=arena_t)
    if rank == 0:
        net.upload_weights()
    ctx.sync()
    broadcast_weight_arena(arena_t, src=0)
    torch.cuda.synchronize()
    x = np.random.default_rng(1234).random((batch, 3, 224, 224), dtype=np.float32)
    net.x.upload(x)
    net.forward()
    ctx.sync()
    logits = torch.from_numpy(net.logits.numpy().copy())
    ref = logits.clone()
    dist.broadcast(ref, src=0)
    same = bool(torch.equal(logits.view(torch.int32), ref.view(torch.int32)))
    finite = bool(torch.isfinite(logits).all()) and float(logits.abs().max()) > 0
    flags = torch.tensor([int(same and finite)], dtype=torch.int32)
    dist.all_reduce(flags, op=dist.ReduceOp.MIN)
    if rank == 0:
        print(f"multi-rank check: world={world} identical_logits={bool(flags.item())}")
    dist.destroy_process_group()
    sys.exit(0 if flags.item() == 1 else 1)


if __name__ == "__main__":
    main()
