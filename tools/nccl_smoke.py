#!/usr/bin/env python3
"""RCCL sanity check of the data-parallel plumbing on however many GPUs torchrun gives us (works with 1)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from adafocus_amd.parallel import gather_logits, gather_variable, shard_range  # noqa: E402

rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
s, e = shard_range(37, rank, world)
local_logits = torch.arange(s, e, device=dev, dtype=torch.float32)[:, None].repeat(1, 5)
full = gather_variable(local_logits)
assert torch.equal(full[:, 0].cpu(), torch.arange(37, dtype=torch.float32)), full[:, 0]
eq = gather_logits(torch.full((4, 3), float(rank), device=dev))
assert eq.shape == (4 * world, 3)
out = torch.empty((world * 4, 3), device=dev)
dist.all_gather_into_tensor(out, torch.full((4, 3), float(rank), device=dev))   # the raw collective, even at world = 1
torch.cuda.synchronize()
dist.barrier()
if rank == 0:
    print("nccl smoke ok: world=%d" % world)
dist.destroy_process_group()
