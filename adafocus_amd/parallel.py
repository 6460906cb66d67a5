"""Data-parallel offline evaluation: one process per GPU, clips sharded by index, ONE small
collective per batch.

The reference never shards evaluation (its val loader has no sampler, every rank evaluates the
whole set -- ACT/main_dist.py:239; STH/evaluate.py is single-GPU), so this is new functionality
with no reference call site.  Clips are independent in eval mode (BN uses
running statistics, GRU state is per clip), hence no data-path exchange: each rank owns full
weight replicas (~192 MB fp32) and the only traffic is an all-gather of the (B_local, classes)
fp32 logits -- 51 KB per rank per batch at B=64, C=200, latency-bound on xGMI.  `backend="nccl"`
is RCCL on ROCm; the CPU tests use gloo.
"""
import torch
import torch.distributed as dist

__all__ = ["shard_range", "gather_logits", "gather_logits_async", "gather_variable", "bind_to_gpu_numa"]


def shard_range(n_items, rank, world):
    """Contiguous block partition of range(n_items): returns (start, stop) for `rank`.
    The first n_items % world ranks get one extra item; concatenating the shards in rank order
    restores the original order."""
    base, extra = divmod(int(n_items), int(world))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def gather_logits(local, group=None):
    """All-gather equally sized (B_local, C) logits into (world*B_local, C), rank-major."""
    if not (dist.is_available() and dist.is_initialized()):
        return local
    world = dist.get_world_size(group)
    if world == 1:
        return local
    local = local.contiguous()
    out = torch.empty((world * local.shape[0],) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
    dist.all_gather_into_tensor(out, local, group=group)
    return out


def gather_logits_async(local, comm_stream, group=None):
    """gather_logits issued on `comm_stream`, ordered after whatever the CURRENT stream has enqueued (the GRU scan that
    produced `local`): the collective is ~50 KB per rank and latency-bound on xGMI, so it runs beside the next batch's
    trunk instead of in front of it.  The result belongs to `comm_stream`: consume it after
    `torch.cuda.current_stream().wait_stream(comm_stream)` or a device synchronisation.  On a CPU tensor (gloo tests) or with
    comm_stream=None it is the plain blocking gather."""
    if comm_stream is None or not local.is_cuda:
        return gather_logits(local, group)
    cur = torch.cuda.current_stream(local.device)
    comm_stream.wait_stream(cur)
    with torch.cuda.stream(comm_stream):
        local.record_stream(comm_stream)
        return gather_logits(local, group)


def _cpulist(text):
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def bind_to_gpu_numa(device_index):
    """Pin this process (one rank per GPU) to the CPUs of the NUMA node its GPU hangs off: the rank's Python thread, its
    staging threads and the pinned host buffers they touch stay next to the device (8 ranks on a 2-socket host otherwise
    share whatever cores the scheduler picks).  Reads the node from sysfs via the device's PCI address; returns the node
    number, or None when it cannot be determined (no sysfs entry, node -1, not Linux) -- in which case nothing changes."""
    import os
    try:
        pr = torch.cuda.get_device_properties(device_index)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        if node < 0:
            return None
        cpus = _cpulist(open("/sys/devices/system/node/node%d/cpulist" % node).read())
        cpus &= os.sched_getaffinity(0)
        if not cpus:
            return None
        os.sched_setaffinity(0, cpus)
        return node
    except Exception:
        return None


def gather_variable(local, group=None):
    """All-gather of ragged shards (last batch of an epoch): pads to the largest shard, gathers,
    trims.  Returns the concatenation in rank order."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    n = torch.tensor([local.shape[0]], device=local.device, dtype=torch.int64)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes)
    pad = torch.zeros((m,) + tuple(local.shape[1:]), device=local.device, dtype=local.dtype)
    pad[:local.shape[0]] = local
    full = gather_logits(pad, group)
    return torch.cat([full[r * m:r * m + sizes[r]] for r in range(world)], dim=0)
