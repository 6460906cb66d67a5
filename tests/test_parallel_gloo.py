"""world_size-2 gloo test of the data-parallel evaluation plumbing (runs on CPU).  The per-rank
"model" is a deterministic function of the clip index, so the gathered result must equal the
single-process result for the full clip set."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from adafocus_amd.parallel import gather_logits, gather_variable, shard_range


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 64, 65, 1000):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            for (a0, a1), (b0, b1) in zip(spans, spans[1:]):
                assert a1 == b0 and a1 >= a0
            assert max(e - s for s, e in spans) - min(e - s for s, e in spans) <= 1


def _fake_logits(idx):
    return torch.stack([torch.sin(idx.float() * (k + 1)) for k in range(5)], dim=1)


def _worker(rank, world, port, n_clips, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s, e = shard_range(n_clips, rank, world)
    local = _fake_logits(torch.arange(s, e))
    if n_clips % world == 0:
        full = gather_logits(local)
    else:
        full = gather_variable(local)
    ok = torch.equal(full, _fake_logits(torch.arange(n_clips)))
    ret[rank] = bool(ok)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(n_clips):
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n_clips, ret), nprocs=world, join=True)
    assert all(ret[r] for r in range(world)), dict(ret)


def test_gather_logits_world2_even():
    _run(64)


def test_gather_variable_world2_ragged():
    _run(33)


def test_numa_binding_helper_is_harmless_without_a_gpu():
    from adafocus_amd import parallel as P
    assert P._cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11} and P._cpulist("") == set()
    before = os.sched_getaffinity(0)
    assert P.bind_to_gpu_numa(0) is None or isinstance(P.bind_to_gpu_numa(0), int)
    if not torch.cuda.is_available():
        assert os.sched_getaffinity(0) == before
    # on a CPU tensor the side-stream gather degrades to the blocking one
    x = torch.arange(6.0).view(2, 3)
    assert torch.equal(P.gather_logits_async(x, None), x)
