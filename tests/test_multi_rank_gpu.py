"""The N > 1 code paths on ONE GPU: two ranks (processes) share cuda:0 and talk over gloo -- RCCL refuses two ranks on one
device, and gpurun leases single-GPU boxes, so this is as close to the 8-GPU launch as a test can get here: real HIP kernels
in every rank, sharded evaluation with a ragged last batch, device tensors through gather_variable."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

N_CLIPS, T, BS = 13, 4, 4


def _setup():
    from adafocus_amd import synth
    from adafocus_amd.gfv_net import GFV
    from bench_extras import act_args, synth_model_state
    args = act_args(T, 96, BS)
    m = GFV(args).eval()
    m.load_state_dict(synth_model_state(m, 1007), strict=True)
    gen = np.random.Generator(np.random.PCG64([41, 7]))
    clips = torch.from_numpy(gen.integers(0, 256, size=(N_CLIPS, 224, 224, T * 3), dtype=np.uint8))
    labels = torch.from_numpy(gen.integers(0, 200, size=(N_CLIPS, 1)).astype(np.int64))

    class DS:
        def __len__(self):
            return N_CLIPS

        def __getitem__(self, i):
            return clips[i], labels[i]
    return m, args, DS(), synth


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from adafocus_amd import evaluate as E
    from adafocus_amd.parallel import bind_to_gpu_numa
    bind_to_gpu_numa(0)                      # must be harmless wherever it runs
    m, args, ds, _ = _setup()
    m = m.to("cuda:0")
    out = E.validate(ds, m, torch.nn.CrossEntropyLoss(), args, rank=rank, world=world, batch_size=BS, quiet=True)
    ret[rank] = out[:3]
    dist.barrier()
    dist.destroy_process_group()


def test_validate_two_ranks_sharing_the_gpu_ragged_last_batch():
    """13 uint8 clips, batches of 4: rank 0 gets 7 clips (4 + 3), rank 1 gets 6 (4 + 2).  Both ranks must report the metrics of
    the single-process run over all 13 clips (logits and targets all-gathered from device tensors of different lengths)."""
    from adafocus_amd import evaluate as E
    m, args, ds, _ = _setup()
    m = m.to("cuda:0")
    single = E.validate(ds, m, torch.nn.CrossEntropyLoss(), args, batch_size=BS, quiet=True)
    del m
    torch.cuda.empty_cache()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    for r in range(2):
        assert np.allclose(ret[r], single[:3], atol=1e-3), (r, ret[r], single[:3])
