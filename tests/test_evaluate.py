"""Row f3: metric helpers and the sharded validate loop (CPU).  Metrics are pinned to vectors from the
reference's own accuracy / cal_map (G9); the loop is exercised with a stand-in model on 2 gloo ranks."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from adafocus_amd import evaluate as E
from oracle import ref_metrics as RM
from tests.helpers import golden


def _g9_inputs():
    gen9 = np.random.Generator(np.random.PCG64([99, 0xC0]))
    lg = torch.from_numpy(gen9.standard_normal((300, 20)).astype(np.float32) * 2)
    tg = torch.from_numpy(gen9.integers(0, 20, size=(300, 1)).astype(np.int64))
    lg[torch.arange(300), tg[:, 0]] += 1.5
    tg2 = torch.cat([tg, torch.full((300, 1), -1, dtype=torch.int64)], 1)
    tg2[::7, 1] = (tg2[::7, 0] + 3) % 20
    return lg, tg, tg2


def test_metrics_match_reference_vectors():
    g = golden("g9_metrics")
    lg, tg, tg2 = _g9_inputs()
    a1, a5 = E.accuracy(lg, tg[:, 0], topk=(1, 5))
    assert np.allclose(a1.numpy(), g["acc1"]) and np.allclose(a5.numpy(), g["acc5"])
    m_ap, ap = E.cal_map(lg, tg)
    assert abs(float(m_ap) - float(g["mAP"][0])) < 1e-4 and np.allclose(ap.numpy(), g["ap"], atol=1e-4)
    m_ap2, ap2 = E.cal_map(lg, tg2)
    assert abs(float(m_ap2) - float(g["mAP_multi"][0])) < 1e-4 and np.allclose(ap2.numpy(), g["ap_multi"], atol=1e-4)
    # the numpy oracle agrees with both
    o1, o5 = RM.accuracy(lg.numpy(), tg[:, 0].numpy(), (1, 5))
    assert abs(o1 - float(g["acc1"][0])) < 1e-4 and abs(o5 - float(g["acc5"][0])) < 1e-4
    om, oap = RM.cal_map(lg.numpy(), tg2.numpy())
    assert abs(om - float(g["mAP_multi"][0])) < 1e-3 and np.allclose(oap, g["ap_multi"], atol=1e-3)


def test_meters_format():
    m = E.AverageMeter("Acc@1", ":6.2f")
    m.update(50.0, 2)
    m.update(100.0, 2)
    assert str(m) == "Acc@1 100.00 ( 75.00)"
    p = E.ProgressMeter(12, m, prefix="Test: ")
    assert p.print(3, quiet=True) == "Test: [ 3/12]\tAcc@1 100.00 ( 75.00)\n"


class _Args:
    num_segments, num_classes, batch_size, gpu, dataset = 4, 10, 8, None, "actnet"


class _FakeModel(torch.nn.Module):
    """Deterministic stand-in with the GFV call signature: logits depend only on the clip content."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.linspace(-1, 1, 10)[None, :], requires_grad=False)

    def forward(self, **kw):
        x = kw["input"]
        b = x.shape[0]
        s = x.reshape(b, -1).mean(1, keepdim=True)
        last = torch.sin(s * 37.0 + self.w * 5.0)
        return last.repeat_interleave(_Args.num_segments, 0), last


class _Data:
    def __init__(self, n):
        g = torch.Generator().manual_seed(5)
        self.x = torch.randn(n, 12, 4, 4, generator=g)
        self.y = torch.randint(0, 10, (n, 2), generator=g)

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, i):
        return self.x[i], self.y[i]


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = E.validate(_Data(37), _FakeModel(), torch.nn.CrossEntropyLoss(), _Args(), rank=rank, world=world, quiet=True)
    ret[rank] = out[:3]
    dist.barrier()
    dist.destroy_process_group()


def test_validate_sharded_equals_single_process():
    single = E.validate(_Data(37), _FakeModel(), torch.nn.CrossEntropyLoss(), _Args(), quiet=True)
    assert single[3][-1].startswith(" * Acc@1")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    for r in range(2):
        assert np.allclose(ret[r], single[:3], atol=1e-4), (ret[r], single[:3])
