"""Row f3: metric helpers and the sharded validate loop (CPU).  Metrics are pinned to vectors from the
reference's own accuracy / cal_map (G9); the loop is exercised with a stand-in model on 2 gloo ranks."""
import os
import socket

import numpy as np
import pytest
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


class _Stage2Args(_Args):
    train_stage = 2


class _FakeStage2(_FakeModel):
    """The per-step surface of the stage-2 validation branch (ACT/main_dist.py:343-366): the prediction sharpens with every step and depends
    on the steps seen since restart_batch."""

    def __init__(self):
        super().__init__()
        self.calls = []

    def glance(self, scan):
        b = scan.shape[0]
        t = scan.shape[1] // 3
        return scan.reshape(b, t, 3, 4, 4), scan.reshape(b, t, -1).mean(2, keepdim=True)

    def one_step_act(self, img, fmap, fvec, restart_batch=False, training=True):
        assert not training and img.shape[1:] == (3, 4, 4) and fmap.shape[1:] == (3, 4, 4) and fvec.shape[1:] == (1,)
        if restart_batch:
            self.acc, self.n = torch.zeros_like(fvec), 0
        self.calls.append(bool(restart_batch))
        self.acc, self.n = self.acc + img.reshape(img.shape[0], -1).mean(1, keepdim=True), self.n + 1
        logits = torch.sin(self.acc / self.n * 37.0 + self.w * 5.0) * self.n
        return logits, logits, None, torch.zeros(img.shape[0], 2), torch.zeros_like(logits)


def test_validate_stage2_walks_the_mdp_step_by_step():
    """args.train_stage = 2 (ACT/main_dist.py:343-366, 411-418): glance once, one_step_act per frame with restart_batch on the first, metrics
    from the last step's prediction, a 'mAP @ time step' line per step -- against a hand-rolled loop over the same stand-in."""
    data, model, args = _Data(21), _FakeStage2(), _Stage2Args()
    top1, top5, m_ap, logs = E.validate(data, model, torch.nn.CrossEntropyLoss(), args, quiet=True)
    nb = (21 + args.batch_size - 1) // args.batch_size
    assert model.calls == ([True] + [False] * (args.num_segments - 1)) * nb
    ref = _FakeStage2()
    fm, fv = ref.glance(data.x)
    fr = data.x.view(21, args.num_segments, 3, 4, 4)
    steps = [ref.one_step_act(fr[:, s], fm[:, s], fv[:, s], restart_batch=(s == 0), training=False)[1] for s in range(args.num_segments)]
    acc1, acc5 = E.accuracy(steps[-1], data.y[:, 0], topk=(1, 5))
    want_map, _ = E.cal_map(steps[-1], data.y[:, 0:1])
    assert abs(top1 - acc1[0].item()) < 1e-4 and abs(top5 - acc5[0].item()) < 1e-4 and abs(m_ap - float(want_map)) < 1e-4
    lines = [ln for ln in logs if ln.startswith("mAP @ time step")]
    assert len(lines) == args.num_segments
    for i, ln in enumerate(lines):
        want, _ = E.cal_map(steps[i], data.y[:, 0:1])
        assert ln == "mAP @ time step {step}: {mAP:.5f}\n".format(mAP=float(want), step=i)

    class _S4(_Args):
        train_stage = 4
    with pytest.raises(NotImplementedError):
        E.validate(data, model, torch.nn.CrossEntropyLoss(), _S4(), quiet=True)


def test_validate_stages_0_and_1_call_the_reference_forms():
    """train_stage 1: model(input=, scan=, training=False, backbone_pred=False, one_step=False) (ACT/main_dist.py:334-340); train_stage 0:
    model(input=scan, scan=None, glancer=args.pretrain_glancer, backbone_pred=True, one_step=False).mean(1) (:372-376)."""
    seen = []

    class M(_FakeModel):
        def forward(self, **kw):
            seen.append({k: (v if not torch.is_tensor(v) else tuple(v.shape)) for k, v in kw.items()})
            if kw["backbone_pred"]:
                b = kw["input"].shape[0]
                return super().forward(input=kw["input"])[1][:, None, :].expand(b, _Args.num_segments, 10)
            return super().forward(**kw)

    class A1(_Args):
        train_stage, consensus = 1, "gru"

    class A0(_Args):
        train_stage, pretrain_glancer = 0, True
    data = _Data(9)
    want = E.validate(data, _FakeModel(), torch.nn.CrossEntropyLoss(), _Args(), quiet=True)[:3]
    r1 = E.validate(data, M(), torch.nn.CrossEntropyLoss(), A1(), quiet=True)[:3]
    assert seen[0]["one_step"] is False and seen[0]["backbone_pred"] is False and seen[0]["training"] is False and seen[0]["scan"] == seen[0]["input"]
    assert r1 == pytest.approx(want, abs=1e-4)
    del seen[:]
    r0 = E.validate(data, M(), torch.nn.CrossEntropyLoss(), A0(), quiet=True)[:3]
    assert seen[0]["backbone_pred"] is True and seen[0]["glancer"] is True and seen[0]["scan"] is None and seen[0]["one_step"] is False
    assert r0 == pytest.approx(want, abs=1e-4)


# ---------------------------------------------------------------------------------------------------------------------
# Something-Something loop (STH/evaluate.py:165-226)
class _SthArgs:
    num_segments_glancer, num_segments_focuser, num_classes, batch_size, gpu = 4, 4, 10, 8, None
    video_div, glance_size, patch_size = 2, 4, 2


class _FakeSth(torch.nn.Module):
    """Stand-in with the STH GFV surface: logits depend on the clip content and on how many focusing steps have been taken;
    the baseline logits are a fixed function of the clip, so rewards are deterministic too."""

    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.linspace(-1, 1, 10)[None, :], requires_grad=False)
        self.calls = []

    def glance(self, g):
        b = g.shape[0]
        return g.reshape(b, 4, 3, 4, 4), torch.sin(g.reshape(b, -1).mean(1, keepdim=True) * 11.0 + self.w)

    def action_stage2(self, focuser_image, fm, glog, step, args, prev_local_patch=None, training=True, with_baseline=True):
        assert not training and focuser_image.shape[1:] == (4, 3, 4, 4)
        assert (prev_local_patch is None) == (step == 0)
        b = focuser_image.shape[0]
        nff = args.num_segments_focuser // args.video_div
        seen = focuser_image[:, :(step + 1) * nff].reshape(b, -1).mean(1, keepdim=True)
        pred = torch.sin(seen * 23.0 + self.w * 3.0) + glog
        base = torch.cos(seen * 5.0 + self.w) if with_baseline else None
        patch = focuser_image[:, :(step + 1) * nff, :, :2, :2]
        return pred, base, patch


class _SthData:
    def __init__(self, n):
        g = torch.Generator().manual_seed(7)
        self.g = torch.randn(n, 12, 4, 4, generator=g)
        self.f = torch.randn(n, 12, 4, 4, generator=g)
        self.y = torch.randint(0, 10, (n,), generator=g)

    def __len__(self):
        return self.g.shape[0]

    def __getitem__(self, i):
        return self.g[i], self.f[i], self.y[i]


def _sth_reference(ds, model, a):
    """The reference's loop (STH/evaluate.py:180-213), one batch = the whole set."""
    g, f, y = ds.g, ds.f.view(-1, 4, 3, 4, 4), ds.y
    fm, glog = model.glance(g)
    patch, rew = None, []
    for step in range(a.video_div):
        pred, base, patch = model.action_stage2(f, fm, glog, step, a, prev_local_patch=patch, training=False)
        conf = torch.gather(torch.softmax(pred, 1), 1, y.view(-1, 1)).view(-1)
        bsl = torch.gather(torch.softmax(base, 1), 1, y.view(-1, 1)).view(-1)
        rew.append(float((conf - bsl).mean()))
    a1, a5 = E.accuracy(pred, y, topk=(1, 5))
    return float(a1), float(a5), rew, pred


def _sth_worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = E.validate_sth(_SthData(37), _FakeSth(), torch.nn.CrossEntropyLoss(), _SthArgs(), rank=rank, world=world, quiet=True)
    ret[rank] = (out[0], out[1], out[2])
    dist.barrier()
    dist.destroy_process_group()


def test_validate_sth_matches_the_reference_loop_and_shards():
    a1, a5, rew, pred = _sth_reference(_SthData(37), _FakeSth(), _SthArgs())
    t1, t5, r, logs, lg, tg = E.validate_sth(_SthData(37), _FakeSth(), torch.nn.CrossEntropyLoss(), _SthArgs(), quiet=True,
                                             return_logits=True)
    assert abs(t1 - a1) < 1e-4 and abs(t5 - a5) < 1e-4 and np.allclose(r, rew, atol=1e-6)
    assert torch.allclose(lg, pred) and torch.equal(tg, _SthData(37).y)
    assert logs[-1].startswith(" * Acc@1") and len(logs) == 2 * 5 + 1          # 5 batches of 8: progress + reward lines
    nb = E.validate_sth(_SthData(37), _FakeSth(), torch.nn.CrossEntropyLoss(), _SthArgs(), quiet=True, with_baseline=False)
    assert abs(nb[0] - a1) < 1e-4 and nb[2] == [None, None]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ret = mp.Manager().dict()
    mp.spawn(_sth_worker, args=(2, port, ret), nprocs=2, join=True)
    for rk in range(2):
        assert abs(ret[rk][0] - a1) < 1e-4 and abs(ret[rk][1] - a5) < 1e-4 and np.allclose(ret[rk][2], rew, atol=1e-6)


def test_validate_eight_ranks_ragged_shards():
    """37 clips over 8 ranks (shards of 5,5,5,5,5,4,4,4; batches of 8 -> every shard is one ragged batch): every rank reports
    the metrics of the whole set."""
    single = E.validate(_Data(37), _FakeModel(), torch.nn.CrossEntropyLoss(), _Args(), quiet=True)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(8, port, ret), nprocs=8, join=True)
    assert len(ret) == 8
    for r in range(8):
        assert np.allclose(ret[r], single[:3], atol=1e-4), (r, ret[r], single[:3])
