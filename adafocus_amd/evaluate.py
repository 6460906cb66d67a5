"""Evaluation harness for the offline-inference path (row f3 of SURVEY.md §8f): the metric and
meter helpers of ACT/ops/utils.py and the stage-3 branch of ``validate`` (ACT/main_dist.py:307-422),
with the one thing the reference lacks -- the validation set is SHARDED over ranks and the logits are
all-gathered (the reference evaluates the whole set on every rank, main_dist.py:239).
``validate_sth`` is the Something-Something loop (STH/evaluate.py:165-226): two frame streams, the
``video_div`` focusing steps, the reward bookkeeping of the (optional) baseline branch.

Metrics run on the host over the gathered logits exactly as in the reference (they are O(N*C) and not
on the hot path).  ``cal_map`` keeps the reference's label handling, including its re-ranking of the
label values that occur in the evaluated set (utils.py:56-60, called with assumes_starts_zero=False).
"""
import os
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import torch
import torch.nn.functional as F

from .parallel import gather_variable, shard_range

__all__ = ["AverageMeter", "ProgressMeter", "accuracy", "get_multi_hot", "cal_map", "validate", "validate_sth"]


class AverageMeter(object):
    """utils.py:11-33."""

    def __init__(self, name, fmt=":f"):
        self.name, self.fmt = name, fmt
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count

    def __str__(self):
        return ("{name} {val" + self.fmt + "} ({avg" + self.fmt + "})").format(**self.__dict__)


class ProgressMeter(object):
    """utils.py:95-111."""

    def __init__(self, num_batches, *meters, prefix=""):
        digits = len(str(num_batches // 1))
        self.batch_fmtstr = "[{:" + str(digits) + "d}/" + ("{:" + str(digits) + "d}").format(num_batches) + "]"
        self.meters, self.prefix = meters, prefix

    def print(self, batch, quiet=False):
        out = "\t".join([self.prefix + self.batch_fmtstr.format(batch)] + [str(m) for m in self.meters])
        if not quiet:
            print(out)
        return out + "\n"


def accuracy(output, target, topk=(1,)):
    """Top-k accuracy in percent (utils.py:35-49)."""
    with torch.no_grad():
        maxk = max(topk)
        _, pred = output.topk(maxk, 1, True, True)
        correct = pred.t().eq(target.reshape(1, -1).expand(maxk, -1))
        return [correct[:k].reshape(-1).float().sum(0, keepdim=True).mul_(100.0 / target.size(0)) for k in topk]


def get_multi_hot(test_y, classes, assumes_starts_zero=True):
    """utils.py:51-66: (N, L) integer labels (-1 = absent) -> (N, classes) multi-hot; with
    assumes_starts_zero=False the label VALUES present are first re-ranked to 0..K-1 (in place)."""
    bs = test_y.shape[0]
    if not assumes_starts_zero:
        nxt = 0
        for val in torch.unique(test_y):
            if val >= 0:
                test_y[test_y == val] = nxt
                nxt += 1
    gt = torch.zeros(bs, classes + 1)        # the extra column absorbs the -1 labels
    rows = torch.arange(bs)
    for i in range(test_y.shape[1]):
        gt[rows, test_y[:, i]] = 1
    return gt[:, :classes]


def cal_map(output, old_test_y):
    """Mean average precision in percent over softmax scores (utils.py:68-88).  Returns (mAP, per-class AP)."""
    n, ncls = output.size(0), output.size(1)
    gt = get_multi_hot(old_test_y.clone(), ncls, False)
    probs = F.softmax(output, dim=1)
    rank = torch.arange(1, n + 1).float()
    ap = torch.zeros(ncls)
    for k in range(ncls):
        _, order = torch.sort(probs[:, k], 0, True)
        truth = gt[:, k][order]
        precision = truth.float().cumsum(0).div(rank)
        ap[k] = precision[truth.bool()].sum() / max(float(truth.sum()), 1)
    return ap.mean() * 100, ap * 100


STAGE_THREADS = 4    # worker threads that copy a batch's clips into pinned memory
HOST_THREADS = int(os.environ.get("ADAF_EVAL_HOST_THREADS", "4"))     # torch intra-op threads while a loop runs (restored afterwards); 0 = leave the team alone


class _HostThreads:
    """The evaluation loops keep the host busy with many small jobs (launches, per-clip copies into pinned memory, three scalars per
    batch); torch's default intra-op team is one thread per core (128 on the GPU boxes' hosts) and every Tensor.copy_ / small CPU op
    wakes all of it.  Measured on validate_sth from uint8 clips (64-clip batches, 28.6 ms of GPU work each): 128 threads 34-38 ms per
    batch, 1-4 threads 31 ms; without the baseline branch 24-29 -> 19.9 ms (17.7 ms of GPU work).
    SIDE EFFECT: torch.set_num_threads is process-wide, so any other thread doing CPU torch work during an evaluation call runs on the
    capped team too (dataset transforms in DataLoader WORKER PROCESSES are not affected).  Overlapping calls are reference-counted under
    a lock: the first one in saves the team size, the last one out restores it.  ADAF_EVAL_HOST_THREADS=0 opts out (no cap)."""
    _lock = threading.Lock()
    _depth = 0
    _saved = None

    def __enter__(self):
        cls = _HostThreads
        with cls._lock:
            if cls._depth == 0:
                cls._saved = torch.get_num_threads()
                if HOST_THREADS > 0 and cls._saved > HOST_THREADS:
                    torch.set_num_threads(HOST_THREADS)
            cls._depth += 1
        return self

    def __exit__(self, *exc):
        cls = _HostThreads
        with cls._lock:
            cls._depth -= 1
            if cls._depth == 0 and cls._saved is not None and torch.get_num_threads() != cls._saved:
                torch.set_num_threads(cls._saved)
        return False


_PINNED = {}     # (slot, batch size, item shape, dtype) -> pinned staging buffer, reused across validate() calls


class _Prefetcher:
    """Batches of this rank's shard, one ahead: batch i+1 is stacked into pinned host memory and copied to the GPU on a
    side stream while batch i computes (the reference stacks in DataLoader workers and copies synchronously with
    ``.cuda()``, main_dist.py:329-331).  On a CPU device it degrades to plain stacking."""

    def __init__(self, dataset, start, stop, bs, dev):
        self.ds, self.dev, self.bs = dataset, dev, bs
        self.los = list(range(start, stop, bs))
        self.stop = stop
        self.cuda = dev.type == "cuda"
        self.stream = torch.cuda.Stream(device=dev) if self.cuda else None
        self.devbuf = [None, None]          # device-side landing buffers, reused (no allocator traffic per batch)
        self.consumed = [None, None]        # event: the forward that read a device slot has been enqueued ... and finishes
        self.copied = [None, None]          # event of the last H2D copy out of each pinned slot
        self.pool = ThreadPoolExecutor(max_workers=STAGE_THREADS) if self.cuda else None
        self.slot = 0
        self.next = None
        self._issue(0)

    def _issue(self, i):
        if i >= len(self.los):
            self.next = None
            return
        lo = self.los[i]
        items = [self.ds[j] for j in range(lo, min(lo + self.bs, self.stop))]
        tgt = torch.stack([it[1] for it in items])
        multi = isinstance(items[0][0], (tuple, list))      # a sample made of several tensors (two frame streams): one buffer each
        if not self.cuda:
            if multi:
                data = tuple(torch.stack([it[0][k] for it in items]).to(self.dev) for k in range(len(items[0][0])))
            else:
                data = torch.stack([it[0] for it in items]).to(self.dev)
            self.next = (data, tgt, None, 0)
            return
        firsts = list(items[0][0]) if multi else [items[0][0]]
        n = len(items)
        if self.copied[self.slot] is not None:
            self.copied[self.slot].synchronize()        # the previous copy out of this slot has left the host buffers
        hosts, devs = [], []
        for k, first in enumerate(firsts):
            key = (self.slot, k, self.bs, tuple(first.shape), first.dtype)
            buf = _PINNED.get(key)
            if buf is None:       # page-locking ~150-600 MB costs tens of milliseconds: do it once per process
                buf = torch.empty((self.bs,) + tuple(first.shape), dtype=first.dtype).pin_memory()
                _PINNED[key] = buf
            hosts.append(buf[:n])
            dbuf = self.devbuf[self.slot].get(k) if self.devbuf[self.slot] else None
            if dbuf is None or dbuf.shape[1:] != first.shape or dbuf.dtype != first.dtype:
                # allocated UNDER THE COPY STREAM: the caching allocator hands a block freed on stream A only to allocations on stream A.
                # Allocated on the compute stream (as this was until round 4), slot 1's landing buffer -- first needed while batch 0's
                # forward is still queued -- could be a block that forward had just freed: the H2D copy then landed in memory that
                # queued kernels of batch 0 were still going to write (seen as wrong logits for batch 1 only, video_div = 2, gone under
                # AMD_SERIALIZE_KERNEL=3)
                with torch.cuda.stream(self.stream):
                    dbuf = torch.empty((self.bs,) + tuple(first.shape), dtype=first.dtype, device=self.dev)
                if self.devbuf[self.slot] is None:
                    self.devbuf[self.slot] = {}
                self.devbuf[self.slot][k] = dbuf
            devs.append(dbuf[:n])
        # the labels ride along: a pageable .to(device) inside the loop would block the host until the stream drains
        tkey = ("tgt", self.slot, self.bs, tuple(tgt.shape[1:]), tgt.dtype)
        tpin = _PINNED.get(tkey)
        if tpin is None:
            tpin = torch.empty((self.bs,) + tuple(tgt.shape[1:]), dtype=tgt.dtype).pin_memory()
            _PINNED[tkey] = tpin
        tpin[:n].copy_(tgt)

        def put(j):       # each part of a sample straight into its row of its pinned batch buffer (large copies release the GIL)
            parts = items[j][0] if multi else (items[j][0],)
            for k, q in enumerate(parts):
                hosts[k][j].copy_(q)
        # (measured on the GPU boxes' 256-thread hosts, 64 clips per batch: a pool of STAGE_THREADS workers beats both one copy_ after
        # the other on this thread -- 2.4x slower for the uint8 loops -- and staging / copying in chunks of 8-16 clips)
        list(self.pool.map(put, range(n)))
        with torch.cuda.stream(self.stream):
            if self.consumed[self.slot] is not None:
                self.stream.wait_event(self.consumed[self.slot])   # the batch that last used this slot has been computed
            for h_, d_ in zip(hosts, devs):
                d_.copy_(h_, non_blocking=True)
            tgt_dev = tpin[:n].to(self.dev, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.copied[self.slot] = ev
        self.next = (tuple(devs) if multi else devs[0], tgt_dev, ev, self.slot)
        self.slot ^= 1

    def __iter__(self):
        for i in range(len(self.los)):
            images, tgt, ev, slot = self.next
            if ev is not None:
                cur = torch.cuda.current_stream(self.dev)
                cur.wait_event(ev)
                tgt.record_stream(cur)
                for d_ in (images if isinstance(images, tuple) else (images,)):
                    d_.record_stream(cur)      # (the landing buffers belong to the copy stream's pool; the forward reads them on this one)

            def stage_next(n=i + 1, slot=slot):
                if self.cuda:      # called right after this batch's forward was enqueued: marks the end of its reads
                    done = torch.cuda.Event()
                    done.record(torch.cuda.current_stream(self.dev))
                    self.consumed[slot] = done
                self._issue(n)
            yield i, images, tgt, stage_next


@torch.no_grad()
def validate(dataset, model, criterion, args, rank=0, world=1, batch_size=None, device=None, quiet=False):
    """Evaluation loop of ACT/main_dist.py:307-422 for `args.train_stage` 3 (default; the documented evaluate command), 2 (the policy's MDP
    step by step), 1 (the stage-1 forward) or 0 (a backbone's own classifier) -- arguments, behaviour and return value: `_validate` below --, run with the host's intra-op thread team capped (`_HostThreads`)."""
    with _HostThreads():
        return _validate(dataset, model, criterion, args, rank, world, batch_size, device, quiet)


def _validate(dataset, model, criterion, args, rank=0, world=1, batch_size=None, device=None, quiet=False):
    """Stage-3 evaluation (main_dist.py:307-422, branch :367-371; `args.train_stage == 2`: branch :343-366 through GFV.one_step_act, fp32 clips
    only, with the per-step mAP lines of :411-418) over this rank's shard of `dataset`
    (indexable -> (images, target (L,) int64) with images either the reference's normalised fp32 ``(T*3,H,W)`` clip or
    the loader's stacked uint8 ``(H,W,T*3)`` clip, which is normalised on the GPU -- row f1).  Every rank returns the
    metrics of the WHOLE set: logits and targets are all-gathered once at the end.  The next batch is staged and copied
    while the current one computes.  Returns (top1, top5, mAP, logs)."""
    bs = batch_size or args.batch_size
    start, stop = shard_range(len(dataset), rank, world)
    nb = (stop - start + bs - 1) // bs
    batch_time, losses = AverageMeter("Time", ":6.3f"), AverageMeter("Loss", ":.4e")
    top1, top5, mean_ap = AverageMeter("Acc@1", ":6.2f"), AverageMeter("Acc@5", ":6.2f"), AverageMeter("mAP", ":6.2f")
    progress = ProgressMeter(nb, batch_time, losses, top1, top5, prefix="Test: ")
    model.eval()
    dev = device or next(model.parameters()).device
    logs, preds, step_logits, targets = [], [], [], []
    end = time.time()
    def finish(item):     # host side of a batch: the three scalars, the meters, the log line
        nonlocal end
        bi_, b_, loss_, acc1_, acc5_ = item
        losses.update(loss_.item(), b_)
        top1.update(acc1_[0].item(), b_)
        top5.update(acc5_[0].item(), b_)
        batch_time.update(time.time() - end)
        end = time.time()
        logs.append(progress.print(bi_, quiet=quiet or rank != 0))

    pending = None        # batch i's scalars are read back after batch i+1 has been enqueued: the GPU never idles on them
    lagging = None        # uint8 path: batch i's loss / accuracy kernels are enqueued after batch i+1's forward

    def score(outputs, pred, target, target_full, b, bi, done=None):
        nonlocal pending
        if done is not None:
            torch.cuda.current_stream(dev).wait_event(done)
        loss = criterion(outputs, target.view(b, -1).expand(b, args.num_segments).reshape(-1))
        acc1, acc5 = accuracy(pred, target, topk=(1, 5))
        preds.append(pred)
        step_logits.append(outputs.reshape(b, args.num_segments, -1))
        targets.append(target_full)
        if pending is not None:
            finish(pending)
        pending = (bi, b, loss, acc1, acc5)

    stage = int(getattr(args, "train_stage", 3))
    if stage not in (0, 1, 2, 3):
        raise NotImplementedError("validate: train_stage %d" % stage)
    for bi, images, target_full, stage_next in _Prefetcher(dataset, start, stop, bs, dev):
        target_full = target_full.to(dev)      # already there (and asynchronous) on the GPU path
        target = target_full[:, 0]
        b = images.shape[0]
        if stage in (0, 1):
            # main_dist.py:334-340 (stage 1: the stage-1 forward in eval mode -- glancer + focuser over all frames at once + classifier) and
            # :372-376 (stage 0: one backbone's own classifier on the glancer-sized frames, averaged over the frames)
            if images.dtype == torch.uint8:
                raise NotImplementedError("validate(train_stage=%d) takes the reference's normalised fp32 (T*3,H,W) clips" % stage)
            g = getattr(args, "glance_size", images.shape[-1])
            scan = images
            if g != images.shape[-1]:
                from . import hip_ops
                scan = hip_ops.resize_nearest(images.reshape(-1, 1, images.shape[2], images.shape[3]), g).view(b, -1, g, g)
            if stage == 0:
                pred = model(input=scan, scan=None, glancer=args.pretrain_glancer, backbone_pred=True, one_step=False).mean(1)
                loss = criterion(pred, target)
            else:
                output, pred = model(input=images, scan=scan, training=False, backbone_pred=False, one_step=False)
                if getattr(args, "consensus", "gru") == "gru":
                    loss = criterion(output, target.view(b, -1).expand(b, args.num_segments).reshape(-1))
                else:
                    loss = criterion(output, target)
            stage_next()
            acc1, acc5 = accuracy(pred, target, topk=(1, 5))
            preds.append(pred)
            targets.append(target_full)
            if pending is not None:
                finish(pending)
            pending = (bi, b, loss, acc1, acc5)
            continue
        if stage == 2:
            # main_dist.py:343-366: the policy's MDP step by step through GFV.one_step_act(training=False); `pred` is the last step's
            # last_out, the loss the last step's, every step's prediction is kept for the per-step mAP lines.  (The reference also evaluates
            # get_reward() per step here and drops the result; nothing observable depends on it.)
            if images.dtype == torch.uint8:
                raise NotImplementedError("validate(train_stage=2) takes the reference's normalised fp32 (T*3,H,W) clips")
            t = args.num_segments
            g = getattr(args, "glance_size", images.shape[-1])
            scan = images
            if g != images.shape[-1]:
                from . import hip_ops
                scan = hip_ops.resize_nearest(images.reshape(-1, 1, images.shape[2], images.shape[3]), g).view(b, -1, g, g)
            fmap, fvec = model.glance(scan)
            frames = images.view(b, t, 3, images.shape[2], images.shape[3])
            local = []
            for s in range(t):
                output, pred, _, _, _ = model.one_step_act(frames[:, s], fmap[:, s], fvec[:, s], restart_batch=(s == 0), training=False)
                local.append(pred)
            stage_next()
            loss = criterion(output, target)
            acc1, acc5 = accuracy(pred, target, topk=(1, 5))
            preds.append(pred)
            step_logits.append(torch.stack(local, 1))          # (B, T, C)
            targets.append(target_full)
            if pending is not None:
                finish(pending)
            pending = (bi, b, loss, acc1, acc5)
            continue
        if images.dtype == torch.uint8:
            # the model's own two-stream pipeline: this batch's ingest + glancer + policy overlap the previous batch's
            # hot path; its scores are enqueued one batch late so the consumer stream never stalls the front half
            outputs, pred, _, done, handoff = model.offline_forward_pipelined(images, args.num_segments)
            torch.cuda.current_stream(dev).wait_event(handoff)     # the clip buffer is free once the front half has read it
            stage_next()
            if lagging is not None:
                score(*lagging)
            lagging = (outputs, pred, target, target_full, b, bi, done)
            continue
        # input_prime = F.interpolate(images, (glance_size, glance_size)), main_dist.py:331-332 (identity at 224)
        g = getattr(args, "glance_size", images.shape[-1])
        scan = images
        if g != images.shape[-1]:
            from . import hip_ops
            scan = hip_ops.resize_nearest(images.reshape(-1, 1, images.shape[2], images.shape[3]), g).view(b, -1, g, g)
        outputs, pred = model(input=images, scan=scan, training=False, backbone_pred=False, one_step=True, gpu=args.gpu)
        stage_next()          # host-side stacking + H2D of the next batch run under this batch's kernels
        score(outputs, pred, target, target_full, b, bi)
    if lagging is not None:
        score(*lagging)
    if pending is not None:
        finish(pending)
    if dev.type == "cuda":
        # a persistent GRU scan whose grid barrier timed out poisons its logits with NaN and returns hipSuccess:
        # make that a loud failure of the evaluation, never an accuracy figure
        from . import hip_ops
        starved = hip_ops.gru_scan_timeouts(dev)
        if starved:
            raise RuntimeError("adafocus_amd.validate: %d GRU scan block(s) timed out at their grid barrier (results are NaN-"
                               "poisoned); rerun with hip_ops.set_gru_persistent(2) (cooperative launch)" % starved)
    ncls = args.num_classes
    empty = torch.zeros((0, ncls), device=dev)
    all_pred = gather_variable(torch.cat(preds) if preds else empty).cpu()
    all_tgt = gather_variable(torch.cat(targets) if targets else torch.zeros((0, 1), dtype=torch.int64, device=dev)).cpu()
    acc1, acc5 = accuracy(all_pred, all_tgt[:, 0], topk=(1, 5))
    if getattr(args, "dataset", "actnet") == "fcvid":
        m_ap, _ = cal_map(all_pred, all_tgt)
    else:
        m_ap, _ = cal_map(all_pred, all_tgt[:, 0:1])
    mean_ap.update(float(m_ap), 1)
    logs.append("mAP: {mAP:.5f}\n".format(mAP=mean_ap.avg))
    summary = " * Acc@1 {:.5f} Acc@5 {:.5f} mAP {:.5f}".format(acc1[0].item(), acc5[0].item(), mean_ap.avg)
    if rank == 0 and not quiet:
        print(summary)
    logs.append(summary + "\n")
    if stage == 2:         # main_dist.py:411-418: the prediction after every step of the MDP
        all_steps = gather_variable(torch.cat(step_logits) if step_logits else torch.zeros((0, args.num_segments, ncls), device=dev)).cpu()
        for i in range(args.num_segments):
            m_i, _ = cal_map(all_steps[:, i, :], all_tgt if getattr(args, "dataset", "actnet") == "fcvid" else all_tgt[:, 0:1])
            line = "mAP @ time step {step}: {mAP:.5f}\n".format(mAP=float(m_i), step=i)
            logs.append(line)
            if rank == 0 and not quiet:
                print(line)
    return acc1[0].item(), acc5[0].item(), mean_ap.avg, logs


class _TwoStream:
    """(glancer clip, focuser clip, target) items as a two-part sample: the prefetcher stages each clip straight into its own pinned
    batch buffer and copies both with the batch (no per-sample torch.cat, no shared shape: the two streams may differ in length)."""

    def __init__(self, ds):
        self.ds = ds

    def __len__(self):
        return len(self.ds)

    def __getitem__(self, i):
        g, f, t = self.ds[i]
        # (a torch.cat here is a 9.6 MB single-threaded copy per sample: it was most of a batch's 123 ms)
        return (g, f), torch.as_tensor(t).reshape(-1)[:1]


@torch.no_grad()
def validate_sth(dataset, model, criterion, args, rank=0, world=1, batch_size=None, device=None, quiet=False, with_baseline=True,
                 return_logits=False):
    """Something-Something evaluation loop of STH/evaluate.py:165-226 (arguments, behaviour and return value: `_validate_sth` below),
    run with the host's intra-op thread team capped (`_HostThreads`)."""
    with _HostThreads():
        return _validate_sth(dataset, model, criterion, args, rank, world, batch_size, device, quiet, with_baseline, return_logits)


def _validate_sth(dataset, model, criterion, args, rank=0, world=1, batch_size=None, device=None, quiet=False, with_baseline=True,
                  return_logits=False):
    """Something-Something evaluation (STH/evaluate.py:165-226) over this rank's shard of `dataset` (indexable ->
    (glancer_images, focuser_images, target): either normalised fp32 (Tg*3,H,W) / (Tf*3,H,W) clips like the reference's loader
    emits, or the loader's stacked uint8 (H,W,Tg*3) / (H,W,Tf*3) clips, normalised on the GPU (row f1: 154 MB instead of 617 MB of
    H2D per 64-clip batch at T = 8 + 8).
    Per batch: nearest resize of the glancer frames to glance_size (:188), `model.glance`, then for each of the
    `args.video_div` focusing steps `model.action_stage2(..., training=False)` with the previous steps' patches carried
    (:198-201), the loss, and -- when `with_baseline` (the reference always computes it; it only feeds the logged reward)
    -- reward = p_target(pred) - p_target(baseline) (:203-210).  Metrics are those of the WHOLE set on every rank: the last
    step's logits, the targets and the per-sample rewards are all-gathered once at the end (the reference evaluates the
    full set on each rank).  Returns (top1, top5, [mean reward per step], logs[, logits, targets])."""
    bs = batch_size or args.batch_size
    start, stop = shard_range(len(dataset), rank, world)
    nb = (stop - start + bs - 1) // bs
    batch_time, losses = AverageMeter("Time", ":6.3f"), AverageMeter("Loss", ":.4e")
    top1, top5 = AverageMeter("Acc@1", ":6.2f"), AverageMeter("Acc@5", ":6.2f")
    reward_list = [AverageMeter("Rew", ":6.5f") for _ in range(args.video_div)]
    progress = ProgressMeter(nb, batch_time, losses, top1, top5, prefix="Test: ")
    model.eval()
    dev = device or next(model.parameters()).device
    tg3 = 3 * args.num_segments_glancer
    logs, preds, targets, rewards = [], [], [], [[] for _ in range(args.video_div)]
    end = time.time()
    pending = None

    def finish(item):       # one batch late: the GPU never idles on the read-back of three scalars
        nonlocal end
        bi_, b_, loss_, acc1_, acc5_, rews_ = item
        losses.update(loss_.item(), b_)
        top1.update(acc1_[0].item(), b_)
        top5.update(acc5_[0].item(), b_)
        for meter, r in zip(reward_list, rews_):
            if r is not None:
                meter.update(r.mean().item(), b_)
        batch_time.update(time.time() - end)
        end = time.time()
        logs.append(progress.print(bi_, quiet=quiet or rank != 0))
        logs.append(" ".join(str(m.avg) for m in reward_list) + "\n")

    for bi, images, target, stage_next in _Prefetcher(_TwoStream(dataset), start, stop, bs, dev):
        target = target.to(dev)[:, 0]
        glancer_images, focuser_flat = images
        b = glancer_images.shape[0]
        g = getattr(args, "glance_size", None)
        local_patch, pred, loss, rews = None, None, None, []
        if glancer_images.dtype == torch.uint8:
            # the loader's stacked uint8 clips (H, W, T*3) -- what Stack() hands ToTorchFormatTensor (STH/ops/transforms.py:303-336):
            # 4x fewer bytes over PCIe, normalised on the GPU in the reference's op order (GroupNormalize, :64-77: bit-exact, G8)
            # straight into the pixel-major frames the glancer's stem and the patch gather read
            from . import hip_ops
            from ._lib import LAYOUT_NHWC4
            from .transforms import ingest_uint8
            hh = glancer_images.shape[1]
            g4 = ingest_uint8(glancer_images, args.num_segments_glancer, model.input_mean, model.input_std)
            f4 = ingest_uint8(focuser_flat, args.num_segments_focuser, model.input_mean, model.input_std)
            if g is not None and g != hh:         # F.interpolate(glancer_images, (glance_size, glance_size)): nearest (evaluate.py:188)
                g4 = hip_ops.resize_nearest(g4, g, LAYOUT_NHWC4)
            fm4, glog = model.glance_nhwc4(g4, b)
            for step in range(args.video_div):
                pred, base, local_patch = model.action_stage2_nhwc4(f4, fm4, glog, step, args, prev_patch4=local_patch,
                                                                    with_baseline=with_baseline)
                loss = criterion(pred, target)
                if with_baseline:
                    conf = torch.gather(F.softmax(pred, 1), 1, target.view(-1, 1)).view(-1)
                    bsl = torch.gather(F.softmax(base, 1), 1, target.view(-1, 1)).view(-1)
                    rews.append(conf - bsl)
                    rewards[step].append(rews[-1])
                else:
                    rews.append(None)
        else:
            hh, ww = glancer_images.shape[2], glancer_images.shape[3]
            if g is not None and g != hh:         # F.interpolate(glancer_images, (glance_size, glance_size)): nearest (evaluate.py:188)
                from . import hip_ops
                glancer_images = hip_ops.resize_nearest(glancer_images.reshape(-1, 1, hh, ww), g).view(b, tg3, g, g)
            focuser_images = focuser_flat.reshape(b, args.num_segments_focuser, 3, hh, ww)
            fm, glog = model.glance(glancer_images)
            for step in range(args.video_div):
                pred, base, local_patch = model.action_stage2(focuser_images, fm, glog, step, args, prev_local_patch=local_patch,
                                                              training=False, with_baseline=with_baseline)
                loss = criterion(pred, target)
                if with_baseline:
                    conf = torch.gather(F.softmax(pred, 1), 1, target.view(-1, 1)).view(-1)
                    bsl = torch.gather(F.softmax(base, 1), 1, target.view(-1, 1)).view(-1)
                    rews.append(conf - bsl)
                    rewards[step].append(rews[-1])
                else:
                    rews.append(None)
        stage_next()
        acc1, acc5 = accuracy(pred, target, topk=(1, 5))
        preds.append(pred)
        targets.append(target)
        if pending is not None:
            finish(pending)
        pending = (bi, b, loss, acc1, acc5, rews)
    if pending is not None:
        finish(pending)
    if dev.type == "cuda":
        from . import hip_ops
        starved = hip_ops.gru_scan_timeouts(dev)
        if starved:
            raise RuntimeError("adafocus_amd.validate_sth: %d GRU scan block(s) timed out at their grid barrier" % starved)
    ncls = args.num_classes
    all_pred = gather_variable(torch.cat(preds) if preds else torch.zeros((0, ncls), device=dev)).cpu()
    all_tgt = gather_variable(torch.cat(targets) if targets else torch.zeros((0,), dtype=torch.int64, device=dev)).cpu()
    acc1, acc5 = accuracy(all_pred, all_tgt, topk=(1, 5))
    mean_rewards = []
    for step in range(args.video_div):
        if with_baseline:
            r = gather_variable(torch.cat(rewards[step]) if rewards[step] else torch.zeros((0,), device=dev)).cpu()
            mean_rewards.append(float(r.mean()) if r.numel() else 0.0)
        else:
            mean_rewards.append(None)
    summary = " * Acc@1 {:.5f} Acc@5 {:.5f} reward of each step: {}".format(acc1[0].item(), acc5[0].item(), mean_rewards)
    if rank == 0 and not quiet:
        print(summary)
    logs.append(summary + "\n")
    out = (acc1[0].item(), acc5[0].item(), mean_rewards, logs)
    return out + (all_pred, all_tgt) if return_logits else out
