"""Tensor-level wrappers over the C ABI (include/adafocus.h).  Everything here enqueues
hand-written gfx950 kernels on the current HIP stream; nothing falls back to ATen.

Activation tensors are fp32 NHWC (``(N, H, W, C)`` contiguous) unless a function says otherwise.
"""
import ctypes as C

import torch

from . import _lib as L
from ._lib import ACT_NONE, ACT_RELU, ACT_RELU6, ACT_SIGMOID, ACT_SWISH, LAYOUT_NCHW, LAYOUT_NHWC, LAYOUT_NHWC4  # noqa: F401
from ._lib import MATH_F32, MATH_F32_SPLIT_BF16  # noqa: F401
from ._lib import DTYPE_F32, DTYPE_F16  # noqa: F401


def _h(t):
    return L.handle(t.device)


def crop_gather(frames, actions, patch, frames_per_action=1, layout=LAYOUT_NCHW, return_coords=False):
    """Batched get_patch (ACT/models/utils.py:37-51).  frames (N,C,H,W) fp32, actions (M,2) fp32 in
    [0,1] with M*frames_per_action == N.  Returns (N,C,P,P) [NCHW], (N,P,P,C) [NHWC] or
    (N,P,P,4) [NHWC4]; optionally also the int32 (M,2) window origins."""
    L.need_gpu_f32(frames, actions)
    if frames.dim() != 4 or actions.dim() != 2 or actions.shape[1] != 2:
        raise ValueError("crop_gather: frames (N,C,H,W) and actions (M,2) expected")
    frames = frames.contiguous()
    actions = actions.contiguous()
    n, c, hh, ww = frames.shape
    p = int(patch)
    if layout == LAYOUT_NCHW:
        out = torch.empty((n, c, p, p), device=frames.device, dtype=torch.float32)
    elif layout == LAYOUT_NHWC:
        out = torch.empty((n, p, p, c), device=frames.device, dtype=torch.float32)
    else:
        out = torch.empty((n, p, p, 4), device=frames.device, dtype=torch.float32)
    coords = torch.empty((actions.shape[0], 2), device=frames.device, dtype=torch.int32) if return_coords else None
    h = _h(frames)
    L.check(L.load_library().adaf_crop_gather_f32(h, L.ptr(frames), n, c, hh, ww, L.ptr(actions), actions.shape[0],
                                                  int(frames_per_action), p, L.ptr(out), layout, L.ptr(coords),
                                                  L.stream_ptr()), h)
    return (out, coords) if return_coords else out


def pack_conv_weight(w_oihw, cin_pad=None):
    """OIHW -> OHWI (input channels zero-padded to a multiple of 4)."""
    L.need_gpu_f32(w_oihw)
    w = w_oihw.contiguous()
    co, ci, kh, kw = w.shape
    cp = cin_pad or ((ci + 3) // 4 * 4)
    out = torch.empty((co, kh, kw, cp), device=w.device, dtype=torch.float32)
    h = _h(w)
    L.check(L.load_library().adaf_pack_conv_weight_f32(h, L.ptr(w), co, ci, kh, kw, cp, L.ptr(out), L.stream_ptr()), h)
    return out


def fold_bn(gamma, beta, mean, var, eps=1e-5):
    L.need_gpu_f32(gamma, beta, mean, var)
    c = gamma.numel()
    scale = torch.empty(c, device=gamma.device, dtype=torch.float32)
    bias = torch.empty(c, device=gamma.device, dtype=torch.float32)
    h = _h(gamma)
    L.check(L.load_library().adaf_fold_bn_f32(h, L.ptr(gamma.contiguous()), L.ptr(beta.contiguous()),
                                              L.ptr(mean.contiguous()), L.ptr(var.contiguous()), C.c_float(eps), c,
                                              L.ptr(scale), L.ptr(bias), L.stream_ptr()), h)
    return scale, bias


def conv2d_bn_act(x, w_ohwi, scale=None, bias=None, residual=None, stride=1, pad=0, act=ACT_NONE, tsm_segments=0,
                  tsm_div=8, tile=0, naive=False, out=None):
    """x (N,H,W,Cin) NHWC, w (Cout,KH,KW,Cin) -> (N,OH,OW,Cout).  `naive=True` runs the
    one-thread-per-output cross-check kernel instead of the MFMA engine (tests only)."""
    L.need_gpu_f32(x, w_ohwi, scale, bias, residual)
    x = x.contiguous()
    w_ohwi = w_ohwi.contiguous()
    n, hh, ww, cin = x.shape
    cout, kh, kw, cin_w = w_ohwi.shape
    if cin_w != cin:
        raise ValueError("conv2d_bn_act: x has %d channels, weight expects %d" % (cin, cin_w))
    oh = (hh + 2 * pad - kh) // stride + 1
    ow = (ww + 2 * pad - kw) // stride + 1
    if out is None:
        out = torch.empty((n, oh, ow, cout), device=x.device, dtype=torch.float32)
    if residual is not None:
        residual = residual.contiguous()
    p = L.ConvParams(n=n, h=hh, w=ww, cin=cin, cout=cout, kh=kh, kw=kw, stride=stride, pad=pad, act=act,
                     tsm_segments=tsm_segments, tsm_div=tsm_div, ldx=0, ldo=0, ldr=0, tile=tile)
    h = _h(x)
    lib = L.load_library()
    fn = lib.adaf_conv2d_naive_f32 if naive else lib.adaf_conv2d_bn_act_f32
    L.check(fn(h, C.byref(p), L.ptr(x), L.ptr(w_ohwi), L.ptr(scale), L.ptr(bias), L.ptr(residual), L.ptr(out),
               L.stream_ptr()), h)
    return out


def linear(x, weight, bias=None, act=ACT_NONE, tile=0):
    """nn.Linear on the MFMA engine: x (rows, in) row-major, weight (out, in)."""
    rows, fin = x.shape
    y = conv2d_bn_act(x.reshape(rows, 1, 1, fin), weight.reshape(weight.shape[0], 1, 1, fin), None, bias, act=act,
                      tile=tile)
    return y.reshape(rows, weight.shape[0])


def maxpool3x3s2(x):
    L.need_gpu_f32(x)
    x = x.contiguous()
    n, hh, ww, c = x.shape
    out = torch.empty((n, (hh - 1) // 2 + 1, (ww - 1) // 2 + 1, c), device=x.device, dtype=torch.float32)
    h = _h(x)
    L.check(L.load_library().adaf_maxpool3x3s2_f32(h, L.ptr(x), n, hh, ww, c, L.ptr(out), L.stream_ptr()), h)
    return out


def global_avgpool(x):
    L.need_gpu_f32(x)
    x = x.contiguous()
    n, hh, ww, c = x.shape
    out = torch.empty((n, c), device=x.device, dtype=torch.float32)
    h = _h(x)
    L.check(L.load_library().adaf_global_avgpool_f32(h, L.ptr(x), n, hh * ww, c, L.ptr(out), c, L.stream_ptr()), h)
    return out


def temporal_shift(x, n_segment, fold_div, layout=LAYOUT_NCHW):
    """TemporalShift.shift (STH/ops/temporal_shift.py:28-46).  x (NT,C,H,W) [NCHW] or (NT,H,W,C)."""
    L.need_gpu_f32(x)
    x = x.contiguous()
    if layout == LAYOUT_NCHW:
        nt, c, hh, ww = x.shape
    else:
        nt, hh, ww, c = x.shape
    out = torch.empty_like(x)
    h = _h(x)
    L.check(L.load_library().adaf_temporal_shift_f32(h, L.ptr(x), nt, c, hh * ww, int(n_segment), int(fold_div), layout,
                                                     L.ptr(out), L.stream_ptr()), h)
    return out


def gru_cls_forward(x, w_ih, w_hh, b_ih, b_hh, fc_w, fc_b):
    """RecurrentClassifier.forward (ACT/models/gfv_net.py:427-435).  x (B,T,F) (last dim may be a
    strided view with stride 1) -> (logits (B*T,C), last (B,C))."""
    L.need_gpu_f32(x, w_ih, w_hh, b_ih, b_hh, fc_w, fc_b)
    b, t, f = x.shape
    if x.stride(2) != 1 or x.stride(0) != t * x.stride(1):
        x = x.contiguous()
    hid, ncls = w_hh.shape[1], fc_w.shape[0]
    lib = L.load_library()
    ws_bytes = lib.adaf_gru_cls_workspace_bytes(b, t, hid)
    ws = torch.empty(max(ws_bytes // 4, 1), device=x.device, dtype=torch.float32)
    logits = torch.empty((b * t, ncls), device=x.device, dtype=torch.float32)
    last = torch.empty((b, ncls), device=x.device, dtype=torch.float32)
    h = _h(x)
    L.check(lib.adaf_gru_cls_forward_f32(h, L.ptr(x), x.stride(1), b, t, f, hid, ncls, L.ptr(w_ih.contiguous()),
                                         L.ptr(w_hh.contiguous()), L.ptr(b_ih.contiguous()), L.ptr(b_hh.contiguous()),
                                         L.ptr(fc_w.contiguous()), L.ptr(fc_b.contiguous()), L.ptr(logits), L.ptr(last),
                                         L.ptr(ws), ws_bytes, L.stream_ptr()), h)
    return logits, last


def fc_meanpool_forward(feat, batch, fc_w, fc_b, global_logit=None):
    """mean_t FC(f_t) (+ mean_t glancer logits) -- STH/models/gfv_net.py:164-174.
    feat (B*T,F); global_logit (B,Tg,C) or None -> (B,C)."""
    L.need_gpu_f32(feat, fc_w, fc_b, global_logit)
    feat = feat.contiguous()
    rows, f = feat.shape
    t = rows // batch
    ncls = fc_w.shape[0]
    ws = torch.empty(max(rows * ncls, 1), device=feat.device, dtype=torch.float32)
    out = torch.empty((batch, ncls), device=feat.device, dtype=torch.float32)
    tg = 0
    if global_logit is not None:
        global_logit = global_logit.contiguous()
        tg = global_logit.shape[1]
    h = _h(feat)
    L.check(L.load_library().adaf_fc_meanpool_forward_f32(h, L.ptr(feat), batch, t, f, ncls, L.ptr(fc_w.contiguous()),
                                                          L.ptr(fc_b.contiguous()), L.ptr(global_logit), tg, L.ptr(out),
                                                          L.ptr(ws), ws.numel() * 4, L.stream_ptr()), h)
    return out


def copy2d(src, dst_view):
    """dst_view[:, :] = src for a row-strided destination (torch.cat on dim=1 without a new tensor)."""
    L.need_gpu_f32(src, dst_view)
    rows, cols = src.shape
    if src.stride(1) != 1 or dst_view.stride(1) != 1:
        raise ValueError("copy2d: unit inner stride required")
    h = _h(src)
    L.check(L.load_library().adaf_copy2d_f32(h, L.ptr(src), src.stride(0), L.ptr(dst_view), dst_view.stride(0), rows, cols,
                                             L.stream_ptr()), h)
    return dst_view


class _StreamScratch:
    """One scratch tensor per HIP stream: passes enqueued on different streams (pipelined batches) must not share
    intermediates.  Each tensor is allocated while its stream is current, so the caching allocator only ever reuses
    its memory in that stream's order; the least recently used entries are dropped beyond `cap` streams (short-lived
    streams would otherwise pin gigabytes each).  `cap` defaults to 6 (bench.py uses 3 streams + the model's own 2) and
    can be raised with ADAF_SCRATCH_STREAMS; an eviction is reported once per object on stderr, because rotating over
    more streams than `cap` means a multi-GB reallocation per call."""

    def __init__(self, device, cap=None):
        import os
        self.device = device
        self.cap = int(cap or os.environ.get("ADAF_SCRATCH_STREAMS", 6))
        self._ws, self._warned = {}, False

    def get(self, need_bytes):
        key = torch.cuda.current_stream(self.device).cuda_stream
        ws = self._ws.pop(key, None)
        if ws is None or ws.numel() * 4 < need_bytes:
            ws = torch.empty(max((need_bytes + 3) // 4, 1), device=self.device, dtype=torch.float32)
        self._ws[key] = ws                       # (re)insert as most recently used
        while len(self._ws) > self.cap:
            self._ws.pop(next(iter(self._ws)))
            if not self._warned:
                import sys
                print("adafocus_amd: more than %d streams rotate over one network's scratch space; the least recently used "
                      "buffer was dropped (set ADAF_SCRATCH_STREAMS to keep more)" % self.cap, file=sys.stderr)
                self._warned = True
        return ws

    def __len__(self):
        return len(self._ws)


class ResNet50Trunk:
    """adaf_resnet50: the whole local CNN (stem ... layer4, avgpool) as ~55 back-to-back launches on
    one stream, weights packed / BN folded once.  Replaces ResNet.get_featmap(x, pooled=True)
    (ACT/models/resnet.py:211-225) and TSN.forward(no_reshape=True) (STH/models/tsn.py:215-241)."""

    def __init__(self, device):
        self.device = torch.device(device)
        self._h = L.handle(self.device)
        self._lib = L.load_library()
        net = C.c_void_p()
        L.check(self._lib.adaf_resnet50_create(self._h, C.byref(net)), self._h)
        self._net = net
        self._scratch = _StreamScratch(self.device)
        self.n_launches = self._lib.adaf_resnet50_launch_count(net)

    def __del__(self):
        try:
            if getattr(self, "_net", None):
                self._lib.adaf_resnet50_destroy(self._net)
                self._net = None
        except Exception:
            pass

    def load(self, params):
        """params: mapping torchvision-style name -> tensor (on this device).  Non-trunk entries
        (fc.*, num_batches_tracked) are ignored."""
        keep = []
        for name, t in params.items():
            if name.startswith("fc.") or name.endswith("num_batches_tracked"):
                continue
            L.need_gpu_f32(t)
            t = t.detach().contiguous()
            keep.append(t)
            L.check(self._lib.adaf_resnet50_set_param(self._net, name.encode(), L.ptr(t), t.numel()), self._h)
        L.check(self._lib.adaf_resnet50_finalize(self._net, L.stream_ptr()), self._h)
        del keep

    def _workspace(self, n, patch):
        """One scratch buffer per HIP stream: passes enqueued on different streams (pipelined batches)
        must not share intermediates."""
        need = self._lib.adaf_resnet50_workspace_bytes(self._net, n, patch)
        return self._scratch.get(need), need

    def forward(self, patches_nhwc4, tsm_segments=0, tsm_div=8, out=None):
        """patches (N,P,P,4) -> (N,2048); `out` may be a (N,2048) row-strided view (e.g. the tail of
        the GRU input matrix)."""
        L.need_gpu_f32(patches_nhwc4, out)
        x = patches_nhwc4.contiguous()
        n, p = x.shape[0], x.shape[1]
        if out is None:
            out = torch.empty((n, 2048), device=x.device, dtype=torch.float32)
        ws, need = self._workspace(n, p)
        L.check(self._lib.adaf_resnet50_forward(self._net, L.ptr(x), n, p, int(tsm_segments), int(tsm_div), L.ptr(out),
                                                out.stride(0), L.ptr(ws), need, L.stream_ptr()), self._h)
        return out

    def forward_frames(self, frames, actions, patch, frames_per_action=1, tsm_segments=0, tsm_div=8, out=None):
        """get_patch + trunk in one call (adaf_resnet50_forward_frames): frames (N,3,H,W) planar or (N,H,W,4) pixel-major, actions
        (k * N / frames_per_action, 2) fp32 (y, x) -> (k * N, 2048); the stem gathers its own windows where its strip kernel applies."""
        L.need_gpu_f32(frames, actions, out)
        L.on_current_device(frames, actions, out)
        x = frames.contiguous()
        pixel_major = x.shape[-1] == 4 and x.shape[1] != 3
        nf = x.shape[0]
        hh, ww = (x.shape[1], x.shape[2]) if pixel_major else (x.shape[2], x.shape[3])
        act = actions.contiguous()
        n = act.shape[0] // (nf // frames_per_action) * nf
        if out is None:
            out = torch.empty((n, 2048), device=x.device, dtype=torch.float32)
        ws, need = self._workspace(n, patch)
        L.check(self._lib.adaf_resnet50_forward_frames(self._net, L.ptr(x), LAYOUT_NHWC4 if pixel_major else LAYOUT_NCHW, nf, hh, ww, L.ptr(act),
                                                       act.shape[0], int(frames_per_action), int(patch), int(tsm_segments), int(tsm_div),
                                                       L.ptr(out), out.stride(0), L.ptr(ws), need, L.stream_ptr()), self._h)
        return out

    def forward_map(self, patches_nhwc4, tsm_segments=0, tsm_div=8):
        """patches (N,P,P,4) -> (featmap (N,s,s,2048) NHWC, pooled feature (N,2048)): ResNet.get_featmap(x, pooled=False)."""
        L.need_gpu_f32(patches_nhwc4)
        x = patches_nhwc4.contiguous()
        n, p = x.shape[0], x.shape[1]
        s = int(self._lib.adaf_resnet50_map_size(p))
        fmap = torch.empty((n, s, s, 2048), device=x.device, dtype=torch.float32)
        feat = torch.empty((n, 2048), device=x.device, dtype=torch.float32)
        ws, need = self._workspace(n, p)
        L.check(self._lib.adaf_resnet50_forward_map(self._net, L.ptr(x), n, p, int(tsm_segments), int(tsm_div), L.ptr(fmap), L.ptr(feat), 2048,
                                                    L.ptr(ws), need, L.stream_ptr()), self._h)
        return fmap, feat

    def profile(self, patches_nhwc4, tsm_segments=0, tsm_div=8):
        """One forward bracketed by HIP events per launch.  Returns a list of dicts
        {ms, flops, bytes, tile} (flops = 0 for the pooling launches)."""
        x = patches_nhwc4.contiguous()
        n, p = x.shape[0], x.shape[1]
        out = torch.empty((n, 2048), device=x.device, dtype=torch.float32)
        ws, need = self._workspace(n, p)
        k = self.n_launches = self._lib.adaf_resnet50_launch_count(self._net)
        ms = (C.c_float * k)()
        fl = (C.c_double * k)()
        by = (C.c_double * k)()
        tl = (C.c_int * k)(*([-1] * k))
        L.check(self._lib.adaf_resnet50_forward_profiled(self._net, L.ptr(x), n, p, int(tsm_segments), int(tsm_div),
                                                         L.ptr(out), 2048, L.ptr(ws), need, L.stream_ptr(), ms, fl, by,
                                                         tl), self._h)
        return [dict(ms=ms[i], flops=fl[i], bytes=by[i], tile=tl[i]) for i in range(k) if tl[i] != -1]

    def set_fusion(self, on):
        """Stage-1 conv2 -> conv3 (-> next conv1) and stem + max-pool as single launches (default on; bit-identical)."""
        L.check(self._lib.adaf_resnet50_set_fusion(self._net, int(on)), self._h)   # 2 = also the fused stem where it does not pay (tests)

    def set_shift_place(self, place):
        """'blockres' (default: the shift inside every Bottleneck conv1's operand load) or 'block' (the shift in front of the whole
        Bottleneck: conv1, downsample and identity read the shifted map) -- STH/ops/temporal_shift.py:99-142."""
        code = {"blockres": 0, "block": 1}[place]
        L.check(self._lib.adaf_resnet50_set_shift_place(self._net, code), self._h)

    def set_latency_rows(self, rows):
        """Convs whose GEMM has at most `rows` rows take the small-batch form (conv_lat.hip; bit-identical); 0 = never, < 0 = default."""
        L.check(self._lib.adaf_resnet50_set_latency_rows(self._net, int(rows)), self._h)

    def set_tiles(self, tiles):
        arr = (C.c_int * len(tiles))(*tiles)
        L.check(self._lib.adaf_resnet50_set_tiles(self._net, arr, len(tiles)), self._h)

    def set_math(self, mode):
        """"f32" (default: fp32 MFMA, exact FMA chain) or "split_bf16" (opt-in: fp32 operands decomposed into three
        bf16 parts, six bf16 MFMA products per element pair, fp32 accumulate -- include/adafocus.h ADAF_MATH_*)."""
        code = {"f32": MATH_F32, "split_bf16": MATH_F32_SPLIT_BF16}.get(mode, mode)
        L.check(self._lib.adaf_resnet50_set_math(self._net, int(code)), self._h)


def set_gru_persistent(mode, device=None):
    """GRU scans: 0/False = two launches per step, 1/True = one persistent kernel (default), 2 = persistent kernel via
    hipLaunchCooperativeKernel (include/adafocus.h)."""
    dev = torch.device(device if device is not None else ("cuda", torch.cuda.current_device()))
    h = L.handle(dev)
    L.check(L.load_library().adaf_set_gru_persistent(h, int(mode)), h)


def gru_scan_timeouts(device=None):
    """Blocks of persistent GRU scans whose grid barrier timed out since the handle was created (they NaN-poison their
    outputs).  Synchronises the device; 0 = every scan completed normally."""
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    h = L.handle(dev)
    n = C.c_uint(0)
    L.check(L.load_library().adaf_gru_scan_timeouts(h, C.byref(n)), h)
    return int(n.value)


def set_conv_pos_major(on, device=None):
    """k x k convs: position-major tiles with padding-tap skipping on / off (bit-identical; include/adafocus.h)."""
    dev = torch.device(device if device is not None else ("cuda", torch.cuda.current_device()))
    h = L.handle(dev)
    L.check(L.load_library().adaf_set_conv_pos_major(h, int(on)), h)


def pack_dw_weight(w_c133):
    """PyTorch depthwise weight [C,1,3,3] -> [3,3,C]."""
    L.need_gpu_f32(w_c133)
    w = w_c133.contiguous()
    c = w.shape[0]
    out = torch.empty((3, 3, c), device=w.device, dtype=torch.float32)
    h = _h(w)
    L.check(L.load_library().adaf_pack_dw_weight_f32(h, L.ptr(w), c, L.ptr(out), L.stream_ptr()), h)
    return out


def dwconv3x3_bn_act(x, w_33c, scale, bias, stride=1, act=ACT_RELU6):
    """Depthwise 3x3 (pad 1) + BN affine + activation; x (N,H,W,C) NHWC."""
    L.need_gpu_f32(x, w_33c, scale, bias)
    x = x.contiguous()
    n, hh, ww, c = x.shape
    out = torch.empty((n, (hh - 1) // stride + 1, (ww - 1) // stride + 1, c), device=x.device, dtype=torch.float32)
    h = _h(x)
    L.check(L.load_library().adaf_dwconv3x3_bn_act_f32(h, L.ptr(x), n, hh, ww, c, stride, L.ptr(w_33c.contiguous()),
                                                       L.ptr(scale.contiguous()), L.ptr(bias.contiguous()), act,
                                                       L.ptr(out), L.stream_ptr()), h)
    return out


def grid_actions(logits, table):
    """(rows, A) logits + (A,2) table -> (idx int64 (rows,), actions fp32 (rows,2)); first-max argmax."""
    L.need_gpu_f32(logits, table)
    logits = logits.contiguous()
    rows, a = logits.shape
    idx = torch.empty((rows,), device=logits.device, dtype=torch.int64)
    act = torch.empty((rows, 2), device=logits.device, dtype=torch.float32)
    h = _h(logits)
    L.check(L.load_library().adaf_grid_actions_f32(h, L.ptr(logits), rows, a, L.ptr(table.contiguous()), L.ptr(idx), L.ptr(act),
                                                   L.stream_ptr()), h)
    return idx, act


def argmax_rows(logits):
    """First-maximum arg-max of every row (ActorCritic.act's `.max(1)[1]`, ACT/models/ppo.py:94) -> int64 (rows,)."""
    L.need_gpu_f32(logits)
    logits = logits.contiguous()
    rows, a = logits.shape
    idx = torch.empty((rows,), device=logits.device, dtype=torch.int64)
    h = _h(logits)
    L.check(L.load_library().adaf_grid_actions_f32(h, L.ptr(logits), rows, a, None, L.ptr(idx), None, L.stream_ptr()), h)
    return idx


def gru_seq_forward(x, w_ih, w_hh, b_ih, b_hh, h0=None):
    """nn.GRU(batch_first=True): x (B,T,F), h0 (B,H) or None (= zeros) -> hidden states (B,T,H)."""
    L.need_gpu_f32(x, w_ih, w_hh, b_ih, b_hh, h0)
    if h0 is not None:
        h0 = h0.contiguous()
    b, t, f = x.shape
    if x.stride(2) != 1 or x.stride(0) != t * x.stride(1):
        x = x.contiguous()
    hid = w_hh.shape[1]
    lib = L.load_library()
    ws_bytes = lib.adaf_gru_cls_workspace_bytes(b, t, hid)
    ws = torch.empty(max(ws_bytes // 4, 1), device=x.device, dtype=torch.float32)
    hs = torch.empty((b, t, hid), device=x.device, dtype=torch.float32)
    h = _h(x)
    L.check(lib.adaf_gru_seq_forward_f32(h, L.ptr(x), x.stride(1), b, t, f, hid, L.ptr(w_ih.contiguous()),
                                         L.ptr(w_hh.contiguous()), L.ptr(b_ih.contiguous()), L.ptr(b_hh.contiguous()),
                                         L.ptr(h0), L.ptr(hs), L.ptr(ws), ws_bytes, L.stream_ptr()), h)
    return hs


class MobileNetV2Net:
    """adaf_mobilenetv2: the glancer's feature extractor on the conv engine + depthwise kernel."""

    def __init__(self, device):
        self.device = torch.device(device)
        self._h = L.handle(self.device)
        self._lib = L.load_library()
        net = C.c_void_p()
        L.check(self._lib.adaf_mobilenetv2_create(self._h, C.byref(net)), self._h)
        self._net = net
        self._scratch = _StreamScratch(self.device)

    def __del__(self):
        try:
            if getattr(self, "_net", None):
                self._lib.adaf_mobilenetv2_destroy(self._net)
                self._net = None
        except Exception:
            pass

    def load(self, params):
        """params: neutral name ('stem.weight', 'b3.dw.bn.running_var', ...) -> tensor on this device."""
        keep = []
        for name, t in params.items():
            L.need_gpu_f32(t)
            t = t.detach().contiguous()
            keep.append(t)
            L.check(self._lib.adaf_mobilenetv2_set_param(self._net, name.encode(), L.ptr(t), t.numel()), self._h)
        L.check(self._lib.adaf_mobilenetv2_finalize(self._net, L.stream_ptr()), self._h)
        del keep

    def set_fusion(self, on):
        """Expand 1x1 -> depthwise 3x3 in one kernel for the high-resolution blocks (default on)."""
        # bit 0: fused kernels on; bit 2 (value 4): one frame chunk at a time instead of two side by side; bit 3 (value 8):
        # expand -> depthwise kernel + project launch instead of the whole-block kernel of b3 / b5 / b6 (A/B switches)
        L.check(self._lib.adaf_mobilenetv2_set_fusion(self._net, int(on)), self._h)

    def forward(self, frames_nhwc4, tsm_segments=0, tsm_div=8, want_vec=True):
        """(N,S,S,4) -> featmap (N,S/32,S/32,1280) NHWC, featvec (N,1280) or None."""
        L.need_gpu_f32(frames_nhwc4)
        x = frames_nhwc4.contiguous()
        n, s = x.shape[0], x.shape[1]
        fs = s
        for _ in range(5):          # stem + four stride-2 blocks, each a 3x3 / pad 1 / stride 2 window
            fs = (fs - 1) // 2 + 1
        fmap = torch.empty((n, fs, fs, 1280), device=x.device, dtype=torch.float32)
        fvec = torch.empty((n, 1280), device=x.device, dtype=torch.float32) if want_vec else None
        need = self._lib.adaf_mobilenetv2_workspace_bytes(self._net, n, s, int(tsm_segments))
        ws = self._scratch.get(need)     # per stream: consecutive batches' glancer passes may overlap
        L.check(self._lib.adaf_mobilenetv2_forward(self._net, L.ptr(x), n, s, int(tsm_segments), int(tsm_div), L.ptr(fmap),
                                                   L.ptr(fvec), 1280, L.ptr(ws), need, L.stream_ptr()), self._h)
        return fmap, fvec


def ingest_u8(clips_hwc_u8, frames, mean, std):
    """Loader tail fused (ACT/ops/transforms.py:305-336,64-77): (B, H, W, T*3) uint8 stacked clips ->
    (B*T, H, W, 4) fp32 normalised pixel-major frames (lane 3 = 0)."""
    if not clips_hwc_u8.is_cuda or clips_hwc_u8.dtype != torch.uint8:
        raise L.AdafError("ingest_u8: a uint8 tensor on the GPU is required")
    L.on_current_device(clips_hwc_u8)
    x = clips_hwc_u8.contiguous()
    b, hh, ww, c = x.shape
    if c != 3 * frames:
        raise ValueError("ingest_u8: last dim %d != 3*frames" % c)
    out = torch.empty((b * frames, hh, ww, 4), device=x.device, dtype=torch.float32)
    m = (C.c_float * 3)(*[float(v) for v in mean])
    s = (C.c_float * 3)(*[float(v) for v in std])
    h = _h(x)
    L.check(L.load_library().adaf_ingest_u8_f32(h, L.ptr(x), b, int(frames), hh, ww, m, s, L.ptr(out), L.stream_ptr()), h)
    return out


def crop_gather_nhwc4(frames_nhwc4, actions, patch, frames_per_action=1, return_coords=False):
    """get_patch from pixel-major (N,H,W,4) frames -> (N,P,P,4)."""
    L.need_gpu_f32(frames_nhwc4, actions)
    x = frames_nhwc4.contiguous()
    actions = actions.contiguous()
    n, hh, ww, _ = x.shape
    p = int(patch)
    out = torch.empty((n, p, p, 4), device=x.device, dtype=torch.float32)
    coords = torch.empty((actions.shape[0], 2), device=x.device, dtype=torch.int32) if return_coords else None
    h = _h(x)
    L.check(L.load_library().adaf_crop_gather_nhwc4_f32(h, L.ptr(x), n, hh, ww, L.ptr(actions), actions.shape[0],
                                                        int(frames_per_action), p, L.ptr(out), L.ptr(coords),
                                                        L.stream_ptr()), h)
    return (out, coords) if return_coords else out


def _out_shape(n, c, oh, ow, layout):
    return {LAYOUT_NCHW: (n, c, oh, ow), LAYOUT_NHWC: (n, oh, ow, c), LAYOUT_NHWC4: (n, oh, ow, 4)}[layout]


def crop_resize(frames, actions, patch, size=None, frames_per_action=1, layout=LAYOUT_NCHW, return_coords=False):
    """(y, x, size) -> patch x patch: window of `size` pixels at floor(action * (H - size)), bilinear (align_corners=False)
    resample to patch x patch (include/adafocus.h adaf_crop_resize_f32).  frames (N,3,H,W) planar or (N,H,W,4) pixel-major;
    size: None (= patch: the plain gather), an int, or an int32 tensor (M,) of per-action sizes."""
    L.need_gpu_f32(frames, actions)
    frames = frames.contiguous()
    actions = actions.contiguous()
    in4 = frames.shape[-1] == 4 and frames.shape[1] != 3
    if in4:
        n, hh, ww, _ = frames.shape
        c = 3
    else:
        n, c, hh, ww = frames.shape
    p = int(patch)
    sizes, sdef = None, p
    if isinstance(size, torch.Tensor):
        if size.dtype != torch.int32 or not size.is_cuda or size.numel() != actions.shape[0]:
            raise ValueError("crop_resize: per-action sizes must be an int32 GPU tensor of shape (n_actions,)")
        sizes = size.contiguous()
    elif size is not None:
        sdef = int(size)
    out = torch.empty(_out_shape(n, c, p, p, layout), device=frames.device, dtype=torch.float32)
    coords = torch.empty((actions.shape[0], 2), device=frames.device, dtype=torch.int32) if return_coords else None
    h = _h(frames)
    L.check(L.load_library().adaf_crop_resize_f32(h, L.ptr(frames), LAYOUT_NHWC4 if in4 else LAYOUT_NCHW, n, c, hh, ww, L.ptr(actions),
                                                  actions.shape[0], int(frames_per_action), L.ptr(sizes), sdef, p, L.ptr(out), layout,
                                                  L.ptr(coords), L.stream_ptr()), h)
    return (out, coords) if return_coords else out


def resize_nearest(frames, out_hw, layout=LAYOUT_NCHW):
    """F.interpolate(frames, out_hw) with the default nearest mode (ACT/main_dist.py:331-332), bit-exact.
    frames (N,C,H,W) planar or (N,H,W,4) pixel-major -> `layout`."""
    L.need_gpu_f32(frames)
    frames = frames.contiguous()
    in4 = frames.shape[-1] == 4 and frames.shape[1] != 3
    if in4:
        n, hh, ww, _ = frames.shape
        c = 3
    else:
        n, c, hh, ww = frames.shape
    oh, ow = (int(out_hw), int(out_hw)) if isinstance(out_hw, int) else (int(out_hw[0]), int(out_hw[1]))
    out = torch.empty(_out_shape(n, c, oh, ow, layout), device=frames.device, dtype=torch.float32)
    h = _h(frames)
    L.check(L.load_library().adaf_resize_nearest_f32(h, L.ptr(frames), LAYOUT_NHWC4 if in4 else LAYOUT_NCHW, n, c, hh, ww, oh, ow,
                                                     L.ptr(out), layout, L.stream_ptr()), h)
    return out


# ---- N2: half-precision storage ---------------------------------------------------------------------------------
def _need_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise L.AdafError("adafocus_amd runs on MI355X only: got a %s tensor" % t.device)
    L.on_current_device(*tensors)


def cast(x, dtype):
    """fp32 <-> fp16 element-wise conversion on the HIP side (round-to-nearest-even)."""
    _need_gpu(x)
    x = x.contiguous()
    to16 = dtype == torch.float16
    if x.dtype not in (torch.float32, torch.float16) or (x.dtype == torch.float16) == to16:
        raise ValueError("cast: fp32 -> fp16 or fp16 -> fp32")
    out = torch.empty(x.shape, device=x.device, dtype=dtype)
    h = _h(x)
    L.check(L.load_library().adaf_cast_f32_f16(h, L.ptr(x), x.numel(), L.ptr(out), 1 if to16 else 0, L.stream_ptr()), h)
    return out


def pack_conv_weight_f16(w_oihw, cin_pad=None):
    """OIHW fp32 -> OHWI fp16 (input channels zero-padded to a multiple of 8)."""
    L.need_gpu_f32(w_oihw)
    w = w_oihw.contiguous()
    co, ci, kh, kw = w.shape
    cp = cin_pad or ((ci + 7) // 8 * 8)
    out = torch.empty((co, kh, kw, cp), device=w.device, dtype=torch.float16)
    h = _h(w)
    L.check(L.load_library().adaf_pack_conv_weight_f16(h, L.ptr(w), co, ci, kh, kw, cp, L.ptr(out), L.stream_ptr()), h)
    return out


def conv2d_bn_act_f16(x, w_ohwi, scale=None, bias=None, residual=None, stride=1, pad=0, act=ACT_NONE, out_dtype=torch.float16,
                      tile=0):
    """adaf_conv2d_bn_act_f16: x (N,H,W,Cin) fp16 with fp16 weights (or fp32 x / w with an fp16 store), fp32 accumulate and
    epilogue, `out_dtype` store."""
    _need_gpu(x, w_ohwi, scale, bias, residual)
    x = x.contiguous()
    w_ohwi = w_ohwi.contiguous()
    if x.dtype != w_ohwi.dtype:
        raise ValueError("conv2d_bn_act_f16: x and w must share a dtype")
    n, hh, ww, cin = x.shape
    cout, kh, kw, cin_w = w_ohwi.shape
    if cin_w != cin:
        raise ValueError("conv2d_bn_act_f16: x has %d channels, weight expects %d" % (cin, cin_w))
    oh = (hh + 2 * pad - kh) // stride + 1
    ow = (ww + 2 * pad - kw) // stride + 1
    out = torch.empty((n, oh, ow, cout), device=x.device, dtype=out_dtype)
    if residual is not None:
        if residual.dtype != torch.float16:
            raise ValueError("conv2d_bn_act_f16: the residual is fp16")
        residual = residual.contiguous()
    p = L.ConvParams(n=n, h=hh, w=ww, cin=cin, cout=cout, kh=kh, kw=kw, stride=stride, pad=pad, act=act, tsm_segments=0,
                     tsm_div=8, ldx=0, ldo=0, ldr=0, tile=tile)
    h = _h(x)
    L.check(L.load_library().adaf_conv2d_bn_act_f16(h, C.byref(p), L.ptr(x), DTYPE_F16 if x.dtype == torch.float16 else DTYPE_F32,
                                                    L.ptr(w_ohwi), L.ptr(scale), L.ptr(bias), L.ptr(residual), L.ptr(out),
                                                    DTYPE_F16 if out_dtype == torch.float16 else DTYPE_F32, L.stream_ptr()), h)
    return out


def dwconv3x3_bn_act_f16(x, w_33c, scale, bias, stride=1, act=ACT_RELU6):
    """Depthwise 3x3 on fp16 activations (N,H,W,C); taps, BN affine and the sum in fp32."""
    _need_gpu(x, w_33c, scale, bias)
    if x.dtype != torch.float16:
        raise ValueError("dwconv3x3_bn_act_f16: fp16 activations expected")
    x = x.contiguous()
    n, hh, ww, c = x.shape
    out = torch.empty((n, (hh - 1) // stride + 1, (ww - 1) // stride + 1, c), device=x.device, dtype=torch.float16)
    h = _h(x)
    L.check(L.load_library().adaf_dwconv3x3_bn_act_f16(h, L.ptr(x), n, hh, ww, c, stride, L.ptr(w_33c.contiguous()),
                                                       L.ptr(scale.contiguous()), L.ptr(bias.contiguous()), act, L.ptr(out),
                                                       L.stream_ptr()), h)
    return out


# ---- EfficientNet building blocks and network (BASELINE config 5; parity unpinned -- see include/adafocus.h) ------------
def _dt(t):
    if t.dtype == torch.float16:
        return DTYPE_F16
    if t.dtype == torch.float32:
        return DTYPE_F32
    raise L.AdafError("adafocus_amd: fp32 or fp16 storage expected, got %s" % t.dtype)


def pack_dw_weight_kxk(w_c1kk):
    """PyTorch depthwise filter (C,1,K,K) -> (K*K, C)."""
    L.need_gpu_f32(w_c1kk)
    w = w_c1kk.contiguous()
    c, k = w.shape[0], w.shape[-1]
    out = torch.empty((k * k, c), device=w.device, dtype=torch.float32)
    h = _h(w)
    L.check(L.load_library().adaf_pack_dw_weight_kxk_f32(h, L.ptr(w), c, k, L.ptr(out), L.stream_ptr()), h)
    return out


def dwconv_same_bn_act(x, w_kkc, scale, bias, k, stride=1, act=ACT_SWISH, want_pool=False):
    """Depthwise k x k with TensorFlow-SAME padding + BN affine + activation.  x (N,H,W,C) fp32 | fp16 NHWC ->
    (N,ceil(H/s),ceil(W/s),C) in x's dtype [, squeeze mean (N,C) fp32]."""
    _need_gpu(x)
    L.need_gpu_f32(w_kkc, scale, bias)
    x = x.contiguous()
    n, hh, ww, c = x.shape
    oh, ow = -(-hh // stride), -(-ww // stride)
    out = torch.empty((n, oh, ow, c), device=x.device, dtype=x.dtype)
    lib = L.load_library()
    pool = ws = None
    need = 0
    if want_pool:
        pool = torch.empty((n, c), device=x.device, dtype=torch.float32)
        need = lib.adaf_dwconv_same_workspace_bytes(n, hh, ww, c, int(k), int(stride), _dt(x))
        ws = torch.empty(max(need, 16), device=x.device, dtype=torch.uint8)
    h = _h(x)
    L.check(lib.adaf_dwconv_same_bn_act(h, L.ptr(x), _dt(x), n, hh, ww, c, int(k), int(stride), L.ptr(w_kkc.contiguous()),
                                        L.ptr(scale.contiguous()), L.ptr(bias.contiguous()), int(act), L.ptr(out), L.ptr(pool),
                                        L.ptr(ws), need, L.stream_ptr()), h)
    return (out, pool) if want_pool else out


def se_gate(pool_mean, w_reduce, b_reduce, w_expand, b_expand):
    """sigmoid(W_e swish(W_r m + b_r) + b_e): pool_mean (N,C), W_r (SQ,C), W_e (C,SQ) -> gate (N,C)."""
    L.need_gpu_f32(pool_mean, w_reduce, b_reduce, w_expand, b_expand)
    n, c = pool_mean.shape
    sq = w_reduce.shape[0]
    gate = torch.empty((n, c), device=pool_mean.device, dtype=torch.float32)
    h = _h(pool_mean)
    L.check(L.load_library().adaf_se_gate_f32(h, L.ptr(pool_mean.contiguous()), n, c, L.ptr(w_reduce.reshape(sq, c).contiguous()),
                                              L.ptr(b_reduce.contiguous()), sq, L.ptr(w_expand.reshape(c, sq).contiguous()),
                                              L.ptr(b_expand.contiguous()), L.ptr(gate), L.stream_ptr()), h)
    return gate


def conv1x1_gated_bn(x, gate, w, scale=None, bias=None, residual=None):
    """x (N,H,W,Cin) fp32 | fp16, gate (N,Cin) fp32 or None, w (Cout,Cin) in x's dtype -> (N,H,W,Cout) in x's dtype."""
    _need_gpu(x, w)
    x = x.contiguous()
    n, hh, ww, cin = x.shape
    cout = w.shape[0]
    if w.dtype != x.dtype or (residual is not None and residual.dtype != x.dtype):
        raise L.AdafError("conv1x1_gated_bn: w / residual must have x's dtype")
    out = torch.empty((n, hh, ww, cout), device=x.device, dtype=x.dtype)
    h = _h(x)
    L.check(L.load_library().adaf_conv1x1_gated_bn(h, L.ptr(x), _dt(x), n, hh * ww, cin, L.ptr(gate.contiguous() if gate is not None else None),
                                                   L.ptr(w.reshape(cout, cin).contiguous()), cout,
                                                   L.ptr(scale.contiguous() if scale is not None else None),
                                                   L.ptr(bias.contiguous() if bias is not None else None),
                                                   L.ptr(residual.contiguous() if residual is not None else None), L.ptr(out),
                                                   L.stream_ptr()), h)
    return out


class EffNetNet:
    """adaf_effnet: EfficientNet feature extractor (MBConv + squeeze-and-excite + swish; csrc/effnet.hip)."""

    def __init__(self, device, width, depth):
        self.device = torch.device(device)
        self._h = L.handle(self.device)
        self._lib = L.load_library()
        net = C.c_void_p()
        L.check(self._lib.adaf_effnet_create(self._h, C.c_float(width), C.c_float(depth), C.byref(net)), self._h)
        self._net = net
        self._scratch = _StreamScratch(self.device)
        self.feature_dim = self._lib.adaf_effnet_feature_dim(net)
        self.dtype = DTYPE_F32
        self.pad_size = 0          # default image size the SAME padding is computed for (0 = the input's own); the owner sets it

    def __del__(self):
        try:
            if getattr(self, "_net", None):
                self._lib.adaf_effnet_destroy(self._net)
                self._net = None
        except Exception:
            pass

    def blocks(self):
        out = []
        for i in range(self._lib.adaf_effnet_block_count(self._net)):
            info = (C.c_int * 8)()
            L.check(self._lib.adaf_effnet_block_info(self._net, i, info), self._h)
            out.append(dict(zip(("k", "stride", "expand", "cin", "cout", "hid", "sq", "stem"), list(info))))
        return out

    def set_dtype(self, dtype):
        code = {"f32": DTYPE_F32, "f16": DTYPE_F16}.get(dtype, dtype)
        L.check(self._lib.adaf_effnet_set_dtype(self._net, int(code)), self._h)
        self.dtype = int(code)

    def set_fusion(self, on):
        """fp16 storage: whole-image MBConv kernels for the stride-1 blocks on maps up to 9 x 9 (default on); off = the four-launch plan."""
        L.check(self._lib.adaf_effnet_set_fusion(self._net, int(bool(on))), self._h)

    def whole_blocks(self, size, pad_size=None):
        """MBConv blocks of a forward at this input size that run as one launch each (csrc/mbconv_whole.hip)."""
        return int(self._lib.adaf_effnet_whole_blocks(self._net, int(size), int(self.pad_size if pad_size is None else pad_size)))

    def fused_expand_blocks(self, size, pad_size=None):
        """MBConv blocks of a forward at this input size whose expand conv runs inside the depthwise launch (fp16 storage; effnet.hip XN > 0)."""
        return int(self._lib.adaf_effnet_fused_expand_blocks(self._net, int(size), int(self.pad_size if pad_size is None else pad_size)))

    def load(self, params):
        keep = []
        for name, t in params.items():
            L.need_gpu_f32(t)
            t = t.detach().contiguous()
            keep.append(t)
            L.check(self._lib.adaf_effnet_set_param(self._net, name.encode(), L.ptr(t), t.numel()), self._h)
        L.check(self._lib.adaf_effnet_finalize(self._net, L.stream_ptr()), self._h)
        del keep

    def out_size(self, size, pad_size=0):
        """Spatial size of the feature map for an input of `size` (SAME padding computed for pad_size or size)."""
        ps = pad_size or size

        def step(hw, ps, k, s):
            tot = max((-(-ps // s) - 1) * s + k - ps, 0)
            return (hw + tot - k) // s + 1, -(-ps // s)
        hw, ps = step(size, ps, 3, 2)
        for b in self.blocks():
            hw, ps = step(hw, ps, b["k"], b["stride"])
        return hw

    def forward(self, frames_nhwc4, pad_size=0, want_map=False, want_vec=True, out=None):
        """(N,S,S,4) fp32 -> (featmap (N,s,s,F) fp32 NHWC or None, featvec (N,F) or None).  `out`: write the vector into
        this (N, >= F) fp32 view (row stride = its stride(0))."""
        L.need_gpu_f32(frames_nhwc4)
        x = frames_nhwc4.contiguous()
        n, s = x.shape[0], x.shape[1]
        f = self.feature_dim
        fs = self.out_size(s, pad_size)
        fmap = torch.empty((n, fs, fs, f), device=x.device, dtype=torch.float32) if want_map else None
        fvec, ld = None, f
        if out is not None:
            fvec, ld = out, out.stride(0)
        elif want_vec:
            fvec = torch.empty((n, f), device=x.device, dtype=torch.float32)
        need = self._lib.adaf_effnet_workspace_bytes(self._net, n, s, int(pad_size))
        ws = self._scratch.get(need)
        L.check(self._lib.adaf_effnet_forward(self._net, L.ptr(x), n, s, int(pad_size), -1, None, L.ptr(fmap), L.ptr(fvec), int(ld),
                                              L.ptr(ws), need, L.stream_ptr()), self._h)
        return fmap, fvec

    def forward_blocks(self, frames_nhwc4, upto, pad_size=None):
        """Output of the first `upto` MBConv blocks (0 = the stem) as (N,h,w,c) in the storage dtype (tests).  pad_size: the image
        size the SAME padding is computed for; None = this engine's default (set by its owner), 0 = the input's own size."""
        L.need_gpu_f32(frames_nhwc4)
        x = frames_nhwc4.contiguous()
        n, s = x.shape[0], x.shape[1]
        pad_size = self.pad_size if pad_size is None else pad_size
        ps = pad_size or s

        def step(hw, ps, k, st):
            tot = max((-(-ps // st) - 1) * st + k - ps, 0)
            return (hw + tot - k) // st + 1, -(-ps // st)
        hw, ps = step(s, ps, 3, 2)
        blocks = self.blocks()
        c = blocks[0]["stem"]
        for b in blocks[:upto]:
            hw, ps = step(hw, ps, b["k"], b["stride"])
            c = b["cout"]
        dt = torch.float16 if self.dtype == DTYPE_F16 else torch.float32
        out = torch.empty((n, hw, hw, c), device=x.device, dtype=dt)
        need = self._lib.adaf_effnet_workspace_bytes(self._net, n, s, int(pad_size))
        ws = self._scratch.get(need)
        L.check(self._lib.adaf_effnet_forward(self._net, L.ptr(x), n, s, int(pad_size), int(upto), L.ptr(out), None, None, 0,
                                              L.ptr(ws), need, L.stream_ptr()), self._h)
        return out
