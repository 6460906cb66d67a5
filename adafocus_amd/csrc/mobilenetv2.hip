// MobileNetV2 feature extractor (the glancer) as one object on the
// conv engine + the depthwise kernel.  ACT/models/mobilenet.py:71-148 and STH/models/mobilenetv2.py
// (with the temporal shift of STH/models/gfv_net.py:238-241 on the expand conv of residual blocks).
//
// Per inverted-residual block: expand 1x1 (+BN+ReLU6) on the MFMA engine, depthwise 3x3 (+BN+ReLU6)
// on the VALU kernel, project 1x1 (+BN, + residual in the epilogue) on the MFMA engine.  Frames are
// processed in chunks so the 6x-expanded intermediates stay within a bounded workspace.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "adaf_internal.h"

struct MbConv {
    std::string name;   // "stem", "b3.expand", "b3.dw", "b3.project", "head"
    int cin, cout, k, stride, groups;
    int cin_pad;
    float* w = nullptr;
    float* scale = nullptr;
    float* bias = nullptr;
};

struct MbBlock {
    int inp, oup, stride, t;
    int expand, dw, project;   // indices into convs (-1 = absent)
};

struct adaf_mobilenetv2 {
    adaf_handle* h = nullptr;
    std::map<std::string, std::pair<const float*, size_t>> params;
    std::vector<MbConv> convs;
    std::vector<MbBlock> blocks;
    int stem = 0, head = 0;
    bool fuse = true;       // expand -> depthwise in one kernel where the shape allows (mbconv.hip)
    bool whole = true;      // ... and the whole stride-1 block (expand -> depthwise -> project + identity) where that shape allows
    bool finalized = false;
    // Two frame chunks travel through the network side by side (the second on this library-owned stream, forked from and
    // joined to the caller's stream by events -- still fully asynchronous): the tail's launches are 30-100 us each and
    // leave the device half empty on their own; a neighbour fills the ramps and tails.
    // One helper stream + event pair PER CALLER STREAM: forwards issued from different streams (pipelined batches,
    // bench --streams) must not serialise their second chunks on one shared helper.
    AdafAuxPool aux;        // (adaf_internal.h: LRU over caller streams, mutex-guarded)
    bool pair = true;
};

namespace {

int mfail(adaf_handle* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    return code;
}

const int kSetting[7][4] = {{1, 16, 1, 1}, {6, 24, 2, 2}, {6, 32, 3, 2}, {6, 64, 4, 2}, {6, 96, 3, 1}, {6, 160, 3, 2}, {6, 320, 1, 1}};

inline int cdiv_out(int in, int k, int stride, int pad) { return (in + 2 * pad - k) / stride + 1; }

void build(adaf_mobilenetv2* net) {
    net->convs.clear();
    net->blocks.clear();
    net->stem = 0;
    net->convs.push_back({"stem", 3, 32, 3, 2, 1, 4});
    int cin = 32, bi = 1;
    for (auto& s : kSetting) {
        for (int i = 0; i < s[2]; ++i, ++bi) {
            MbBlock b{cin, s[1], i == 0 ? s[3] : 1, s[0], -1, -1, -1};
            const int hid = cin * s[0];
            char nm[32];
            if (s[0] != 1) {
                snprintf(nm, sizeof(nm), "b%d.expand", bi);
                b.expand = (int)net->convs.size();
                net->convs.push_back({nm, cin, hid, 1, 1, 1, cin});
            }
            snprintf(nm, sizeof(nm), "b%d.dw", bi);
            b.dw = (int)net->convs.size();
            net->convs.push_back({nm, hid, hid, 3, b.stride, hid, hid});
            snprintf(nm, sizeof(nm), "b%d.project", bi);
            b.project = (int)net->convs.size();
            net->convs.push_back({nm, hid, s[1], 1, 1, 1, hid});
            net->blocks.push_back(b);
            cin = s[1];
        }
    }
    net->head = (int)net->convs.size();
    net->convs.push_back({"head", cin, 1280, 1, 1, 1, cin});
}

// floats per frame of the four chunk buffers (block in/out ping-pong, expanded, depthwise out)
void slab_sizes(const adaf_mobilenetv2* net, int size, size_t* io, size_t* ex, size_t* dw) {
    int hw = cdiv_out(size, 3, 2, 1);
    *io = (size_t)hw * hw * 32;
    *ex = 0;
    *dw = 0;
    for (auto& b : net->blocks) {
        const int hid = b.inp * b.t;
        const int ohw = cdiv_out(hw, 3, b.stride, 1);
        if (b.t != 1 && (size_t)hw * hw * hid > *ex) *ex = (size_t)hw * hw * hid;
        if ((size_t)ohw * ohw * hid > *dw) *dw = (size_t)ohw * ohw * hid;
        if ((size_t)ohw * ohw * b.oup > *io) *io = (size_t)ohw * ohw * b.oup;
        hw = ohw;
    }
    if ((size_t)hw * hw * 32 > *ex) *ex = (size_t)hw * hw * 32;
}

int chunk_size() {   // frames per pass through the network (ADAF_MBV2_CHUNK overrides, for tuning)
    const int c = adaf_options().mbv2_chunk;
    return c;
}

int chunk_frames(int n, int tsm_segments) {
    const int kChunk = chunk_size();
    int c = n < kChunk ? n : kChunk;
    // a batch that fits one chunk but is large enough for two halves to fill the machine each (>= 256 frames) travels as a PAIR of half
    // chunks on the two streams like any larger batch does: 512 frames (the Something-Something glancer at 64 clips x 8) 5.15 -> 4.9 ms
    // (tools/glancer_chunk_ab.py); a frame's arithmetic does not depend on the chunk it travels in
    const bool halve = n <= kChunk && n >= 512;
    if (halve) c = (n + 1) / 2;
    if (tsm_segments > 0) {
        if (halve) c += (tsm_segments - c % tsm_segments) % tsm_segments;     // whole clips per chunk: round the half UP
        else c -= c % tsm_segments;
        if (c <= 0) c = tsm_segments;
    }
    return c;
}

int run_conv(adaf_mobilenetv2* net, const MbConv& L, const float* in, int n, int hh, int ww, int act, const float* res,
             float* out, int tsm_T, int tsm_div, hipStream_t st) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    const int oh = cdiv_out(hh, L.k, L.stride, L.k / 2), ow = cdiv_out(ww, L.k, L.stride, L.k / 2);
    a.x = in; a.w = L.w; a.scale = L.scale; a.bias = L.bias; a.res = res; a.out = out;
    a.M = n * oh * ow; a.N = L.cout; a.K = L.k * L.k * L.cin_pad;
    a.cin = L.cin_pad; a.H = hh; a.W = ww; a.OH = oh; a.OW = ow; a.KH = a.KW = L.k; a.stride = L.stride; a.pad = L.k / 2;
    a.ldx = L.cin_pad; a.ldo = L.cout; a.ldr = L.cout; a.act = act;
    a.tsm_T = tsm_T; a.tsm_fold = tsm_T > 0 ? L.cin / tsm_div : 0; a.tsm_hw = hh * ww;
    a.zeros = net->h->zeros;
    a.vec_epi = (L.cout % 4 == 0) ? 1 : 0;
    return adaf_launch_conv_gemm(a, 0, net->h->cus, st) > 0 ? ADAF_OK : ADAF_E_LAUNCH;
}

}  // namespace

extern "C" {

int adaf_mobilenetv2_create(adaf_handle* h, adaf_mobilenetv2** out) {
    if (!h || !out) return ADAF_E_BADARG;
    adaf_mobilenetv2* net = new adaf_mobilenetv2();
    net->h = h;
    build(net);
    *out = net;
    return ADAF_OK;
}

int adaf_mobilenetv2_destroy(adaf_mobilenetv2* net) {
    if (!net) return ADAF_OK;
    net->aux.destroy();
    for (auto& L : net->convs) {
        if (L.w) (void)hipFree(L.w);
        if (L.scale) (void)hipFree(L.scale);
        if (L.bias) (void)hipFree(L.bias);
    }
    delete net;
    return ADAF_OK;
}

int adaf_mobilenetv2_set_fusion(adaf_mobilenetv2* net, int on) {
    if (!net) return ADAF_E_BADARG;
    net->fuse = (on & 1) != 0;
    net->pair = (on & 4) == 0;     // bit 2 set: one frame chunk at a time (A/B)
    net->whole = (on & 8) == 0;    // bit 3 set: expand -> depthwise kernel + project launch instead of the whole-block kernel (A/B)
    return ADAF_OK;
}

int adaf_mobilenetv2_set_param(adaf_mobilenetv2* net, const char* name, const float* dev_ptr, size_t numel) {
    if (!net || !name || !dev_ptr) return ADAF_E_BADARG;
    net->params[name] = std::make_pair(dev_ptr, numel);
    net->finalized = false;
    return ADAF_OK;
}

int adaf_mobilenetv2_finalize(adaf_mobilenetv2* net, void* stream) {
    if (!net) return ADAF_E_BADARG;
    adaf_handle* h = net->h;
    hipStream_t st = (hipStream_t)stream;
    auto get = [&](const std::string& key, size_t numel, const float** p) -> int {
        auto it = net->params.find(key);
        if (it == net->params.end()) return mfail(h, ADAF_E_STATE, "mobilenetv2: missing parameter '%s'", key.c_str());
        if (it->second.second != numel)
            return mfail(h, ADAF_E_BADARG, "mobilenetv2: '%s' has %zu elements, expected %zu", key.c_str(), it->second.second, numel);
        *p = it->second.first;
        return ADAF_OK;
    };
    for (auto& L : net->convs) {
        const float *w, *g, *b, *m, *v;
        int rc;
        const bool dw = L.groups > 1;
        const size_t wn_in = dw ? (size_t)L.cout * 9 : (size_t)L.cout * L.cin * L.k * L.k;
        if ((rc = get(L.name + ".weight", wn_in, &w))) return rc;
        if ((rc = get(L.name + ".bn.weight", L.cout, &g))) return rc;
        if ((rc = get(L.name + ".bn.bias", L.cout, &b))) return rc;
        if ((rc = get(L.name + ".bn.running_mean", L.cout, &m))) return rc;
        if ((rc = get(L.name + ".bn.running_var", L.cout, &v))) return rc;
        const size_t wn = dw ? (size_t)9 * L.cout : (size_t)L.cout * L.k * L.k * L.cin_pad;
        if (!L.w && hipMalloc(reinterpret_cast<void**>(&L.w), wn * sizeof(float)) != hipSuccess) return mfail(h, ADAF_E_NOMEM, "mobilenetv2: hipMalloc");
        if (!L.scale && hipMalloc(reinterpret_cast<void**>(&L.scale), L.cout * sizeof(float)) != hipSuccess) return mfail(h, ADAF_E_NOMEM, "mobilenetv2: hipMalloc");
        if (!L.bias && hipMalloc(reinterpret_cast<void**>(&L.bias), L.cout * sizeof(float)) != hipSuccess) return mfail(h, ADAF_E_NOMEM, "mobilenetv2: hipMalloc");
        if (dw) adaf_launch_pack_dw_weight(w, L.cout, L.w, st);
        else adaf_launch_pack_weight(w, L.cout, L.cin, L.k, L.k, L.cin_pad, L.w, st);
        adaf_launch_fold_bn(g, b, m, v, 1e-5f, L.cout, L.scale, L.bias, st);
    }
    hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess) return mfail(h, ADAF_E_LAUNCH, "mobilenetv2 finalize: %s", hipGetErrorString(e));
    net->aux.prepare(4);        // helper streams exist before the first forward (which may be captured into a HIP graph)
    net->finalized = true;
    return ADAF_OK;
}

size_t adaf_mobilenetv2_workspace_bytes(const adaf_mobilenetv2* net, int n, int size, int tsm_segments) {
    if (!net || n <= 0 || size <= 0) return 0;
    size_t io, ex, dw;
    slab_sizes(net, size, &io, &ex, &dw);
    const int chunk = chunk_frames(n, tsm_segments);
    return (size_t)chunk * (2 * io + ex + dw) * sizeof(float) * (n > chunk ? 2 : 1);     // two chunks in flight
}

int adaf_mobilenetv2_forward(adaf_mobilenetv2* net, const float* frames_nhwc4, int n, int size, int tsm_segments,
                             int tsm_div, float* featmap, float* featvec, int ldvec, void* ws, size_t ws_bytes,
                             void* stream) {
    if (!net) return ADAF_E_BADARG;
    adaf_handle* h = net->h;
    if (!net->finalized) return mfail(h, ADAF_E_STATE, "mobilenetv2: finalize() has not been called");
    if (!frames_nhwc4 || !featmap || !ws) return mfail(h, ADAF_E_BADARG, "mobilenetv2: null pointer");
    if (n <= 0 || size < 32) return mfail(h, ADAF_E_BADARG, "mobilenetv2: need n > 0 and size >= 32");
    if (tsm_segments > 0 && (n % tsm_segments || tsm_div <= 0)) return mfail(h, ADAF_E_BADARG, "mobilenetv2: n %% tsm_segments != 0");
    if (featvec && (ldvec < 1280 || ldvec % 4)) return mfail(h, ADAF_E_LAYOUT, "mobilenetv2: ldvec >= 1280 and %% 4 == 0 required");
    if (ws_bytes < adaf_mobilenetv2_workspace_bytes(net, n, size, tsm_segments)) return mfail(h, ADAF_E_NOMEM, "mobilenetv2: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    size_t io, ex, dws;
    slab_sizes(net, size, &io, &ex, &dws);
    const int chunk = chunk_frames(n, tsm_segments);
    float* bufA = static_cast<float*>(ws);
    float* bufB = bufA + (size_t)chunk * io;
    float* bufE = bufB + (size_t)chunk * io;
    float* bufD = bufE + (size_t)chunk * ex;

    // one chunk of frames through the whole network on stream `st` with its own quarter of the workspace
    auto run_chunk = [&](int f0, int nc, float* bufA, float* bufB, float* bufE, float* bufD, hipStream_t st) -> int {
        int hw = cdiv_out(size, 3, 2, 1);
        int rc;
        float* cur = bufA;
        float* nxt = bufB;
        size_t first_block = 0;
        const MbBlock& b1 = net->blocks[0];
        if (net->fuse && b1.t == 1 && b1.inp == 32 && b1.oup == 16 && b1.stride == 1) {
            // stem + block 1 in one kernel: the two 32-channel maps at the stem's resolution never reach HBM
            const MbConv &S = net->convs[net->stem], &D = net->convs[b1.dw], &P = net->convs[b1.project];
            MbStemArgs sa;
            memset(&sa, 0, sizeof(sa));
            sa.x = frames_nhwc4 + (size_t)f0 * size * size * 4; sa.n = nc; sa.S = size; sa.H1 = hw;
            sa.ws = S.w; sa.ss = S.scale; sa.bs = S.bias; sa.wd = D.w; sa.sd = D.scale; sa.bd = D.bias;
            sa.wp = P.w; sa.sp = P.scale; sa.bp = P.bias; sa.out = bufA; sa.zeros = net->h->zeros;
            adaf_launch_mb_stem_b1(sa, net->h->cus, st);
            first_block = 1;
        } else if ((rc = run_conv(net, net->convs[net->stem], frames_nhwc4 + (size_t)f0 * size * size * 4, nc, size, size,
                                  ADAF_ACT_RELU6, nullptr, bufA, 0, 0, st)))
            return mfail(h, rc, "mobilenetv2: stem launch");
        for (size_t bidx = first_block; bidx < net->blocks.size(); ++bidx) {
            const MbBlock& b = net->blocks[bidx];
            const int hid = b.inp * b.t;
            const bool residual = b.stride == 1 && b.inp == b.oup;
            const float* dw_in = cur;
            const MbConv& D = net->convs[b.dw];
            bool fused = false;
            if (b.expand >= 0) {
                const MbConv& E = net->convs[b.expand];
                const bool tsm = tsm_segments > 0 && residual;      // STH/models/gfv_net.py:238-241
                fused = net->fuse && adaf_mb_expand_dw_ok(b.inp, hid, hw);
                // the 14 x 14 blocks (64 / 96 input channels): expand -> depthwise on strips (mbstrip.hip), the project conv stays on the engine
                const bool strip_xd = net->fuse && net->whole && !fused && adaf_mb_expand_dw_strip_ok(b.inp, hid, b.stride, hw, hw);
                if (strip_xd) fused = true;
                const float* ein = cur;
                int fused_T = 0, strip_T = 0;
                // the expand -> depthwise strip kernel, whose lanes load whole 4-channel groups of one shift kind (fold % 4 == 0: the 64- / 96-channel blocks):
                // the shift rides in their pixel loads (MbFuseArgs::tsm_T), nothing is materialised and the identity rows are the input itself
                // (the whole-block strip kernel of the 32-channel blocks has no register left for the frame offset: 252 of 256; its input stays materialised)
                const bool strip_shift = tsm && strip_xd && (b.inp / tsm_div) % 4 == 0;
                if (strip_shift) strip_T = tsm_segments;
                else if (tsm) {
                    if (!fused && (b.inp / tsm_div) % 4 == 0) fused_T = tsm_segments;   // shift fused into the operand load
                    else {   // fold not a multiple of 4 channels (24-channel block), or the fused kernel: materialise the shift once
                        float* sh = fused ? bufE : bufD;
                        adaf_launch_tshift(cur, nc, b.inp, hw * hw, tsm_segments, tsm_div, ADAF_LAYOUT_NHWC, sh, st);
                        ein = sh;
                    }
                }
                if (fused) {
                    MbFuseArgs fa;
                    memset(&fa, 0, sizeof(fa));
                    fa.x = ein; fa.n = nc; fa.H = hw; fa.W = hw; fa.cin = b.inp;
                    fa.we = E.w; fa.se = E.scale; fa.be = E.bias; fa.wd = D.w; fa.sd = D.scale; fa.bd = D.bias;
                    fa.out = bufD; fa.hid = hid; fa.OH = fa.OW = cdiv_out(hw, 3, b.stride, 1);
                    fa.zeros = net->h->zeros;
                    fa.tsm_T = strip_T; fa.tsm_fold = strip_T ? b.inp / tsm_div : 0;
                    const MbConv& P = net->convs[b.project];
                    if (net->whole && adaf_mb_block_ok(b.inp, hid, b.oup, b.stride, hw) && P.cin_pad == hid) {
                        // the whole block in one launch: neither expanded map reaches HBM
                        fa.wp = P.w; fa.sp = P.scale; fa.bp = P.bias; fa.cout = b.oup;
                        fa.res = residual ? cur : nullptr;
                        fa.out2 = nxt;
                        adaf_launch_mb_block(fa, st);
                        float* t = cur; cur = nxt; nxt = t;
                        hw = fa.OH;
                        continue;
                    }
                    if (strip_xd) adaf_launch_mb_expand_dw_strip(fa, st);
                    else adaf_launch_mb_expand_dw(fa, b.stride, st);
                } else {
                    if ((rc = run_conv(net, E, ein, nc, hw, hw, ADAF_ACT_RELU6, nullptr, bufE, fused_T, tsm_div, st)))
                        return mfail(h, rc, "mobilenetv2: expand launch");
                    dw_in = bufE;
                }
            }
            if (!fused) adaf_launch_dwconv3x3(dw_in, nc, hw, hw, hid, b.stride, D.w, D.scale, D.bias, ADAF_ACT_RELU6, bufD, st);
            const int ohw = cdiv_out(hw, 3, b.stride, 1);
            if ((rc = run_conv(net, net->convs[b.project], bufD, nc, ohw, ohw, ADAF_ACT_NONE, residual ? cur : nullptr, nxt, 0,
                               0, st)))
                return mfail(h, rc, "mobilenetv2: project launch");
            float* t = cur; cur = nxt; nxt = t;
            hw = ohw;
        }
        float* fm = featmap + (size_t)f0 * hw * hw * 1280;
        if ((rc = run_conv(net, net->convs[net->head], cur, nc, hw, hw, ADAF_ACT_RELU6, nullptr, fm, 0, 0, st)))
            return mfail(h, rc, "mobilenetv2: head launch");
        if (featvec) adaf_launch_avgpool(fm, nc, hw * hw, 1280, featvec + (size_t)f0 * ldvec, ldvec, st);
        return ADAF_OK;
    };
    const size_t per_chunk = (size_t)chunk * (2 * io + ex + dws);
    AdafAuxPool::Aux* ax = (net->pair && n > chunk) ? net->aux.get(st) : nullptr;
    const bool pair = ax != nullptr;
    float* base2 = static_cast<float*>(ws) + per_chunk;
    for (int f0 = 0; f0 < n; f0 += chunk) {
        const int nc = (n - f0) < chunk ? (n - f0) : chunk;
        int rc;
        if (pair && f0 + chunk < n) {
            const int f1 = f0 + chunk;
            const int nc1 = (n - f1) < chunk ? (n - f1) : chunk;
            (void)hipEventRecord(ax->ev_fork, st);
            (void)hipStreamWaitEvent(ax->stream, ax->ev_fork, 0);
            rc = run_chunk(f0, nc, bufA, bufB, bufE, bufD, st);
            if (!rc)
                rc = run_chunk(f1, nc1, base2, base2 + (size_t)chunk * io, base2 + (size_t)chunk * 2 * io,
                               base2 + (size_t)chunk * (2 * io + ex), ax->stream);
            // ALWAYS join, also when a launch failed after the fork: whatever the helper stream still has queued writes the
            // caller's workspace / outputs, and the caller may reuse them as soon as this call returns
            (void)hipEventRecord(ax->ev_join, ax->stream);
            (void)hipStreamWaitEvent(st, ax->ev_join, 0);
            if (rc) return rc;
            f0 = f1;
        } else if ((rc = run_chunk(f0, nc, bufA, bufB, bufE, bufD, st))) return rc;
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ADAF_OK : mfail(h, ADAF_E_LAUNCH, "mobilenetv2 forward: %s", hipGetErrorString(e));
}

}  // extern "C"
