// EfficientNet (B0..B7 by compound scaling; B3 = BASELINE.json config 5's local CNN) as one object: MBConv blocks with
// squeeze-and-excite, swish, 3x3 / 5x5 depthwise convolutions with TensorFlow-SAME (asymmetric) padding, BN eps 1e-3.
//
// PARITY UNPINNED.  The reference has no EfficientNet on a live path: STH/ops/models_ada.py:6,69-75 imports the third-party,
// un-vendored, un-pinned `efficientnet_pytorch` in dead AR-Net code, and STH/ops/net_flops_table.py:17,29 lists
// "efficientnet-b3": feature dim 1536, (1.80 GFLOPs, 12 M params).  What is built here is the published algorithm of that
// package (model.py MBConvBlock.forward / EfficientNet.extract_features, utils.py round_filters / round_repeats /
// Conv2dStaticSamePadding / MemoryEfficientSwish), restated on the CPU by oracle/ref_effnet.py.
//
// Launch plan per block (activations NHWC, fp32 or fp16 storage; accumulation, BN affine, swish, SE always fp32):
//   expand 1x1 + BN + swish           conv engine (conv_gemm.hip; fp16-operand DMA tiles in the fp16 mode)
//   depthwise k x k + BN + swish      dw_same_kernel: input tile staged once in LDS, OXT outputs per thread along x, and the
//                                     squeeze (global average pool) as per-block partial sums -- reduced in a fixed order, no
//                                     atomics, so the result is deterministic
//   squeeze -> FC -> swish -> FC -> sigmoid   se_gate_kernel, one block per image (wave-shuffle dot products)
//   project 1x1 + BN (+ identity)     gated_project_kernel: the SE gate multiplies the A operand on its way from HBM to LDS
//                                     (sigmoid(s) * x, then the conv -- the reference's order), MFMA, BN + identity epilogue
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <map>
#include <string>
#include <type_traits>
#include <vector>

#include "adaf_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

// ---- 16-byte chunks of activations: 4 floats or 8 halfs ---------------------------------------------------------------
template <typename T> struct Chunk;
template <> struct Chunk<float> {
    static constexpr int V = 4;
    static __device__ __forceinline__ float one(float v) { return v; }
    static __device__ __forceinline__ void unpack(const u32x4 u, float (&f)[4]) {
        f[0] = __uint_as_float(u.x); f[1] = __uint_as_float(u.y); f[2] = __uint_as_float(u.z); f[3] = __uint_as_float(u.w);
    }
    static __device__ __forceinline__ u32x4 pack(const float (&f)[4]) {
        return u32x4{__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3])};
    }
};
template <> struct Chunk<_Float16> {
    static constexpr int V = 8;
    static __device__ __forceinline__ _Float16 one(float v) { return adaf_f16_of(v); }
    static __device__ __forceinline__ void unpack(const u32x4 u, float (&f)[8]) {
        const f16x8 h = __builtin_bit_cast(f16x8, u);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = (float)h[e];
    }
    static __device__ __forceinline__ u32x4 pack(const float (&f)[8]) {
        f16x8 h;
#pragma unroll
        for (int e = 0; e < 8; ++e) h[e] = adaf_f16_of(f[e]);
        return __builtin_bit_cast(u32x4, h);
    }
};

// logistic function on the exp2 / reciprocal units (1 ulp each; see conv_gemm.hip finish_act)
__device__ __forceinline__ float fast_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f)); }

__device__ __forceinline__ float act_apply(float v, int act) {
    // (one logistic evaluation whichever of the two activations wants it: written as two separate branches the compiler
    // if-converted them and every element paid for both exp2 / rcp pairs)
#pragma clang fp contract(off)       // (the swish product is a rounded value of its own: see act_of)
    if (act == ADAF_ACT_SWISH || act == ADAF_ACT_SIGMOID) {
        const float g = fast_sigmoid(v);
        return act == ADAF_ACT_SWISH ? v * g : g;
    }
    if (act == ADAF_ACT_RELU) return fmaxf(v, 0.f);
    if (act == ADAF_ACT_RELU6) return fminf(fmaxf(v, 0.f), 6.f);
    return v;
}

// The network's own activation (swish) as a compile-time constant.  With the runtime `act` every ELEMENT of an epilogue went through
// two scalar compare-and-branch pairs (the compiler does not unswitch the unrolled epilogue loops: 65 of each per 32 outputs in the
// depthwise kernel, with the relu / relu6 min-max pairs evaluated beside the logistic): act_switch runs the epilogue body once with
// ACT = swish, or once with ACT = -1 (any other activation, resolved per element as before).
template <int ACT>
__device__ __forceinline__ float act_of(float v, int act) {
    // (the product is a ROUNDED value -- it is what gets stored -- so it must not be contracted into the squeeze sum that follows it in
    //  some epilogues and not in others: the pooled sums of every kernel form have to agree to the bit)
#pragma clang fp contract(off)
    if constexpr (ACT == ADAF_ACT_SWISH) return v * fast_sigmoid(v);
    else return act_apply(v, act);
}
template <typename F>
__device__ __forceinline__ void act_switch(int act, F&& f) {
    if (act == ADAF_ACT_SWISH) f(std::integral_constant<int, ADAF_ACT_SWISH>{});
    else f(std::integral_constant<int, -1>{});
}

// a / d for small per-thread index splits (0 <= a < 2^20, d >= 1): trunc((a + 1/2) * (1 / d)) in fp32 -- the real (a + 1/2) / d is at least
// 1 / (2 d) from an integer and the fp32 result within (a / d) 2^-22 of it, so the truncation is exact.  Three VALU instructions
// instead of the ~22 of the emulated u32 division.
struct FastDiv {
    float inv, half_inv;
    int d;
    __device__ __forceinline__ explicit FastDiv(int d_) : inv(__builtin_amdgcn_rcpf((float)d_)), half_inv(0.5f * inv), d(d_) {}
    __device__ __forceinline__ int div(int a) const { return (int)fmaf((float)a, inv, half_inv); }
};

// a * b + c on the full-rate 24-bit multiplier (v_mul_lo_u32 / v_mad_u64_u32 are quarter rate; hipcc turns __mul24 of a value it
// cannot bound back into them).  b is wave-uniform (an SGPR operand); |a|, |b| < 2^23.
__device__ __forceinline__ int mad24s(int a, int b, int c) {
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}

// ---- depthwise k x k, SAME padding, + BN affine + activation + squeeze partial sums -----------------------------------
// A block owns IMB images x one channel slice (CS = LPP 16-byte chunks) x TH output rows.  Its input rows (with the padding
// columns, zero-filled) are staged once in LDS; then thread (image, channel chunk cg, pixel-group lane pg) walks the
// groups of OXT horizontally adjacent outputs pg, pg + PG, ... of its image: every staged value is read and converted once
// per filter row, the K taps of the row sit in registers.  The squeeze (global average pool of the ACTIVATED output) leaves
// as per-block partial sums reduced in a fixed order -- thread partials through LDS, one thread per (image, channel) -- so
// it is deterministic and independent of what else is in the batch.  All index arithmetic inside the loops is incremental
// (runtime divisions were a third of the first version's instructions).
struct DwArgs {
    const void* x;       // [n][H][W][C]
    void* out;           // [n][OH][OW][C]
    const float* wt;     // [K*K][C] taps
    const float* scale;  // [C] folded BN
    const float* bias;
    float* pool_part;    // [n][tiles][C] sums of the activated outputs over the block's pixels, or nullptr
    int n, H, W, C, OH, OW;
    int pad_t, pad_l;    // padding BEFORE the first row / column (SAME padding is asymmetric: the rest falls off the far edge)
    int act;
    int TH, tiles;       // output rows per block; tiles per image (tiles_y * tiles_x)
    int TWG, tiles_x;    // x groups (of OXT outputs) per block, blocks per image along x
    int LPP, CS, slices; // lanes (16-byte chunks) per pixel in a block's channel slice, channels per slice, slices per pixel
    int pitch16;         // LDS pixel pitch in 16-byte units (>= LPP; chosen so neighbouring thread groups hit different banks)
    int WP;              // staged columns: (TWG * OXT - 1) * S + K
    int img_lds;         // bytes of one image's staged tile: ((TH - 1) * S + K) * WP * pitch16 * 16 (a kernel argument: a product of
                         // three runtime values in the index arithmetic of every pixel group is two quarter-rate multiplies)
    int IMB, PG;         // images per block, pixel-group lanes per (image, channel chunk): IMB * PG * LPP <= 256
    int igroups;         // ceil(n / IMB)
    int total;           // work items (= blocks): igroups * tiles * slices
    // XN > 0 (the expand conv computed into the staged tile, see the kernel): x is the block INPUT [n][H][W][cin] and
    const void* xw;      // the expand filter [C][cin] (fp16)
    const float* xscale; // the expand BN [C]
    const float* xbias;
    int cin;
#ifdef EF_TRACE
    unsigned long long* trace;   // [blocks][4 waves][8] s_memtime stamps (tools/dw_trace.py; never compiled into the shipped library)
#endif
};

// Trace build (tools/exp/build_mbw_trace.sh with EF_TRACE): lane 0 of every wave stamps s_memtime at the phase boundaries.
#ifdef EF_TRACE
#define EF_STAMP(slot_) do { if (a.trace && (threadIdx.x & 63) == 0) a.trace[((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8 + (slot_)] = __builtin_readcyclecounter(); } while (0)
#else
#define EF_STAMP(slot_) do { } while (0)
#endif
#ifdef EF_TRACE
static unsigned long long* ef_trace_buf = nullptr;
static int ef_trace_c = 0, ef_trace_k = 0, ef_trace_s = 0;
#endif

//
// XN > 0 (fp16 storage): the EXPAND conv of the MBConv block runs inside the staging step -- the 6x-expanded map never exists in HBM (it
// is 2/3 of what the four-launch plan of blocks 2-8 moves).  a.x is the block's narrow input [n][H][W][cin]; the tile's window pixels
// go through v_mfma_f32_16x16x32_f16 in groups of 16 with the slice's filter rows as the A operand (XN = CS / 16 row tiles held in
// registers) and the pixels as B, so a lane ends up with FOUR CONSECUTIVE CHANNELS of one pixel per row tile: BN + swish on the four
// accumulators at full lanes, one ds_write_b64 into the very layout the loaded path stages -- the taps below do not know the difference.
// Window pixels in the SAME padding are written as zeros (not swish(bias)).  The halo of a tile is expanded again by its neighbours
// (stride 2: 3-12 %; the 18 x 18 maps of blocks 6-7 fit whole: none), which is why the round-3 form of this idea lost on stride-1 tiles.
template <int K, int S, int OXT, typename T, int XN = 0, int KSX = 1>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(K == 5 ? 3 : 4, 8))) void dw_same_kernel(const DwArgs a) {
    constexpr int V = Chunk<T>::V;
    constexpr int NC = (OXT - 1) * S + K;    // input columns the OXT outputs of a thread touch
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    const int tid = threadIdx.x;
    int bid = blockIdx.x;
    EF_STAMP(0);
    {
        // XCD-aware, bijective remap (the hardware places block b on XCD b % 8): every XCD gets a contiguous range of work
        // items, so the channel slices of one tile -- which read interleaved 16 LPP-byte pieces of the same pixels -- and
        // neighbouring tiles -- which share halo rows -- meet in ONE L2.  Measured before: the stride-2 layer of block 2
        // fetched 2.2x its input from the fabric.
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int slice = bid % a.slices; bid /= a.slices;
    const int tile = bid % a.tiles;
    const int img0 = (bid / a.tiles) * a.IMB;
    const int nimg = min(a.IMB, a.n - img0);
    const int c0 = slice * a.CS;
    const int ty = tile / a.tiles_x, tx = tile - ty * a.tiles_x;
    const int oy0 = ty * a.TH;
    const int th = min(a.TH, a.OH - oy0);
    const int xg0 = tx * a.TWG;              // first x group of this tile
    const int ihn = (th - 1) * S + K;        // staged input rows
    const int iy0 = oy0 * S - a.pad_t, ix0 = xg0 * OXT * S - a.pad_l;
    const int pitchB = a.pitch16 * 16;
    const int img_lds = a.img_lds;                                           // bytes of one image's staged tile (ihmax * WP * pitchB)
    char* xin = dsm;                                                         // [IMB][ihmax][WP][pitch16 * 16 B]
    float* wl = reinterpret_cast<float*>(dsm + (size_t)a.IMB * img_lds);     // [K*K][CS] taps
    float* sbl = wl + K * K * a.CS;                                          // [2][CS] BN scale, bias
    // Prologue: EVERY global load that does not depend on another one is requested before the first is used (taps, BN, and below the
    // expand filter rows and the first batch of window pixels), and its integer arithmetic is kept off the slow paths: the launch is
    // VALU-issue bound, the prologue was 250 VALU instructions of a wave's ~1200, and a fifth of those were 32 / 64-bit integer
    // multiplies (quarter rate) of address and index computations (tools/dw_trace.py).  Taps: wave w fetches the filter rows w, w + 4,
    // ... (scalar row base + lane: no per-thread index arithmetic), BN: wave 0 the scales, wave 1 the biases.
    const int lane_ = tid & 63, wave_ = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NTR = (K * K + 3) / 4;
    const int lcs = lane_ < a.CS ? lane_ : 0;
    float tw[NTR];
#pragma unroll
    for (int u = 0; u < NTR; ++u) {
        const int t = wave_ + 4 * u;
        tw[u] = (a.wt + (t < K * K ? t : 0) * a.C + c0)[lcs];
    }
    const float sbv = ((wave_ == 0 ? a.scale : a.bias) + c0)[lcs];
    auto store_taps = [&]() {
        if (lane_ < a.CS) {
#pragma unroll
            for (int u = 0; u < NTR; ++u)
                if (wave_ + 4 * u < K * K) (wl + (wave_ + 4 * u) * a.CS)[lane_] = tw[u];
            if (wave_ < 2) (sbl + wave_ * a.CS)[lane_] = sbv;
        }
    };
    if constexpr (XN == 0) store_taps();
    if constexpr (XN > 0) {
        static_assert(sizeof(T) == 2, "the fused expand is an fp16-storage path");
        const int lane = tid & 63, wave = tid >> 6;
        const int px = lane & 15, kq = lane >> 4;
        const int cin = a.cin;                                             // cin % 8 == 0, cin <= 32 KSX
        const _Float16* wx = static_cast<const _Float16*>(a.xw) + (size_t)c0 * cin;
        // the part of the window that lies inside the image: only those pixels are expanded; the SAME padding around them is zeros
        const int vr0 = max(0, -iy0), vr1 = min(ihn, a.H - iy0), vc0 = max(0, -ix0), vc1 = min(a.WP, a.W - ix0);
        const int vw = vc1 - vc0, nv = (vr1 - vr0) * vw;
        u32x4 af[XN][KSX];
        f32x4 xs[XN], xb[XN];
        const int pxcin = px * cin;
#pragma unroll
        for (int rt = 0; rt < XN; ++rt) {
#pragma unroll
            for (int ks = 0; ks < KSX; ++ks) {
                const int k0 = 32 * ks + 8 * kq;
                af[rt][ks] = *reinterpret_cast<const u32x4*>(wx + (pxcin + 16 * rt * cin + (k0 < cin ? k0 : 0)));
            }
            xs[rt] = *reinterpret_cast<const f32x4*>(a.xscale + c0 + 16 * rt + 4 * kq);
            xb[rt] = *reinterpret_cast<const f32x4*>(a.xbias + c0 + 16 * rt + 4 * kq);
        }
        const int ntot = nimg * nv, ngr = (ntot + 15) >> 4;
        const FastDiv dnv(nv > 0 ? nv : 1), dvw(vw > 0 ? vw : 1);
        const _Float16* xg = static_cast<const _Float16*>(a.x) + (size_t)img0 * a.H * a.W * cin;      // (the block's first image: scalar)
        // the wave's pixel groups g = wave, wave + 4, ... in batches of GB, two batches deep: the B fragments of batch i + 1 are in flight
        // (unconditional loads, 16 bytes per lane) while batch i goes through the MFMAs and the swish
        constexpr int GB = 4 / KSX;
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        auto xload = [&](int g0, u32x4 (&bf)[GB][KSX], int (&dst)[GB]) {
#pragma unroll
            for (int u = 0; u < GB; ++u) {
                const int p = 16 * (g0 + 4 * u) + px;
                const bool valid = p < ntot;
                const int pc = valid ? p : 0;
                // (24-bit multiplies: full rate, where v_mul_lo_u32 / v_mad_u64_u32 are quarter rate -- every operand here is far below 2^23)
                const int im = dnv.div(pc), q = pc - __mul24(im, nv);
                const int vr = dvw.div(q), vc = q - __mul24(vr, vw);
                const int row = vr0 + vr, col = vc0 + vc;
                dst[u] = valid ? __mul24(im, img_lds) + __mul24(__mul24(row, a.WP) + col, pitchB) + 8 * kq : -1;
                // (a 32-bit byte offset from the block's first image: scalar base + vector offset, no 64-bit arithmetic per pixel)
                const unsigned src = 2u * (unsigned)mad24s(mad24s(mad24s(im, a.H, iy0 + row), a.W, ix0 + col), cin, 0);
#pragma unroll
                for (int ks = 0; ks < KSX; ++ks) {
                    const int k0 = 32 * ks + 8 * kq;
                    bf[u][ks] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(xg) + (src + 2u * (unsigned)(k0 < cin ? k0 : 0)));
                }
            }
        };
        auto xcompute = [&](int g0, u32x4 (&bf)[GB][KSX], const int (&dst)[GB]) {
#pragma unroll
            for (int ks = 0; ks < KSX; ++ks)
                if (32 * ks + 8 * kq >= cin) {
#pragma unroll
                    for (int u = 0; u < GB; ++u) bf[u][ks] = u32x4{0u, 0u, 0u, 0u};
                }
#pragma unroll
            for (int u = 0; u < GB; ++u) {
                if (g0 + 4 * u >= ngr) break;                  // (wave-uniform)
#pragma unroll
                for (int rt = 0; rt < XN; ++rt) {
                    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < KSX; ++ks)
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, af[rt][ks]), __builtin_bit_cast(f16x8, bf[u][ks]), acc, 0, 0, 0);
                    h4 o;
                    o.x = adaf_f16_of(act_of<ADAF_ACT_SWISH>(fmaf(acc.x, xs[rt].x, xb[rt].x), ADAF_ACT_SWISH));
                    o.y = adaf_f16_of(act_of<ADAF_ACT_SWISH>(fmaf(acc.y, xs[rt].y, xb[rt].y), ADAF_ACT_SWISH));
                    o.z = adaf_f16_of(act_of<ADAF_ACT_SWISH>(fmaf(acc.z, xs[rt].z, xb[rt].z), ADAF_ACT_SWISH));
                    o.w = adaf_f16_of(act_of<ADAF_ACT_SWISH>(fmaf(acc.w, xs[rt].w, xb[rt].w), ADAF_ACT_SWISH));
                    if (dst[u] >= 0) *reinterpret_cast<h4*>(xin + dst[u] + 32 * rt) = o;
                }
            }
        };
        u32x4 bfA[GB][KSX], bfB[GB][KSX];
        int dstA[GB], dstB[GB];
        constexpr int GSTEP = 4 * GB;
        if (wave < ngr) xload(wave, bfA, dstA);
        // (everything above is in flight; now the LDS side of the prologue)
        if (nv < ihn * a.WP) {                                             // (block-uniform: tiles on the image border)
            const int chunks = nimg * (img_lds >> 4);
            for (int i = tid; i < chunks; i += 256) *reinterpret_cast<u32x4*>(xin + 16 * i) = u32x4{0u, 0u, 0u, 0u};
        }
        store_taps();
        if (nv < ihn * a.WP) __syncthreads();
#pragma unroll
        for (int rt = 0; rt < XN; ++rt)
#pragma unroll
            for (int ks = 0; ks < KSX; ++ks)
                if (32 * ks + 8 * kq >= cin) af[rt][ks] = u32x4{0u, 0u, 0u, 0u};
        EF_STAMP(1);
        for (int g0 = wave; g0 < ngr; g0 += 2 * GSTEP) {
            if (g0 + GSTEP < ngr) xload(g0 + GSTEP, bfB, dstB);
            xcompute(g0, bfA, dstA);
            if (g0 + GSTEP < ngr) {
                if (g0 + 2 * GSTEP < ngr) xload(g0 + 2 * GSTEP, bfA, dstA);
                xcompute(g0 + GSTEP, bfB, dstB);
            }
        }
    } else {
        // staging: lane (pixel slot, chunk) walks the tile's pixels slot, slot + PP, ...
        const FastDiv dlp(a.LPP), dwp(a.WP);
        const int PP = dlp.div(256);
        const int slot = dlp.div(tid), cgl = tid - __mul24(slot, a.LPP);
        if (slot < PP) {
            const int row0 = dwp.div(slot), col0 = slot - __mul24(row0, a.WP);
            const int drow = dwp.div(PP), dcol = PP - drow * a.WP;
            // all offsets advance by precomputed steps (no multiplies in the loop): pixel slot -> slot + PP
            const int lstep = (drow * a.WP + dcol) * pitchB;                  // LDS bytes per step
            const int gstep = (drow * a.W + dcol) * a.C;                      // source elements per step
            const int gwrap = (a.W - a.WP) * a.C;                             // extra source elements when the column wraps into the next row
            for (int im = 0; im < nimg; ++im) {
                const T* xb = static_cast<const T*>(a.x) + (size_t)(img0 + im) * a.H * a.W * a.C + c0 + cgl * V;
                char* dst = xin + (size_t)im * img_lds + cgl * 16;
                int col = col0, row = row0;
                int loff = (row0 * a.WP + col0) * pitchB;
                long long goff = ((long long)(iy0 + row0) * a.W + (ix0 + col0)) * a.C;
                // batches of four: the loads of a batch are all in flight before the first LDS store waits on one
                while (row < ihn) {
                    u32x4 v[4];
                    int off[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int iy = iy0 + row, ix = ix0 + col;
                        off[u] = row < ihn ? loff : -1;
                        // (an unconditional load from a clamped address: a load inside a per-lane branch gets its own s_waitcnt)
                        const bool inb = row < ihn && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
                        v[u] = *reinterpret_cast<const u32x4*>(xb + (inb ? goff : 0));
                        if (!inb) v[u] = u32x4{0u, 0u, 0u, 0u};
                        col += dcol; row += drow; loff += lstep; goff += gstep;
                        if (col >= a.WP) { col -= a.WP; ++row; goff += gwrap; }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        if (off[u] >= 0) *reinterpret_cast<u32x4*>(dst + off[u]) = v[u];
                }
            }
        }
    }
    EF_STAMP(2);
    __syncthreads();
    EF_STAMP(3);
    const int TPI = a.LPP * a.PG;            // threads per image
    const FastDiv dlpp(a.LPP);
    const int im = FastDiv(TPI).div(tid), rem = tid - __mul24(im, TPI);
    const int pg = dlpp.div(rem), cg = rem - __mul24(pg, a.LPP);
    float psum[V];
#pragma unroll
    for (int e = 0; e < V; ++e) psum[e] = 0.f;
    if (im < nimg) {
        const int nxg = min(a.TWG, (a.OW + OXT - 1) / OXT - xg0);     // x groups in this tile
        const FastDiv dnxg(nxg);
        const int dr = dnxg.div(a.PG), dxg = a.PG - dr * nxg;
        int r = dnxg.div(pg), xg = pg - __mul24(r, nxg);
        T* ob = static_cast<T*>(a.out) + ((size_t)(img0 + im) * a.OH + oy0) * a.OW * a.C + c0 + cg * V;
        const char* xim = xin + __mul24(im, img_lds) + cg * 16;
        const int rowstride = a.WP * pitchB;                  // LDS bytes per staged row
        while (r < th) {
            const int ox0 = (xg0 + xg) * OXT;
            const char* rowp = xim + __mul24(__mul24(r * S, a.WP) + xg * (OXT * S), pitchB);
            T* op = ob + __mul24(__mul24(r, a.OW) + ox0, a.C);
            float acc[OXT][V];
#pragma unroll
            for (int o = 0; o < OXT; ++o)
#pragma unroll
                for (int e = 0; e < V; ++e) acc[o][e] = 0.f;
#pragma unroll 1
            for (int ky = 0; ky < K; ++ky) {
                // the K taps of this filter row, then every staged column once: column ci feeds output o through tap
                // kx = ci - o * S (resolved at compile time), so each LDS value is read and converted exactly once
                float w[K][V];
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const float* wp = wl + (ky * K + kx) * a.CS + cg * V;
#pragma unroll
                    for (int e = 0; e < V; e += 4) {
                        const f32x4 w4 = *reinterpret_cast<const f32x4*>(wp + e);
                        w[kx][e] = w4.x; w[kx][e + 1] = w4.y; w[kx][e + 2] = w4.z; w[kx][e + 3] = w4.w;
                    }
                }
#pragma unroll
                for (int ci = 0; ci < NC; ++ci) {
                    float xf[V];
                    Chunk<T>::unpack(*reinterpret_cast<const u32x4*>(rowp + ci * pitchB), xf);
#pragma unroll
                    for (int o = 0; o < OXT; ++o) {
                        const int kx = ci - o * S;
                        if (kx >= 0 && kx < K) {
#pragma unroll
                            for (int e = 0; e < V; ++e) acc[o][e] = fmaf(xf[e], w[kx][e], acc[o][e]);
                        }
                    }
                }
                rowp += rowstride;
            }
            float sc[V], bi[V];
#pragma unroll
            for (int e = 0; e < V; e += 4) {
                const f32x4 s4 = *reinterpret_cast<const f32x4*>(sbl + cg * V + e), b4 = *reinterpret_cast<const f32x4*>(sbl + a.CS + cg * V + e);
                sc[e] = s4.x; sc[e + 1] = s4.y; sc[e + 2] = s4.z; sc[e + 3] = s4.w;
                bi[e] = b4.x; bi[e + 1] = b4.y; bi[e + 2] = b4.z; bi[e + 3] = b4.w;
            }
            act_switch(a.act, [&](auto AC) {
#pragma unroll
                for (int o = 0; o < OXT; ++o) {
                    if (ox0 + o < a.OW) {
                        float v[V];
#pragma unroll
                        for (int e = 0; e < V; ++e) {
                            v[e] = act_of<decltype(AC)::value>(fmaf(acc[o][e], sc[e], bi[e]), a.act);
                            psum[e] += v[e];
                        }
                        *reinterpret_cast<u32x4*>(op + o * a.C) = Chunk<T>::pack(v);
                    }
                }
            });
            xg += dxg; r += dr;
            if (xg >= nxg) { xg -= nxg; ++r; }
        }
    }
    EF_STAMP(4);
    if (a.pool_part) {
        // squeeze: sum over the block's pixels per (image, channel), in a fixed order
        __syncthreads();
        float* red = reinterpret_cast<float*>(dsm);     // [256][V] (the input tiles are dead)
#pragma unroll
        for (int e = 0; e < V; ++e) red[tid * V + e] = psum[e];
        __syncthreads();
        for (int t = tid; t < nimg * a.CS; t += 256) {          // (IMB * CS can exceed the block: 8 images x 48 channels)
            const int ri = t / a.CS, c = t - ri * a.CS;
            const int g = c / V, e = c % V;
            float s = 0.f;
            for (int q = 0; q < a.PG; ++q) s += red[(ri * TPI + q * a.LPP + g) * V + e];
            a.pool_part[((size_t)(img0 + ri) * a.tiles + tile) * a.C + c0 + c] = s;
        }
    }
    EF_STAMP(5);
}

// ---- depthwise k x k on TINY maps (H = W = HW <= 5: the last stages of the network; 5 x 5 at B3's 144^2 patches) -----------------
// On a 5 x 5 map a 5 x 5 window mostly sees padding (14.4 of its 25 taps are inside on average, 6.8 of 9 for 3 x 3), and
// dw_same_kernel multiplies all of them (zeros staged in LDS), reads every staged value through LDS once per filter row, converts it
// on every read and reduces the squeeze sums across threads.  Here a thread owns ONE image x 4 channels: it loads the whole map
// (25 pixels x 4 channels, straight from global memory, every load in flight at once), converts it once, keeps all HW^2 x 4
// accumulators in registers and visits exactly the (output, tap) pairs that fall inside the map -- the loops are unrolled at
// compile time, so "inside" costs nothing -- in the same (ky, kx) order per output as everywhere else (a skipped tap would have
// added an exact zero): the OUTPUT is bit-identical to dw_same_kernel's.  The squeeze sum of an (image, channel) is thread-local -- no LDS,
// no barrier, no partial tiles -- and adds the pixels in raster order, not in dw_same_kernel's (thread partials, then threads): the two
// sums differ in the last bit for about half the channels of a noise input (round 4, tests/test_effnet.py).  b19-b23 (5 x 5 window, 1392 channels): 110 -> ~45 us per 1024 patches.
struct DsArgs {
    const void* x;
    void* out;
    const float* wt;     // [K*K][C]
    const float* scale;
    const float* bias;
    float* pool_part;    // [n][C] sums (tiles = 1) or nullptr
    int n, C, act;
};

template <typename T> struct Quad;      // four consecutive channels of one pixel
template <> struct Quad<float> {
    static __device__ __forceinline__ f32x4 ld(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void st(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct Quad<_Float16> {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ f32x4 ld(const _Float16* p) {
        const h4 v = *reinterpret_cast<const h4*>(p);
        return f32x4{(float)v.x, (float)v.y, (float)v.z, (float)v.w};
    }
    static __device__ __forceinline__ void st(_Float16* p, f32x4 v) { *reinterpret_cast<h4*>(p) = h4{adaf_f16_of(v.x), adaf_f16_of(v.y), adaf_f16_of(v.z), adaf_f16_of(v.w)}; }
};

template <int K, int HW, typename T>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 4))) void dw_small_kernel(const DsArgs a) {
    constexpr int P = (K - 1) / 2;           // SAME padding at stride 1: symmetric
    const int c4 = a.C >> 2;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)a.n * c4) return;
    const int cq = (int)(idx % c4), img = (int)(idx / c4);
    const T* xb = static_cast<const T*>(a.x) + (size_t)img * HW * HW * a.C + 4 * cq;
    f32x4 v[HW * HW];
#pragma unroll
    for (int p = 0; p < HW * HW; ++p) v[p] = Quad<T>::ld(xb + (size_t)p * a.C);
    const f32x4 sc = *reinterpret_cast<const f32x4*>(a.scale + 4 * cq), bi = *reinterpret_cast<const f32x4*>(a.bias + 4 * cq);
    T* ob = static_cast<T*>(a.out) + (size_t)img * HW * HW * a.C + 4 * cq;
    f32x4 psum = {0.f, 0.f, 0.f, 0.f};
    // the output rows in two passes when the map is 5 x 5 (or 4 x 4 under a 5 x 5 window): HW^2 accumulators next to the HW^2
    // converted inputs are 200 + registers = one wave per SIMD; with 3 + 2 rows the kernel fits two or three
    constexpr int RA = (HW == 5 || (HW == 4 && K == 5)) ? (HW + 1) / 2 : HW;
#pragma unroll
    for (int r0 = 0; r0 < HW; r0 += RA) {
        f32x4 acc[RA * HW];
#pragma unroll
        for (int p = 0; p < RA * HW; ++p) acc[p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ky = 0; ky < K; ++ky)
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const f32x4 w = *reinterpret_cast<const f32x4*>(a.wt + (size_t)(ky * K + kx) * a.C + 4 * cq);
#pragma unroll
                for (int oy = r0; oy < r0 + RA; ++oy)
#pragma unroll
                    for (int ox = 0; ox < HW; ++ox) {
                        const int iy = oy + ky - P, ix = ox + kx - P;
                        if (oy >= HW || iy < 0 || iy >= HW || ix < 0 || ix >= HW) continue;       // resolved at compile time
                        f32x4& s = acc[(oy - r0) * HW + ox];
                        const f32x4 xin = v[iy * HW + ix];
                        s.x = fmaf(xin.x, w.x, s.x); s.y = fmaf(xin.y, w.y, s.y); s.z = fmaf(xin.z, w.z, s.z); s.w = fmaf(xin.w, w.w, s.w);
                    }
            }
        act_switch(a.act, [&](auto AC) {
            constexpr int ACT = decltype(AC)::value;
#pragma unroll
            for (int q = 0; q < RA * HW; ++q) {
                const int p = r0 * HW + q;
                if (p >= HW * HW) continue;
                f32x4 o;
                o.x = act_of<ACT>(fmaf(acc[q].x, sc.x, bi.x), a.act);
                o.y = act_of<ACT>(fmaf(acc[q].y, sc.y, bi.y), a.act);
                o.z = act_of<ACT>(fmaf(acc[q].z, sc.z, bi.z), a.act);
                o.w = act_of<ACT>(fmaf(acc[q].w, sc.w, bi.w), a.act);
                psum += o;
                Quad<T>::st(ob + (size_t)p * a.C, o);
            }
        });
    }
    if (a.pool_part) *reinterpret_cast<f32x4*>(a.pool_part + (size_t)img * a.C + 4 * cq) = psum;
}

// ---- expand 1x1 (+BN+activation) for narrow inputs (cin <= 64: blocks 2..8 of B3) ------------------------------------------
// A GEMM with K = 24..48 and N = 6 K is all epilogue: on the conv engine a 128 x 32 tile issues one DMA slice, waits out its
// latency, runs two MFMAs per wave and then spends its life in the epilogue -- 0.9 ms for block 2's expand (1.5 GB of fp16 output,
// 2 TB/s), with no wasted traffic (WRITE_SIZE = the algorithmic bytes).  Here the filter (all N rows, k padded to a whole MFMA
// step) and the BN affine live in LDS for the lifetime of a PERSISTENT block, every wave walks its own 32-row strips with no block
// barrier after the staging, the A fragments of strip i+1 are in flight (plain 16-byte global loads into registers: the fragment
// of a lane is contiguous in memory) while strip i is multiplied, and a strip's outputs leave through a wave-private slab as
// 16-byte pieces of 64 consecutive channels.  Same MFMA instruction and k order as the engine's launch: bit-identical.
struct ExpArgs {
    const void* x;        // [M][K]
    const void* w;        // [N][K] (storage type)
    const float* scale;   // [N]
    const float* bias;
    void* out;            // [M][N]
    int M, K, N, act;
    int KP;               // K padded to a whole MFMA k step (elements)
    int NP;               // N padded to 32
    int strips;           // ceil(M / 32)
};

// WHOLE: a row of the output is not a whole number of 128-byte lines (N = 144 or 288 halfs): 128-byte pieces would straddle lines and
// HBM sees twice as many partial-line writes (measured: block 2 at 2.5 TB/s of stores against 4.4 for the aligned N = 192).  The
// wave then keeps its whole strip (32 rows x N: one contiguous, line-aligned range of the output) in its slab and copies it out
// linearly, 1 KB per store instruction.
template <typename T, int KS, bool WHOLE>
__global__ __launch_bounds__(1024) void ef_expand_kernel(const ExpArgs a) {
    constexpr int V = Chunk<T>::V;
    constexpr int KSTEP = 2 * V;                          // k elements per MFMA step group (16 bytes per lane, two lane halves)
    const int SROW = (WHOLE ? a.NP : 64) * (int)sizeof(T) + 16;   // slab row: 64 channels (WHOLE: all of them) + a 16-byte skew
    constexpr int CPR = 64 / V;                           // 16-byte pieces per 64-channel slab row
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = blockDim.x >> 6;
    const int wpitch = a.KP * (int)sizeof(T) + 16;
    char* wl = dsm;                                                          // [NP][wpitch]
    float* sb = reinterpret_cast<float*>(wl + (size_t)a.NP * wpitch);      // [2][NP]
    char* slab = reinterpret_cast<char*>(sb + 2 * a.NP) + (size_t)wave * 32 * SROW;
    {
        const int CPP = wpitch / 16 - 1, kchunks = a.K / V;
        const T* wb = static_cast<const T*>(a.w);
        for (int i = tid; i < a.NP * CPP; i += blockDim.x) {
            const int h = i / CPP, c = i - h * CPP;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (h < a.N && c < kchunks) v = *reinterpret_cast<const u32x4*>(wb + (size_t)h * a.K + c * V);
            *reinterpret_cast<u32x4*>(wl + (size_t)h * wpitch + c * 16) = v;
        }
        for (int i = tid; i < 2 * a.NP; i += blockDim.x) {
            const int which = i / a.NP, h = i - which * a.NP;
            sb[i] = h < a.N ? (which ? a.bias[h] : a.scale[h]) : 0.f;
        }
    }
    __syncthreads();
    const int nl = lane & 31, half = lane >> 5;
    const T* xb = static_cast<const T*>(a.x);
    const int stride = gridDim.x * nwaves;
    int strip = blockIdx.x * nwaves + wave;
    auto load = [&](int st, u32x4 (&f)[KS]) {         // (unconditional loads from clamped addresses, zeroed afterwards)
        const long long row = (long long)st * 32 + nl;
        const bool ok = st < a.strips && row < a.M;
        const T* src = xb + (size_t)(ok ? row : 0) * a.K;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const int k = kk * KSTEP + half * V;
            f[kk] = *reinterpret_cast<const u32x4*>(src + (k < a.K ? k : 0));
        }
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
            if (!(ok && kk * KSTEP + half * V < a.K)) f[kk] = u32x4{0u, 0u, 0u, 0u};
    };
    u32x4 af[KS], an[KS];
    load(strip, af);
    const int npairs = (a.NP + 63) >> 6;
    const int c = lane % CPR, rsub = lane / CPR;
    for (; strip < a.strips; strip += stride) {
        load(strip + stride, an);
        const long long m0 = (long long)strip * 32;
        for (int cp = 0; cp < npairs; ++cp) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n0 = cp * 64 + j * 32;
                if (n0 < a.NP) {
                    f32x16 acc;
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
                    const char* brow = wl + (size_t)(n0 + nl) * wpitch + half * 16;
#pragma unroll
                    for (int kk = 0; kk < KS; ++kk) {
                        const f32x4 bf = *reinterpret_cast<const f32x4*>(brow + kk * 32);
                        if constexpr (sizeof(T) == 2) {
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[kk]), __builtin_bit_cast(f16x8, bf), acc, 0, 0, 0);
                        } else {
                            const f32x4 a4 = __builtin_bit_cast(f32x4, af[kk]);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, bf.x, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, bf.y, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, bf.z, acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, bf.w, acc, 0, 0, 0);
                        }
                    }
                    const float sc = sb[n0 + nl], bi = sb[a.NP + n0 + nl];
                    char* srow = slab + (size_t)((WHOLE ? n0 : j * 32) + nl) * sizeof(T);
                    act_switch(a.act, [&](auto AC) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const int p = (i & 3) + 8 * (i >> 2) + 4 * half;
                            *reinterpret_cast<T*>(srow + p * SROW) = Chunk<T>::one(act_of<decltype(AC)::value>(fmaf(acc[i], sc, bi), a.act));
                        }
                    });
                }
            }
            if (!WHOLE) {
                __builtin_amdgcn_wave_barrier();
                const int col = cp * 64 + c * V;
                if (col < a.N) {
                    T* ob = static_cast<T*>(a.out) + col;
#pragma unroll
                    for (int r = rsub; r < 32; r += 64 / CPR) {
                        if (m0 + r < a.M)
                            *reinterpret_cast<u32x4*>(ob + (size_t)(m0 + r) * a.N) = *reinterpret_cast<const u32x4*>(slab + r * SROW + c * 16);
                    }
                }
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (WHOLE) {
            // the strip is one contiguous range of the output: piece q = (row q / PPR, chunk q % PPR), lane q = lane, lane + 64, ...
            __builtin_amdgcn_wave_barrier();
            const int PPR = a.N / V;
            const int rows = (int)min((long long)32, a.M - m0);
            const int total = rows * PPR;
            const int dr = 64 / PPR, dc = 64 - dr * PPR;
            int r = lane / PPR, cc = lane - r * PPR;
            T* ob = static_cast<T*>(a.out) + (size_t)m0 * a.N;
            for (int q = lane; q < total; q += 64) {
                *reinterpret_cast<u32x4*>(ob + (size_t)q * V) = *reinterpret_cast<const u32x4*>(slab + r * SROW + cc * 16);
                r += dr; cc += dc;
                if (cc >= PPR) { cc -= PPR; ++r; }
            }
            __builtin_amdgcn_wave_barrier();
        }
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) af[kk] = an[kk];
    }
}

// ---- gated project 1x1 for narrow blocks (K <= 64 hidden channels, N <= 32 outputs: blocks 0-1 of B3, 5.3 M rows each) ----------
// gated_project_kernel walks these as 41 k tiles of 128 rows with a barrier pair per 64-wide k slice -- one slice, i.e. a block
// that loads, synchronises, issues a handful of MFMAs and stores: 250-290 us for 0.7 GB.  The strip form of ef_expand_kernel: the
// filter + BN in LDS for the life of a persistent block, a wave per 32-row strip with no block barrier, the next strip's rows in
// flight while this one is multiplied; the SE gate (per image, per hidden channel) multiplies the A fragments in registers exactly
// as gated_project_kernel's staging does (fp32 product, rounded to the storage type: the operand the MFMA sees is the same), same
// MFMA instruction and k order, same epilogue arithmetic (BN affine, + identity in fp32, activation, one rounding): bit-identical.
struct NprojArgs {
    const void* x;        // [M][K]
    const float* gate;    // [M / HW][K]
    const void* w;        // [N][K] (storage type)
    const float* scale;   // [N]
    const float* bias;
    const void* res;      // [M][N] or nullptr
    void* out;            // [M][N]
    int M, K, N, HW, act;
    int KP, strips;
};

template <typename T, int KS>
__global__ __launch_bounds__(512) void ef_nproj_kernel(const NprojArgs a) {
    constexpr int V = Chunk<T>::V;
    constexpr int KSTEP = 2 * V;
    constexpr int SP = 36;                                // slab row pitch in floats (32 columns + skew)
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = blockDim.x >> 6;
    const int wpitch = a.KP * (int)sizeof(T) + 16;
    char* wl = dsm;                                                   // [32][wpitch]
    float* sb = reinterpret_cast<float*>(wl + 32 * wpitch);           // [2][32]
    float* slab = sb + 64 + wave * 32 * SP;
    {
        const int CPP = wpitch / 16 - 1, kchunks = a.K / V;
        const T* wb = static_cast<const T*>(a.w);
        for (int i = tid; i < 32 * CPP; i += blockDim.x) {
            const int h = i / CPP, c = i - h * CPP;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (h < a.N && c < kchunks) v = *reinterpret_cast<const u32x4*>(wb + (size_t)h * a.K + c * V);
            *reinterpret_cast<u32x4*>(wl + (size_t)h * wpitch + c * 16) = v;
        }
        for (int i = tid; i < 64; i += blockDim.x) {
            const int which = i >> 5, h = i & 31;
            sb[i] = h < a.N ? (which ? (a.bias ? a.bias[h] : 0.f) : (a.scale ? a.scale[h] : 1.f)) : 0.f;
        }
    }
    __syncthreads();
    const int nl = lane & 31, half = lane >> 5;
    const T* xb = static_cast<const T*>(a.x);
    const int stride = gridDim.x * nwaves;
    int strip = blockIdx.x * nwaves + wave;
    auto load = [&](int st, u32x4 (&f)[KS]) {                    // the gated rows of strip st as A fragments
        const long long row = (long long)st * 32 + nl;
        const bool ok = st < a.strips && row < a.M;
        const long long rr = ok ? row : 0;
        const T* src = xb + (size_t)rr * a.K;
        const float* g = a.gate + (size_t)(rr / a.HW) * a.K;
        u32x4 xv[KS];
        f32x4 gq[KS][V / 4];
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {                 // (unconditional loads from clamped addresses, every one in flight before the first use)
            const int k = kk * KSTEP + half * V, kc = k < a.K ? k : 0;
            xv[kk] = *reinterpret_cast<const u32x4*>(src + kc);
#pragma unroll
            for (int e = 0; e < V / 4; ++e) gq[kk][e] = *reinterpret_cast<const f32x4*>(g + kc + 4 * e);
        }
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            float f32[V];
            Chunk<T>::unpack(xv[kk], f32);
#pragma unroll
            for (int e = 0; e < V; e += 4) {
                const f32x4 gv = gq[kk][e / 4];
                f32[e] *= gv.x; f32[e + 1] *= gv.y; f32[e + 2] *= gv.z; f32[e + 3] *= gv.w;
            }
            f[kk] = (ok && kk * KSTEP + half * V < a.K) ? Chunk<T>::pack(f32) : u32x4{0u, 0u, 0u, 0u};
        }
    };
    u32x4 af[KS], an[KS];
    load(strip, af);
    const int PPR = a.N / V;                               // 16-byte pieces per output row
    const T* rs = static_cast<const T*>(a.res);
    T* ob = static_cast<T*>(a.out);
    for (; strip < a.strips; strip += stride) {
        load(strip + stride, an);
        const long long m0 = (long long)strip * 32;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        const char* brow = wl + (size_t)nl * wpitch + half * 16;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            const f32x4 bf = *reinterpret_cast<const f32x4*>(brow + kk * 32);
            if constexpr (sizeof(T) == 2) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[kk]), __builtin_bit_cast(f16x8, bf), acc, 0, 0, 0);
            } else {
                const f32x4 a4 = __builtin_bit_cast(f32x4, af[kk]);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, bf.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, bf.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, bf.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, bf.w, acc, 0, 0, 0);
            }
        }
        const float sc = sb[nl], bi = sb[32 + nl];
#pragma unroll
        for (int i = 0; i < 16; ++i) slab[((i & 3) + 8 * (i >> 2) + 4 * half) * SP + nl] = fmaf(acc[i], sc, bi);
        __builtin_amdgcn_wave_barrier();
        // the strip's outputs are one contiguous range: piece q = (row q / PPR, chunk q % PPR)
        const int rows = (int)min((long long)32, a.M - m0);
        const int total = rows * PPR;
        for (int q = lane; q < total; q += 64) {
            const int r = q / PPR, c = q - r * PPR;
            float v[V];
#pragma unroll
            for (int e = 0; e < V; e += 4) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(slab + r * SP + c * V + e);
                v[e] = t.x; v[e + 1] = t.y; v[e + 2] = t.z; v[e + 3] = t.w;
            }
            const size_t g = (size_t)(m0 + r) * a.N + c * V;
            if (rs) {
                float rr[V];
                Chunk<T>::unpack(*reinterpret_cast<const u32x4*>(rs + g), rr);
#pragma unroll
                for (int e = 0; e < V; ++e) v[e] += rr[e];
            }
            if (a.act != ADAF_ACT_NONE) {
#pragma unroll
                for (int e = 0; e < V; ++e) v[e] = act_apply(v[e], a.act);
            }
            *reinterpret_cast<u32x4*>(ob + g) = Chunk<T>::pack(v);
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) af[kk] = an[kk];
    }
}

// mean[n][c] = sum_t part[n][t][c] / hw  (the stand-alone op's squeeze output)
__global__ void pool_finish_kernel(const float* __restrict__ part, int n, int tiles, int c, float inv_hw, float* __restrict__ mean) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * c) return;
    const int img = idx / c, ch = idx - img * c;
    float s = 0.f;
    for (int t = 0; t < tiles; ++t) s += part[((size_t)img * tiles + t) * c + ch];
    mean[idx] = s * inv_hw;
}

// ---- squeeze-and-excite gate: gate[n][c] = sigmoid(W_e swish(W_r mean[n] + b_r) + b_e) ----------------------------------
// model.py MBConvBlock.forward: x_squeezed = adaptive_avg_pool2d(x, 1); _se_reduce -> swish -> _se_expand; sigmoid(.) * x.
// A block owns G images, so every weight it fetches from L2 is used G times (one image per block moved 1.8 GB of filter
// rows per launch at C = 2304); the two small matrix products run as wave-shuffle dot products / per-channel sums in a fixed
// order.  part [n][tiles][C] partial sums (tiles = 1, inv_hw = 1 for a ready-made mean).
template <int G, int NT>
__global__ __launch_bounds__(NT) void se_gate_kernel(const float* __restrict__ part, int tiles, float inv_hw, int n, int C,
                                                      const float* __restrict__ wr, const float* __restrict__ br, int SQ,
                                                      const float* __restrict__ we, int we_ldc, int we_ldj,
                                                      const float* __restrict__ be, float* __restrict__ gate) {
    // (C % 4 == 0: the launcher checks.)  Everything that comes from L2 is fetched as 16-byte vectors in unrolled batches:
    // the kernel is a chain of dependent L2 round trips otherwise (one block streams up to 1.8 MB of filter rows).
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    float* mean = reinterpret_cast<float*>(dsm);   // [G][C]
    float* sq = mean + G * C;                      // [G][SQ]
    const int img0 = blockIdx.x * G, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ng = min(G, n - img0);
    const int C4 = C >> 2;
#pragma unroll
    for (int g = 0; g < G; ++g)
        for (int c4 = tid; c4 < C4; c4 += NT) {
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
            if (g < ng) {
                const float* p = part + (size_t)(img0 + g) * tiles * C + 4 * c4;
#pragma unroll 8
                for (int t = 0; t < tiles; ++t) s += *reinterpret_cast<const f32x4*>(p + (size_t)t * C);
            }
            *reinterpret_cast<f32x4*>(mean + g * C + 4 * c4) = s * inv_hw;
        }
    __syncthreads();
    for (int j = wave; j < SQ; j += NT / 64) {
        const float* w = wr + (size_t)j * C;
        const float bj = br[j];            // (requested with the row, not behind the reduction)
        float s[G];
#pragma unroll
        for (int g = 0; g < G; ++g) s[g] = 0.f;
#pragma unroll 4
        for (int c4 = lane; c4 < C4; c4 += 64) {
            const f32x4 wv = *reinterpret_cast<const f32x4*>(w + 4 * c4);
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const f32x4 m = *reinterpret_cast<const f32x4*>(mean + g * C + 4 * c4);
                s[g] = fmaf(m.x, wv.x, fmaf(m.y, wv.y, fmaf(m.z, wv.z, fmaf(m.w, wv.w, s[g]))));
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            s[g] = adaf_wave_sum(s[g]);
            if (lane == 0) {
                const float v = s[g] + bj;
                sq[g * SQ + j] = v * fast_sigmoid(v);
            }
        }
    }
    __syncthreads();
    if (we_ldc == 1) {
        // transposed expand filter [SQ][C] (the network's layout): a thread owns 4 consecutive channels
        for (int c4 = tid; c4 < C4; c4 += NT) {
            f32x4 s[G];
            const f32x4 b = *reinterpret_cast<const f32x4*>(be + 4 * c4);
#pragma unroll
            for (int g = 0; g < G; ++g) s[g] = b;
            const float* wp = we + 4 * c4;
#pragma unroll 8
            for (int j = 0; j < SQ; ++j) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wp + (size_t)j * we_ldj);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    // (explicit fused multiply-adds: written `s[g] += wv * q` the compiler contracted the products of some images of the
                    //  group and not of others -- a frame's gate depended, in the last bit, on its position in the batch)
                    const float q = sq[g * SQ + j];
                    s[g].x = fmaf(wv.x, q, s[g].x); s[g].y = fmaf(wv.y, q, s[g].y); s[g].z = fmaf(wv.z, q, s[g].z); s[g].w = fmaf(wv.w, q, s[g].w);
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g)
                if (g < ng) {
                    const f32x4 o = {fast_sigmoid(s[g].x), fast_sigmoid(s[g].y), fast_sigmoid(s[g].z), fast_sigmoid(s[g].w)};
                    *reinterpret_cast<f32x4*>(gate + (size_t)(img0 + g) * C + 4 * c4) = o;
                }
        }
        return;
    }
    for (int c = tid; c < C; c += NT) {
        float s[G];
        const float b = be[c];
#pragma unroll
        for (int g = 0; g < G; ++g) s[g] = b;
        const float* wp = we + (size_t)c * we_ldc;
#pragma unroll 8
        for (int j = 0; j < SQ; ++j) {
            const float wv = wp[(size_t)j * we_ldj];
#pragma unroll
            for (int g = 0; g < G; ++g) s[g] = fmaf(sq[g * SQ + j], wv, s[g]);
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
            if (g < ng) gate[(size_t)(img0 + g) * C + c] = fast_sigmoid(s[g]);
    }
}

// ---- project 1x1 with the SE gate on the A operand + BN (+ identity) ---------------------------------------------------
// out[m, n] = (sum_k (x[m, k] * gate[m / HW, k]) * w[n, k]) * scale[n] + bias[n] (+ res[m, n])
// Block = 128 rows x (32 TN) columns, four waves stacked along M, K walked in 128-byte slices (32 floats / 64 halfs) through
// ONE LDS stage with the next slice prefetched in registers -- the gate multiply happens on that register copy.
struct ProjArgs {
    const void* x;        // [M][K]
    const float* gate;    // [M / HW][K] or nullptr (plain 1x1 conv)
    const void* w;        // [N][K], same element type as x
    const float* scale;   // [N] or nullptr
    const float* bias;    // [N] or nullptr
    const void* res;      // [M][N] or nullptr, same element type as out
    void* out;            // [M][N]
    int M, N, K, HW;
    int act;              // ADAF_ACT_* applied after the BN affine (+ identity)
};

template <typename T, int TN>
__global__ __launch_bounds__(256) void gated_project_kernel(const ProjArgs a) {
    constexpr int V = Chunk<T>::V;
    constexpr int BKE = 8 * V;               // k elements per slice
    constexpr int LDP = 36;                  // LDS row pitch in 32-bit words (128 B + 16 B pad: conflict-free b128 access)
    constexpr int BM = 128, BN = 32 * TN;
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const int q = tid & 7, r = tid >> 3;     // 16-byte chunk of the slice, row inside a 32-row staging pass
    const T* xb = static_cast<const T*>(a.x);
    const T* wb = static_cast<const T*>(a.w);
    const T* arow[4];
    const float* grow[4];
    bool aok[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int m = m0 + r + 32 * p;
        aok[p] = m < a.M;
        const int mm = aok[p] ? m : 0;
        arow[p] = xb + (size_t)mm * a.K;
        grow[p] = a.gate ? a.gate + (size_t)(mm / a.HW) * a.K : nullptr;
    }
    const T* brow[TN];
    bool bok[TN];
#pragma unroll
    for (int p = 0; p < TN; ++p) {
        const int n = n0 + r + 32 * p;
        bok[p] = n < a.N;
        brow[p] = wb + (size_t)(bok[p] ? n : 0) * a.K;
    }
    u32x4 ra[4], rb[TN];
    // (every load of a slice is UNCONDITIONAL, from a clamped address, and issued before the first use: written as `if (ok) { load; gate }`
    // per row, hipcc gave each row its own branch with an s_waitcnt vmcnt(0) inside -- four round trips in a row per slice)
    auto gload = [&](int kt) {
        const int k0 = kt * BKE + q * V;
        const bool kok = k0 < a.K;
        const int kc = kok ? k0 : 0;
        u32x4 xv[4];
        f32x4 gv[4][V / 4];
#pragma unroll
        for (int p = 0; p < 4; ++p) xv[p] = *reinterpret_cast<const u32x4*>(arow[p] + kc);
        if (a.gate) {
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int e = 0; e < V / 4; ++e) gv[p][e] = *reinterpret_cast<const f32x4*>(grow[p] + kc + 4 * e);
        }
#pragma unroll
        for (int p = 0; p < TN; ++p) rb[p] = *reinterpret_cast<const u32x4*>(brow[p] + kc);
#pragma unroll
        for (int p = 0; p < TN; ++p)
            if (!(bok[p] && kok)) rb[p] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            u32x4 v = xv[p];
            if (a.gate) {
                float f[V];
                Chunk<T>::unpack(v, f);
#pragma unroll
                for (int e = 0; e < V; e += 4) {
                    const f32x4 g = gv[p][e / 4];
                    f[e] *= g.x; f[e + 1] *= g.y; f[e + 2] *= g.z; f[e + 3] *= g.w;
                }
                v = Chunk<T>::pack(f);
            }
            ra[p] = (aok[p] && kok) ? v : u32x4{0u, 0u, 0u, 0u};
        }
    };
    auto lstore = [&]() {
#pragma unroll
        for (int p = 0; p < 4; ++p) *reinterpret_cast<u32x4*>(&smem[(r + 32 * p) * LDP + q * 4]) = ra[p];
#pragma unroll
        for (int p = 0; p < TN; ++p) *reinterpret_cast<u32x4*>(&smem[(BM + r + 32 * p) * LDP + q * 4]) = rb[p];
    };
    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
    const int nk = (a.K + BKE - 1) / BKE;
    const float* As = smem + (wave * 32 + (lane & 31)) * LDP + (lane >> 5) * 4;
    const float* Bs = smem + (BM + (lane & 31)) * LDP + (lane >> 5) * 4;
    gload(0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();             // the previous slice has been consumed
        lstore();
        __syncthreads();
        if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const f32x4 af = *reinterpret_cast<const f32x4*>(As + kk * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const f32x4 bf = *reinterpret_cast<const f32x4*>(Bs + j * 32 * LDP + kk * 8);
                if constexpr (sizeof(T) == 2) {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af), __builtin_bit_cast(f16x8, bf), acc[j], 0, 0, 0);
                } else {
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc[j], 0, 0, 0);
                }
            }
        }
    }
    // epilogue.  C layout: col = lane & 31, row = (i & 3) + 8 (i >> 2) + 4 (lane >> 5)
    T* ob = static_cast<T*>(a.out);
    const T* rs = static_cast<const T*>(a.res);
    if (a.N % V == 0) {
        // one 32 x 32 sub-tile at a time through a wave-private LDS slab, so that a lane owns V consecutive channels of a row:
        // 16-byte identity loads and output stores (scalar stores of 2-byte halfs were the slowest part of the first version)
        __syncthreads();                         // every wave is done with the operand stage
        constexpr int SP = 36;                   // slab row pitch in floats
        float* slab = smem + wave * 32 * SP;
        constexpr int CPR = 32 / V;              // 16-byte chunks per sub-tile row: 8 (fp32) or 4 (fp16)
        constexpr int RPI = 64 / CPR;            // rows per wave instruction: 8 or 16
        const int cq = lane % CPR, rsub = lane / CPR;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < 16; ++i) slab[((i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)) * SP + (lane & 31)] = acc[j][i];
            __builtin_amdgcn_wave_barrier();
            const int n = n0 + j * 32 + cq * V;
            if (n >= a.N) continue;
            float sc[V], bi[V];
#pragma unroll
            for (int e = 0; e < V; ++e) { sc[e] = a.scale ? a.scale[n + e] : 1.f; bi[e] = a.bias ? a.bias[n + e] : 0.f; }
#pragma unroll
            for (int u = 0; u < 32 / RPI; ++u) {
                const int row = u * RPI + rsub;
                const int m = m0 + wave * 32 + row;
                if (m >= a.M) continue;
                float v[V];
#pragma unroll
                for (int e = 0; e < V; e += 4) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(slab + row * SP + cq * V + e);
                    v[e] = t.x; v[e + 1] = t.y; v[e + 2] = t.z; v[e + 3] = t.w;
                }
#pragma unroll
                for (int e = 0; e < V; ++e) v[e] = fmaf(v[e], sc[e], bi[e]);
                if (rs) {
                    float r[V];
                    Chunk<T>::unpack(*reinterpret_cast<const u32x4*>(rs + (size_t)m * a.N + n), r);
#pragma unroll
                    for (int e = 0; e < V; ++e) v[e] += r[e];
                }
                if (a.act != ADAF_ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < V; ++e) v[e] = act_apply(v[e], a.act);
                }
                *reinterpret_cast<u32x4*>(ob + (size_t)m * a.N + n) = Chunk<T>::pack(v);
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + j * 32 + (lane & 31);
        if (n >= a.N) continue;
        const float sc = a.scale ? a.scale[n] : 1.f, bi = a.bias ? a.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = m0 + wave * 32 + (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5);
            if (m >= a.M) continue;
            float v = fmaf(acc[j][i], sc, bi);
            if (rs) v += (float)rs[(size_t)m * a.N + n];
            ob[(size_t)m * a.N + n] = Chunk<T>::one(act_apply(v, a.act));
        }
    }
}

// ---- stem: 3x3 / stride 2, 3 (4) -> C0 channels, SAME padding, + BN + swish ------------------------------------------------
// On the generic engine this layer is a K = 36 gather GEMM with per-chunk tap arithmetic (0.75 ms per 1024 patches of 144^2,
// 9x its HBM time).  Here a block owns 8 x 16 output pixels: their 17 x 33 input pixels (16 bytes each: the gather's pixel-major
// fp32 frames) are staged once in LDS; wave w owns output rows 2w, 2w+1 = one 32-row MFMA band whose A value for
// k = (tap, channel) is element `channel` of the staged pixel at a compile-time tap offset from the lane's window origin -- one
// ds_read_b128 per tap, two v_mfma_f32_32x32x2_f32 steps per tap (channels 0|1, then 2|3: the k-lane picks its element), the
// filter bank in registers.  Exact fp32 products in both storage modes; the epilogue goes through a wave-private LDS slab so
// every lane stores 16 bytes of consecutive channels.
struct StemArgs {
    const float* x4;      // [n][S][S][4]
    const float* w;       // [C0][3][3][4]
    const float* scale;   // [C0]
    const float* bias;
    void* out;            // [n][OH][OW][C0]
    int n, S, OH, OW, C0, pad, act;
    int tiles_x, tiles_y;
#ifdef EF_TRACE
    unsigned long long* trace;
#endif
};

template <typename T>
__global__ __launch_bounds__(256) void ef_stem_kernel(const StemArgs a) {
    constexpr int V = Chunk<T>::V;
    constexpr int TH = 8, TW = 16, IH = 2 * TH + 1, IW = 2 * TW + 1;
    constexpr int SP = 68;                                   // slab pitch in floats (64 columns + 4)
    __shared__ __attribute__((aligned(16))) float xin[IH * IW * 4];
    __shared__ __attribute__((aligned(16))) float slab[4][32 * SP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    EF_STAMP(0);
    // a block walks the tiles of one 8-row band of one image: the filter bank is fetched once per band
    const int ty = blockIdx.x % a.tiles_y;
    const int img = blockIdx.x / a.tiles_y;
    const int oy0 = ty * TH;
    const float* xb = a.x4 + (size_t)img * a.S * a.S * 4;
    // filter bank: lane (column, k-lane) keeps w[col][2 s + half] for the 18 steps of both 32-column tiles; it travels through
    // the slab (coalesced 16-byte loads) instead of 36 scalar loads per lane
    const int nl = lane & 31, half = lane >> 5;
    float wreg[2][18];
    {
        float* wl = &slab[0][0];
        for (int i = tid; i < a.C0 * 9; i += 256) *reinterpret_cast<f32x4*>(wl + 4 * i) = *reinterpret_cast<const f32x4*>(a.w + 4 * (size_t)i);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = j * 32 + nl;
#pragma unroll
            for (int st = 0; st < 18; ++st) wreg[j][st] = col < a.C0 ? wl[col * 36 + 2 * st + half] : 0.f;
        }
        __syncthreads();
    }
    float sc[2], bi[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = j * 32 + nl;
        sc[j] = col < a.C0 ? a.scale[col] : 0.f;
        bi[j] = col < a.C0 ? a.bias[col] : 0.f;
    }
    const int cpr = a.C0 / V;                                // 16-byte chunks per output pixel
    T* ob = static_cast<T*>(a.out) + (size_t)img * a.OH * a.OW * a.C0;
    float* sl = slab[wave];
    // this lane's pixel of the band: row 2 wave + (nl >> 4), column nl & 15
    const float* win = xin + ((2 * (2 * wave + (nl >> 4))) * IW + 2 * (nl & 15)) * 4;
    // the window of tile tx + 1 travels (in registers) under the products of tile tx: every load unconditional, from a clamped address, zeroed
    // afterwards where it fell into the padding -- written as a guarded load per pixel the three loads of a thread were three round trips in a
    // row at the head of every tile (SQ_WAIT_INST 37 % of the wave cycles, VALU 13 %)
    constexpr int WPT = (IH * IW + 255) / 256;               // window pixels per thread
    f32x4 wv[WPT];
    const int iy0 = oy0 * 2 - a.pad;
    int wr[WPT], wc[WPT];
#pragma unroll
    for (int u = 0; u < WPT; ++u) {
        const int i = tid + 256 * u;
        wr[u] = i / IW; wc[u] = i - wr[u] * IW;
    }
    auto wload = [&](int tx) {
        const int ix0 = tx * TW * 2 - a.pad;
#pragma unroll
        for (int u = 0; u < WPT; ++u) {
            const int iy = iy0 + wr[u], ix = ix0 + wc[u];
            const bool ok = (unsigned)iy < (unsigned)a.S && (unsigned)ix < (unsigned)a.S;
            wv[u] = *reinterpret_cast<const f32x4*>(xb + (ok ? ((size_t)iy * a.S + ix) * 4 : 0));
        }
#pragma unroll
        for (int u = 0; u < WPT; ++u) {
            const int iy = iy0 + wr[u], ix = ix0 + wc[u];
            if (!((unsigned)iy < (unsigned)a.S && (unsigned)ix < (unsigned)a.S)) wv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    wload(0);
    EF_STAMP(1);
    for (int tx = 0; tx < a.tiles_x; ++tx) {
        const int ox0 = tx * TW;
        if (tx) __syncthreads();                              // the previous tile's window has been consumed
#pragma unroll
        for (int u = 0; u < WPT; ++u)
            if (tid + 256 * u < IH * IW) *reinterpret_cast<f32x4*>(xin + (tid + 256 * u) * 4) = wv[u];
        __syncthreads();
        if (tx == 1) EF_STAMP(2);
        if (tx + 1 < a.tiles_x) wload(tx + 1);
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const f32x4 px = *reinterpret_cast<const f32x4*>(win + ((tap / 3) * IW + (tap % 3)) * 4);
            const float a0 = half ? px.y : px.x, a1 = half ? px.w : px.z;       // k = 4 tap + {0|1}, then 4 tap + {2|3}
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, wreg[j][2 * tap], acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, wreg[j][2 * tap + 1], acc[j], 0, 0, 0);
            }
        }
        // epilogue: BN + activation, through the wave's slab, 16-byte stores of V consecutive channels
        if (tx == 1) EF_STAMP(3);
        __builtin_amdgcn_wave_barrier();
        act_switch(a.act, [&](auto AC) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = j * 32 + nl;
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    sl[((i & 3) + 8 * (i >> 2) + 4 * half) * SP + col] = act_of<decltype(AC)::value>(fmaf(acc[j][i], sc[j], bi[j]), a.act);
            }
        });
        __builtin_amdgcn_wave_barrier();
        if (tx == 1) EF_STAMP(4);
        for (int i = lane; i < 32 * cpr; i += 64) {
            const int p = i / cpr, cq = i - p * cpr;
            const int oy = oy0 + 2 * wave + (p >> 4), ox = ox0 + (p & 15);
            if (oy < a.OH && ox < a.OW) {
                float v[V];
#pragma unroll
                for (int e = 0; e < V; e += 4) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(sl + p * SP + cq * V + e);
                    v[e] = t.x; v[e + 1] = t.y; v[e + 2] = t.z; v[e + 3] = t.w;
                }
                *reinterpret_cast<u32x4*>(ob + ((size_t)oy * a.OW + ox) * a.C0 + cq * V) = Chunk<T>::pack(v);
            }
        }
        if (tx == 1) EF_STAMP(5);
        if (tx == 0) EF_STAMP(6);
    }
    EF_STAMP(7);
}

// ---- stem, packed form (C0 <= 48; B3's 40): the same tile walk with the GEMM on the real problem size.  ef_stem_kernel multiplies
// K = 9 taps x 4 channels (one of them zero) x 64 columns; here k = 3 tap + channel runs to 27 (+ one zero: 14 steps of 2), columns 0-31 are
// one 32-column tile, and columns 32-47 a SIXTEEN-column tile on v_mfma_f32_16x16x4_f32 (two 16-row blocks x 7 steps of 4, half the
// passes of the 32x32x2 form): 14 x 64 + 14 x 32 MFMA cycles per band instead of 36 x 64.  A lane's A value for (step, k lane) is ONE float
// of the staged window at an offset that depends on the lane only through its k lane: the offsets are computed once per thread.  The
// filter bank arrives in lane order from a buffer packed at finalize (no LDS round trip, no barrier).  Exact fp32 products as before, and
// BIT-IDENTICAL to ef_stem_kernel: an fp32 MFMA adds its products in k order as a chain of fused multiply-adds, the non-zero products come
// in the same order here, and the ones dropped were exact zeros (tests/test_effnet.py::test_packed_stem_bit_identical_to_the_k36_stem).
struct StemPArgs {
    const float* x4;      // [n][S][S][4]
    const float* wl;      // packed filter bank [64 lanes][24]: 14 values of the 32-column tile, 7 of the 16-column tile, 3 x 0
    const float* scale;   // [C0]
    const float* bias;
    void* out;            // [n][OH][OW][C0]
    int n, S, OH, OW, C0, pad, act;
    int tiles_x, tiles_y;
#ifdef EF_TRACE
    unsigned long long* trace;
#endif
};

template <typename T, bool TILEB>
__global__ __launch_bounds__(256) void ef_stem_packed_kernel(const StemPArgs a) {
    constexpr int V = Chunk<T>::V;
    constexpr int TH = 8, TW = 16, IH = 2 * TH + 1, IW = 2 * TW + 1;
    constexpr int SP = 52;                                   // slab pitch in floats (48 columns + 4)
    __shared__ __attribute__((aligned(16))) float xin[IH * IW * 4];
    __shared__ __attribute__((aligned(16))) float slab[4][32 * SP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    EF_STAMP(0);
    const int ty = blockIdx.x % a.tiles_y;
    const int img = blockIdx.x / a.tiles_y;
    const int oy0 = ty * TH;
    const float* xb = a.x4 + (size_t)img * a.S * a.S * 4;
    const int nl = lane & 31, half = lane >> 5;            // 32x32x2 operands: row / column nl, k lane `half`
    const int r16 = lane & 15, kq = lane >> 4;             // 16x16x4 operands: row / column r16, k lane kq
    // filter bank: six 16-byte loads of this lane's 24 values
    float wA[14], wB[7];
    {
        const f32x4* wp = reinterpret_cast<const f32x4*>(a.wl + lane * 24);
        f32x4 t[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) t[q] = wp[q];
#pragma unroll
        for (int i = 0; i < 14; ++i) wA[i] = t[i >> 2][i & 3];
#pragma unroll
        for (int i = 0; i < 7; ++i) wB[i] = t[(14 + i) >> 2][(14 + i) & 3];
    }
    const float scA = nl < a.C0 ? a.scale[nl] : 0.f, biA = nl < a.C0 ? a.bias[nl] : 0.f;
    const float scB = 32 + r16 < a.C0 ? a.scale[32 + r16] : 0.f, biB = 32 + r16 < a.C0 ? a.bias[32 + r16] : 0.f;
    // window offsets (floats) of k = 3 tap + channel: tap (k / 3) -> row tap / 3, column tap % 3 of the 3 x 3 window, channel k % 3;
    // k = 27 is the zero step (its filter value is 0: any finite A will do -- offset 0)
    auto koff = [](int k) {
        const int kc = k < 27 ? k : 0;
        const int tap = (kc * 11) >> 5, ch = kc - 3 * tap;          // kc / 3 for kc < 32
        const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
        return (ky * IW + kx) * 4 + ch;
    };
    int offA[14], offB[7];
#pragma unroll
    for (int st = 0; st < 14; ++st) offA[st] = koff(2 * st + half);
#pragma unroll
    for (int st = 0; st < 7; ++st) offB[st] = koff(4 * st + kq);
    const int cpr = a.C0 / V;                                // 16-byte chunks per output pixel
    T* ob = static_cast<T*>(a.out) + (size_t)img * a.OH * a.OW * a.C0;
    float* sl = slab[wave];
    // this lane's pixels of the band (row 2 wave + (p >> 4), column p & 15 for band pixel p): p = nl for the 32-row tile, p = 16 rb + r16 for the 16-row blocks
    const float* winA = xin + ((2 * (2 * wave + (nl >> 4))) * IW + 2 * (nl & 15)) * 4;
    const float* winB0 = xin + ((2 * (2 * wave)) * IW + 2 * r16) * 4;
    const float* winB1 = winB0 + 2 * IW * 4;
    constexpr int WPT = (IH * IW + 255) / 256;               // window pixels per thread
    f32x4 wv[WPT];
    const int iy0 = oy0 * 2 - a.pad;
    int wr[WPT], wc[WPT];
#pragma unroll
    for (int u = 0; u < WPT; ++u) {
        const int i = tid + 256 * u;
        wr[u] = i / IW; wc[u] = i - wr[u] * IW;
    }
    auto wload = [&](int tx) {
        const int ix0 = tx * TW * 2 - a.pad;
#pragma unroll
        for (int u = 0; u < WPT; ++u) {
            const int iy = iy0 + wr[u], ix = ix0 + wc[u];
            const bool ok = (unsigned)iy < (unsigned)a.S && (unsigned)ix < (unsigned)a.S;
            wv[u] = *reinterpret_cast<const f32x4*>(xb + (ok ? ((size_t)iy * a.S + ix) * 4 : 0));
        }
#pragma unroll
        for (int u = 0; u < WPT; ++u) {
            const int iy = iy0 + wr[u], ix = ix0 + wc[u];
            if (!((unsigned)iy < (unsigned)a.S && (unsigned)ix < (unsigned)a.S)) wv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    wload(0);
    EF_STAMP(1);
    for (int tx = 0; tx < a.tiles_x; ++tx) {
        const int ox0 = tx * TW;
        if (tx) __syncthreads();                              // the previous tile's window has been consumed
#pragma unroll
        for (int u = 0; u < WPT; ++u)
            if (tid + 256 * u < IH * IW) *reinterpret_cast<f32x4*>(xin + (tid + 256 * u) * 4) = wv[u];
        __syncthreads();
        if (tx == 1) EF_STAMP(2);
        if (tx + 1 < a.tiles_x) wload(tx + 1);
        f32x16 acc;
        f32x4 accB[2];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        accB[0] = accB[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        float av[14], bv0[7], bv1[7];
#pragma unroll
        for (int st = 0; st < 14; ++st) av[st] = winA[offA[st]];
        if constexpr (TILEB) {
#pragma unroll
            for (int st = 0; st < 7; ++st) { bv0[st] = winB0[offB[st]]; bv1[st] = winB1[offB[st]]; }
        }
#pragma unroll
        for (int st = 0; st < 14; ++st) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[st], wA[st], acc, 0, 0, 0);
            if constexpr (TILEB) {
                if (st < 7) {
                    accB[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv0[st], wB[st], accB[0], 0, 0, 0);
                    accB[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv1[st], wB[st], accB[1], 0, 0, 0);
                }
            }
        }
        // epilogue: BN + activation, through the wave's slab, 16-byte stores of V consecutive channels
        if (tx == 1) EF_STAMP(3);
        __builtin_amdgcn_wave_barrier();
        act_switch(a.act, [&](auto AC) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                sl[((i & 3) + 8 * (i >> 2) + 4 * half) * SP + nl] = act_of<decltype(AC)::value>(fmaf(acc[i], scA, biA), a.act);
            if constexpr (TILEB) {
                // 16x16 result: lane (column r16, k lane kq) holds rows 4 kq + i of its block
#pragma unroll
                for (int rb = 0; rb < 2; ++rb)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        sl[(16 * rb + 4 * kq + i) * SP + 32 + r16] = act_of<decltype(AC)::value>(fmaf(accB[rb][i], scB, biB), a.act);
            }
        });
        __builtin_amdgcn_wave_barrier();
        if (tx == 1) EF_STAMP(4);
        for (int i = lane; i < 32 * cpr; i += 64) {
            const int p = i / cpr, cq = i - p * cpr;
            const int oy = oy0 + 2 * wave + (p >> 4), ox = ox0 + (p & 15);
            if (oy < a.OH && ox < a.OW) {
                float v[V];
#pragma unroll
                for (int e = 0; e < V; e += 4) {
                    const f32x4 t = *reinterpret_cast<const f32x4*>(sl + p * SP + cq * V + e);
                    v[e] = t.x; v[e + 1] = t.y; v[e + 2] = t.z; v[e + 3] = t.w;
                }
                *reinterpret_cast<u32x4*>(ob + ((size_t)oy * a.OW + ox) * a.C0 + cq * V) = Chunk<T>::pack(v);
            }
        }
        if (tx == 1) EF_STAMP(5);
        if (tx == 0) EF_STAMP(6);
    }
    EF_STAMP(7);
}

// [C0][3][3][4] (the stem filter as the engine packs it: tap-major, 4 channels) -> the packed kernel's per-lane bank [64][24]:
// o[lane][st] (st < 14) = w[lane & 31][k = 2 st + (lane >> 5)], o[lane][14 + st] (st < 7) = w[32 + (lane & 15)][k = 4 st + (lane >> 4)], k = 3 tap + channel
__global__ void pack_stem_bank_kernel(const float* __restrict__ w, int c0, float* __restrict__ o) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 64 * 24) return;
    const int lane = idx / 24, j = idx - lane * 24;
    int col, k;
    if (j < 14) { col = lane & 31; k = 2 * j + (lane >> 5); }
    else if (j < 21) { col = 32 + (lane & 15); k = 4 * (j - 14) + (lane >> 4); }
    else { o[idx] = 0.f; return; }
    const int tap = k / 3, ch = k - 3 * tap;
    o[idx] = (col < c0 && k < 27) ? w[(size_t)col * 36 + tap * 4 + ch] : 0.f;
}

// [C,1,K,K] (PyTorch depthwise) -> [K*K][C]
__global__ void pack_dw_kxk_kernel(const float* __restrict__ w, int c, int kk, float* __restrict__ o) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= kk * c) return;
    const int ch = idx % c, tap = idx / c;
    o[idx] = w[(size_t)ch * kk + tap];
}
// [R][C] -> [C][R]
__global__ void transpose_kernel(const float* __restrict__ w, int rows, int cols, float* __restrict__ o) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * cols) return;
    const int r = idx / cols, c = idx - r * cols;
    o[(size_t)c * rows + r] = w[idx];
}
// mean over hw pixels of an fp16 / fp32 NHWC map -> fp32 [n][ldo]
template <typename T>
__global__ void avgpool_any_kernel(const T* __restrict__ x, int n, int hw, int c, float* __restrict__ o, int ldo) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * c) return;
    const int ch = idx % c, img = idx / c;
    float s = 0.f;
    const T* p = x + (size_t)img * hw * c + ch;
    for (int i = 0; i < hw; ++i) s += (float)p[(size_t)i * c];
    o[(size_t)img * ldo + ch] = s / (float)hw;
}

// ---- launch planning ---------------------------------------------------------------------------------------------------
struct DwPlan { int V, LPP, CS, slices, OXT, WP, pitch16, TH, TWG, tiles_y, tiles_x, tiles, IMB, PG; size_t lds; };

// LDS per block the tile planner may spend: the tile size trades halo re-reads and per-block overhead against the number of blocks
// a CU can hold, i.e. the loads in flight (measured on the fp16 network, 12 / 20 / 32 / 48 / 64 KB: 16.0 / 12.6 / 12.0 / 12.0 / 12.4 ms)
constexpr size_t kDwLdsBudget = 48 * 1024;

// Tile = TH output rows x TWG groups of OXT outputs, of IMB images, for one channel slice.  Chosen by a small cost model:
// per output, (staged input chunks x the cost of a load + LDS store) + (thread passes x the tap work of a pass), the passes
// counted with their idle lanes -- a tile whose items do not fill the block's lanes pays for the empty ones.
// xp: the staged tile is COMPUTED (fused expand, dw_same_kernel XN > 0): a staged 16-byte chunk costs eight BN + swish evaluations, not a
// load, so the model leans to tiles with little halo.  (Swept on the fp16 network, LDS budget 32 / 48 / 64 / 80 / 100 KB x chunk cost
// 24 / 48 / 96 / 200: 7.00-7.07 ms at 32 KB, 6.67-6.76 everywhere else -- same-box, 7.65 with the two-launch plan.)
bool plan_dw(int C, int OH, int OW, int K, int S, int esize, DwPlan* p, bool xp = false) {
    const size_t budget = kDwLdsBudget;
    const double stage_cost = xp ? 96.0 : 6.0;
    p->V = 16 / esize;
    if (C % p->V) return false;
    const int chunks = C / p->V;
    p->LPP = 1;
    for (int d = 8; d >= 1; --d)
        if (chunks % d == 0) { p->LPP = d; break; }
    p->CS = p->LPP * p->V;
    p->slices = C / p->CS;
    p->OXT = OW % 4 == 0 ? 4 : OW % 3 == 0 ? 3 : OW % 5 == 0 ? 5 : 4;
    const int nxg = (OW + p->OXT - 1) / p->OXT;
    p->pitch16 = p->LPP | 1;                       // odd pitch: consecutive pixels start in different bank groups
    const size_t fixed = (size_t)(K * K + 2) * p->CS * 4;
    const int lanes = 256 / p->LPP;
    auto img_bytes = [&](int th, int twg) { return (size_t)((th - 1) * S + K) * ((twg * p->OXT - 1) * S + K) * p->pitch16 * 16; };
    double best = 1e300;
    p->TH = 0;
    for (int th = 1; th <= OH; ++th) {
        const int ty = (OH + th - 1) / th;
        if ((OH + ty - 1) / ty != th) continue;                 // even tiles only
        for (int twg = 1; twg <= nxg; ++twg) {
            const int tx = (nxg + twg - 1) / twg;
            if ((nxg + tx - 1) / tx != twg) continue;
            const size_t ib = img_bytes(th, twg);
            if (ib + fixed > budget) break;
            const int groups = th * twg;
            int pg = groups < lanes ? groups : lanes;
            int imb = 256 / (p->LPP * pg);
            while (imb > 1 && (size_t)imb * ib + fixed > budget) --imb;
            if (imb < 1) imb = 1;
            const int passes = (groups + pg - 1) / pg;
            const double staged = (double)imb * ((th - 1) * S + K) * ((twg * p->OXT - 1) * S + K) * p->LPP;
            const double cost = (staged * stage_cost + (double)passes * 256 * K * K * p->OXT + 600.0) / ((double)imb * groups * p->OXT * p->LPP);
            if (cost < best) { best = cost; p->TH = th; p->TWG = twg; p->IMB = imb; p->PG = pg; p->tiles_y = ty; p->tiles_x = tx; }
        }
    }
    if (p->TH == 0) return false;
    p->tiles = p->tiles_y * p->tiles_x;
    p->WP = (p->TWG * p->OXT - 1) * S + K;
    p->lds = (size_t)p->IMB * img_bytes(p->TH, p->TWG) + fixed;
    const size_t red = (size_t)256 * p->V * 4;
    if (p->lds < red) p->lds = red;
    return true;
}

template <int K, int S, int OXT, typename T, int XN = 0, int KSX = 1>
void launch_dw_one(const DwArgs& a, size_t lds, hipStream_t s) {
    if (lds > 48 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&dw_same_kernel<K, S, OXT, T, XN, KSX>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((dw_same_kernel<K, S, OXT, T, XN, KSX>), dim3((unsigned)a.total), dim3(256), lds, s, a);
}

template <int K, int S, typename T, int XN = 0, int KSX = 1>
void launch_dw_oxt(const DwArgs& a, int oxt, size_t lds, hipStream_t s) {
    if (oxt == 3) launch_dw_one<K, S, 3, T, XN, KSX>(a, lds, s);
    else if (oxt == 5) launch_dw_one<K, S, 5, T, XN, KSX>(a, lds, s);
    else launch_dw_one<K, S, 4, T, XN, KSX>(a, lds, s);
}

template <typename T, int XN = 0, int KSX = 1>
bool launch_dw_t(const DwArgs& a, int K, int S, int oxt, size_t lds, hipStream_t s) {
    if (K == 3 && S == 1) launch_dw_oxt<3, 1, T, XN, KSX>(a, oxt, lds, s);
    else if (K == 3 && S == 2) launch_dw_oxt<3, 2, T, XN, KSX>(a, oxt, lds, s);
    else if (K == 5 && S == 1) launch_dw_oxt<5, 1, T, XN, KSX>(a, oxt, lds, s);
    else if (K == 5 && S == 2) launch_dw_oxt<5, 2, T, XN, KSX>(a, oxt, lds, s);
    else return false;
    return true;
}

// tiny maps (H = W <= 5, stride 1): one thread per (image, 4 channels); false = not that shape
template <typename T>
static bool launch_dw_small_t(const DsArgs& a, int K, int HW, hipStream_t s) {
    const unsigned grid = (unsigned)(((long long)a.n * (a.C / 4) + 255) / 256);
#define ADAF_DS(K_, HW_) hipLaunchKernelGGL((dw_small_kernel<K_, HW_, T>), dim3(grid), dim3(256), 0, s, a); return true;
    if (K == 3 && HW == 3) { ADAF_DS(3, 3) }
    if (K == 3 && HW == 4) { ADAF_DS(3, 4) }
    if (K == 3 && HW == 5) { ADAF_DS(3, 5) }
    if (K == 5 && HW == 3) { ADAF_DS(5, 3) }
    if (K == 5 && HW == 4) { ADAF_DS(5, 4) }
    if (K == 5 && HW == 5) { ADAF_DS(5, 5) }
#undef ADAF_DS
    return false;
}

}  // namespace

#ifdef EF_TRACE
extern "C" void adaf_ef_set_trace(unsigned long long* p, int c, int k, int s) { ef_trace_buf = p; ef_trace_c = c; ef_trace_k = k; ef_trace_s = s; }
#endif

// Depthwise k x k (k = 3 | 5, stride 1 | 2), padding pad_t / pad_l before the first row / column and whatever the output
// extent needs after the last; returns the number of partial-sum tiles per image (> 0) or < 0.
// pool_part: [n][tiles][c] floats (adaf_effnet_dw_tiles() says how many) or nullptr.
int adaf_effnet_dw_tiles(int c, int oh, int ow, int k, int stride, int dtype) {
    DwPlan p;
    if (!plan_dw(c, oh, ow, k, stride, dtype == ADAF_DTYPE_F16 ? 2 : 4, &p)) return -1;
    return p.tiles;
}

// expand 1x1 + BN + swish -> depthwise k x k + BN + act + squeeze sums in ONE launch (fp16 storage; dw_same_kernel XN > 0).  x: the block
// input [n][hh][ww][cin] (fp16), xw: the expand filter [c][cin] (fp16).  Returns the partial-sum tiles per image (> 0), or < 0 when the
// shape is not the kernel's (cin % 8, cin <= 64, a channel slice of 48 or 64) -- the caller then runs the two launches.
int adaf_effnet_dw_tiles_fused(int c, int oh, int ow, int k, int stride) {
    DwPlan p;
    if (!plan_dw(c, oh, ow, k, stride, 2, &p, true)) return -1;
    return (p.CS == 48 || p.CS == 64) ? p.tiles : -1;
}

int adaf_launch_dw_expand(const void* x, int n, int hh, int ww, int cin, const void* xw, const float* xscale, const float* xbias, int c, int k,
                          int stride, int pad_t, int pad_l, int oh, int ow, const float* wt, const float* scale, const float* bias, int act,
                          void* out, float* pool_part, hipStream_t s) {
    if (cin % 8 || cin > 64 || cin <= 0 || (k != 3 && k != 5) || (stride != 1 && stride != 2)) return -1;
    DwPlan p;
    if (!plan_dw(c, oh, ow, k, stride, 2, &p, true)) return -1;
    if (p.CS != 48 && p.CS != 64) return -1;
    DwArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.out = out; a.wt = wt; a.scale = scale; a.bias = bias; a.pool_part = pool_part;
    a.xw = xw; a.xscale = xscale; a.xbias = xbias; a.cin = cin;
#ifdef EF_TRACE
    a.trace = (ef_trace_c == c && ef_trace_k == k && ef_trace_s == stride) ? ef_trace_buf : nullptr;
    if (a.trace) fprintf(stderr, "dw trace: C %d k %d s %d  TH %d TWG %d tiles %d IMB %d PG %d LPP %d CS %d slices %d WP %d OXT %d lds %zu blocks %d\n", c, k, stride, p.TH, p.TWG, p.tiles, p.IMB, p.PG, p.LPP, p.CS, p.slices, p.WP, p.OXT, p.lds, ((n + p.IMB - 1) / p.IMB) * p.tiles * p.slices);
#endif
    a.n = n; a.H = hh; a.W = ww; a.C = c; a.OH = oh; a.OW = ow; a.pad_t = pad_t; a.pad_l = pad_l; a.act = act;
    a.TH = p.TH; a.tiles = p.tiles; a.TWG = p.TWG; a.tiles_x = p.tiles_x; a.LPP = p.LPP; a.CS = p.CS; a.slices = p.slices;
    a.pitch16 = p.pitch16; a.WP = p.WP;
    a.img_lds = ((p.TH - 1) * stride + k) * p.WP * p.pitch16 * 16;
    a.IMB = p.IMB; a.PG = p.PG; a.igroups = (n + p.IMB - 1) / p.IMB;
    a.total = a.igroups * a.tiles * a.slices;
    const bool k2 = cin > 32;                                // k steps of 32
    const bool ok = p.CS == 48 ? (k2 ? launch_dw_t<_Float16, 3, 2>(a, k, stride, p.OXT, p.lds, s) : launch_dw_t<_Float16, 3, 1>(a, k, stride, p.OXT, p.lds, s))
                               : (k2 ? launch_dw_t<_Float16, 4, 2>(a, k, stride, p.OXT, p.lds, s) : launch_dw_t<_Float16, 4, 1>(a, k, stride, p.OXT, p.lds, s));
    return ok ? p.tiles : -1;
}

int adaf_launch_dw_same(const void* x, int dtype, int n, int hh, int ww, int c, int k, int stride, int pad_t, int pad_l, int oh,
                        int ow, const float* wt, const float* scale, const float* bias, int act, void* out, float* pool_part,
                        hipStream_t s) {
    const bool small_on = (adaf_options().effnet_plan & ADAF_EF_PLAN_TINY_DW) != 0;       // off = dw_same_kernel everywhere (A/B)
    if (small_on && stride == 1 && hh == ww && hh >= 3 && hh <= 5 && oh == hh && ow == ww && pad_t == (k - 1) / 2 && pad_l == pad_t && c % 4 == 0) {
        DsArgs b;
        b.x = x; b.out = out; b.wt = wt; b.scale = scale; b.bias = bias; b.pool_part = pool_part; b.n = n; b.C = c; b.act = act;
        if (dtype == ADAF_DTYPE_F16 ? launch_dw_small_t<_Float16>(b, k, hh, s) : launch_dw_small_t<float>(b, k, hh, s)) return 1;
    }
    DwPlan p;
    if (!plan_dw(c, oh, ow, k, stride, dtype == ADAF_DTYPE_F16 ? 2 : 4, &p)) return -1;
    DwArgs a;
    memset(&a, 0, sizeof(a));
    a.x = x; a.out = out; a.wt = wt; a.scale = scale; a.bias = bias; a.pool_part = pool_part;
#ifdef EF_TRACE
    a.trace = (ef_trace_c == c && ef_trace_k == k && ef_trace_s == stride) ? ef_trace_buf : nullptr;
    if (a.trace) fprintf(stderr, "dw trace: C %d k %d s %d  TH %d TWG %d tiles %d IMB %d PG %d LPP %d CS %d slices %d WP %d OXT %d lds %zu blocks %d\n", c, k, stride, p.TH, p.TWG, p.tiles, p.IMB, p.PG, p.LPP, p.CS, p.slices, p.WP, p.OXT, p.lds, ((n + p.IMB - 1) / p.IMB) * p.tiles * p.slices);
#endif
    a.n = n; a.H = hh; a.W = ww; a.C = c; a.OH = oh; a.OW = ow; a.pad_t = pad_t; a.pad_l = pad_l; a.act = act;
    a.TH = p.TH; a.tiles = p.tiles; a.TWG = p.TWG; a.tiles_x = p.tiles_x; a.LPP = p.LPP; a.CS = p.CS; a.slices = p.slices;
    a.pitch16 = p.pitch16; a.WP = p.WP;
    a.img_lds = ((p.TH - 1) * stride + k) * p.WP * p.pitch16 * 16;
    a.IMB = p.IMB; a.PG = p.PG; a.igroups = (n + p.IMB - 1) / p.IMB;
    a.total = a.igroups * a.tiles * a.slices;
    const bool ok = dtype == ADAF_DTYPE_F16 ? launch_dw_t<_Float16>(a, k, stride, p.OXT, p.lds, s)
                                            : launch_dw_t<float>(a, k, stride, p.OXT, p.lds, s);
    return ok ? p.tiles : -1;
}

// expand 1x1 with K <= 64 on the persistent strip kernel; false = not eligible (the caller falls back to the conv engine)
template <typename T, bool WHOLE>
static bool launch_ef_expand_t(const ExpArgs& a, int blocks, int waves, size_t lds, hipStream_t s) {
    const int ks = a.KP / (2 * Chunk<T>::V);
#define ADAF_EXP(KS_) case KS_: \
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&ef_expand_kernel<T, KS_, WHOLE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((ef_expand_kernel<T, KS_, WHOLE>), dim3((unsigned)blocks), dim3((unsigned)(64 * waves)), lds, s, a); return true;
    switch (ks) {
        ADAF_EXP(1) ADAF_EXP(2) ADAF_EXP(3) ADAF_EXP(4) ADAF_EXP(5) ADAF_EXP(6) ADAF_EXP(7) ADAF_EXP(8)
        default: return false;
    }
#undef ADAF_EXP
}

bool adaf_launch_ef_expand(const void* x, int dtype, long long m, int k, const void* w, const float* scale, const float* bias, int n, int act,
                           void* out, int cus, hipStream_t s) {
    const bool f16 = dtype == ADAF_DTYPE_F16;
    const int v = f16 ? 8 : 4, es = f16 ? 2 : 4;
    if (k > 64 || k % v || n % v || m <= 0 || m > (1ll << 31) - 64) return false;
    ExpArgs a;
    a.x = x; a.w = w; a.scale = scale; a.bias = bias; a.out = out; a.M = (int)m; a.K = k; a.N = n; a.act = act;
    a.KP = (k + 2 * v - 1) / (2 * v) * (2 * v);
    a.NP = (n + 31) / 32 * 32;
    a.strips = (int)((m + 31) / 32);
    const bool whole = f16 && ((size_t)n * es) % 128 != 0;     // (fp32 rows: measured no better)
    const size_t fixed = (size_t)a.NP * (a.KP * es + 16) + (size_t)2 * a.NP * 4;
    const size_t slab = (size_t)32 * ((whole ? a.NP : 64) * es + 16);
    // waves per block / blocks per CU: the most resident waves the LDS allows (each block carries its own copy of the filter)
    int waves = 0, per_cu = 0;
    for (int w = 2; w <= 16; ++w) {
        const size_t need = fixed + w * slab;
        if (need > 150 * 1024) break;
        int b = (int)((160 * 1024) / need);
        if (b * w > 32) b = 32 / w;
        if (b >= 1 && b * w > waves * per_cu) { waves = w; per_cu = b; }
    }
    if (!waves) return false;
    const size_t lds = fixed + waves * slab;
    int blocks = cus * per_cu;
    if ((long long)blocks * waves > a.strips) blocks = (a.strips + waves - 1) / waves;
    if (f16) return whole ? launch_ef_expand_t<_Float16, true>(a, blocks, waves, lds, s) : launch_ef_expand_t<_Float16, false>(a, blocks, waves, lds, s);
    return whole ? launch_ef_expand_t<float, true>(a, blocks, waves, lds, s) : launch_ef_expand_t<float, false>(a, blocks, waves, lds, s);
}

void adaf_launch_pool_finish(const float* part, int n, int tiles, int c, int hw, float* mean, hipStream_t s) {
    hipLaunchKernelGGL(pool_finish_kernel, dim3((unsigned)(((size_t)n * c + 255) / 256)), dim3(256), 0, s, part, n, tiles, c,
                       1.f / (float)hw, mean);
}

void adaf_launch_se_gate(const float* part, int tiles, int hw, int n, int c, const float* wr, const float* br, int sq,
                         const float* we, int we_ldc, int we_ldj, const float* be, float* gate, hipStream_t s) {
    // one block per G images is one block per CU at n = 1024: the squeeze FC is a chain of L2 round trips per wave (SQ / waves filter
    // rows of C floats each), so the wide layers (C >= 512: SQ = 24..96) run 16 waves per block instead of 4
    constexpr int G = 4;
    if (c >= 512)
        hipLaunchKernelGGL((se_gate_kernel<G, 1024>), dim3((unsigned)((n + G - 1) / G)), dim3(1024), (size_t)G * (c + sq) * 4, s, part, tiles,
                           1.f / (float)hw, n, c, wr, br, sq, we, we_ldc, we_ldj, be, gate);
    else
        hipLaunchKernelGGL((se_gate_kernel<G, 256>), dim3((unsigned)((n + G - 1) / G)), dim3(256), (size_t)G * (c + sq) * 4, s, part, tiles,
                           1.f / (float)hw, n, c, wr, br, sq, we, we_ldc, we_ldj, be, gate);
}

// narrow gated project (K <= 64, N <= 32) on the persistent strip kernel; false = not eligible
template <typename T>
static bool launch_nproj_t(const NprojArgs& a, int blocks, size_t lds, hipStream_t s) {
    const int ks = a.KP / (2 * Chunk<T>::V);
#define ADAF_NP(KS_) case KS_: hipLaunchKernelGGL((ef_nproj_kernel<T, KS_>), dim3((unsigned)blocks), dim3(512), lds, s, a); return true;
    switch (ks) {
        ADAF_NP(1) ADAF_NP(2) ADAF_NP(3) ADAF_NP(4) ADAF_NP(5) ADAF_NP(6) ADAF_NP(7) ADAF_NP(8)
        default: return false;
    }
#undef ADAF_NP
}

static bool launch_narrow_project(const void* x, int dtype, long long m, int hw, int k, const float* gate, const void* w, int n, const float* scale,
                                  const float* bias, const void* res, void* out, int act, int cus, hipStream_t s) {
    const bool on = (adaf_options().effnet_plan & ADAF_EF_PLAN_STRIP_PROJECT) != 0;     // off = gated_project_kernel (A/B)
    const bool f16 = dtype == ADAF_DTYPE_F16;
    const int v = f16 ? 8 : 4, es = f16 ? 2 : 4;
    if (!on || !gate || k > 64 || n > 32 || k % v || n % v || m < 32ll * 8 * cus || m > (1ll << 31) - 64) return false;
    NprojArgs a;
    a.x = x; a.gate = gate; a.w = w; a.scale = scale; a.bias = bias; a.res = res; a.out = out;
    a.M = (int)m; a.K = k; a.N = n; a.HW = hw; a.act = act;
    a.KP = (k + 2 * v - 1) / (2 * v) * (2 * v);
    a.strips = (int)((m + 31) / 32);
    const size_t lds = (size_t)32 * (a.KP * es + 16) + 64 * 4 + (size_t)8 * 32 * 36 * 4;
    const int blocks = cus * 3;
    return f16 ? launch_nproj_t<_Float16>(a, blocks, lds, s) : launch_nproj_t<float>(a, blocks, lds, s);
}

int adaf_launch_gated_project(const void* x, int dtype, int m, int hw, int k, const float* gate, const void* w, int n,
                              const float* scale, const float* bias, const void* res, void* out, hipStream_t s, int act = ADAF_ACT_NONE) {
    const int v = dtype == ADAF_DTYPE_F16 ? 8 : 4;
    if (k % v || m <= 0 || n <= 0 || hw <= 0) return -1;
    {
        static const int cus = [] { int d = 0, c = 256; hipDeviceProp_t p; if (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&p, d) == hipSuccess) c = p.multiProcessorCount; return c; }();
        if (launch_narrow_project(x, dtype, m, hw, k, gate, w, n, scale, bias, res, out, act, cus, s)) return 1;
    }
    ProjArgs a;
    a.x = x; a.gate = gate; a.w = w; a.scale = scale; a.bias = bias; a.res = res; a.out = out; a.M = m; a.N = n; a.K = k; a.HW = hw;
    a.act = act;
    // column tile: the fewest padded columns, then the fewest column tiles (the A panel is re-read per column tile).  fp16 storage
    // with enough row tiles to fill the device twice over: ONE column tile if 160 / 192 / 256 columns hold the layer -- every column
    // tile re-reads AND re-gates the A panel (8 conversions + 8 multiplies + the repack per 16 bytes), the products are nearly free
    // (blocks 14-17 of B3: N = 136 as five 32-wide tiles 120 us, as one 160-wide tile 100 us; with 200 row tiles, blocks 19-23, the
    // single tile loses as much again, so they keep two).
    static const int cus = [] { int d = 0, c = 256; hipDeviceProp_t p; if (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&p, d) == hipSuccess) c = p.multiProcessorCount; return c; }();
    int best = 1, best_pad = 1 << 30;
    for (int tn = 1; tn <= 4; ++tn) {
        const int bn = 32 * tn, pad = (n + bn - 1) / bn * bn;
        if (pad < best_pad || (pad == best_pad && tn > best)) { best_pad = pad; best = tn; }
    }
    if (dtype == ADAF_DTYPE_F16 && (m + 127) / 128 >= 2 * cus && n > 32 * best) {
        for (int tn : {5, 6, 8})
            if (32 * tn >= n) { best = tn; break; }
    }
    const dim3 block(256);
    const dim3 grid((unsigned)((m + 127) / 128), (unsigned)((n + 32 * best - 1) / (32 * best)));
#define ADAF_GP(T, TN) hipLaunchKernelGGL((gated_project_kernel<T, TN>), grid, block, 0, s, a)
    if (dtype == ADAF_DTYPE_F16) {
        switch (best) {
            case 1: ADAF_GP(_Float16, 1); break;
            case 2: ADAF_GP(_Float16, 2); break;
            case 3: ADAF_GP(_Float16, 3); break;
            case 4: ADAF_GP(_Float16, 4); break;
            case 5: ADAF_GP(_Float16, 5); break;
            case 6: ADAF_GP(_Float16, 6); break;
            default: ADAF_GP(_Float16, 8); break;
        }
    } else {
        if (best == 1) ADAF_GP(float, 1); else if (best == 2) ADAF_GP(float, 2); else if (best == 3) ADAF_GP(float, 3); else ADAF_GP(float, 4);
    }
#undef ADAF_GP
    return best;
}

// returns false when the shape is not the one the stem kernel is written for (C0 <= 64, chunks of whole 16 bytes)
void adaf_launch_pack_stem_bank(const float* w, int c0, float* o, hipStream_t s) {
    hipLaunchKernelGGL(pack_stem_bank_kernel, dim3(6), dim3(256), 0, s, w, c0, o);
}

// bank: the packed filter bank of adaf_launch_pack_stem_bank (C0 <= 48: the packed kernel) or nullptr
bool adaf_launch_ef_stem(const float* x4, int dtype, int n, int size, int oh, int ow, int pad, const float* w, const float* bank, const float* scale,
                         const float* bias, int c0, int act, void* out, hipStream_t s) {
    const int v = dtype == ADAF_DTYPE_F16 ? 8 : 4;
    if (c0 > 64 || c0 % v) return false;
    const dim3 block(256);
    if (bank && c0 <= 48 && (adaf_options().effnet_plan & ADAF_EF_PLAN_PACKED_STEM)) {
        StemPArgs a;
        a.x4 = x4; a.wl = bank; a.scale = scale; a.bias = bias; a.out = out; a.n = n; a.S = size; a.OH = oh; a.OW = ow; a.C0 = c0; a.pad = pad; a.act = act;
        a.tiles_x = (ow + 15) / 16; a.tiles_y = (oh + 7) / 8;
#ifdef EF_TRACE
        a.trace = ef_trace_k == -1 ? ef_trace_buf : nullptr;
#endif
        const dim3 grid((unsigned)((size_t)n * a.tiles_y));
        if (dtype == ADAF_DTYPE_F16) {
            if (c0 > 32) hipLaunchKernelGGL((ef_stem_packed_kernel<_Float16, true>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((ef_stem_packed_kernel<_Float16, false>), grid, block, 0, s, a);
        } else {
            if (c0 > 32) hipLaunchKernelGGL((ef_stem_packed_kernel<float, true>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((ef_stem_packed_kernel<float, false>), grid, block, 0, s, a);
        }
        return true;
    }
    StemArgs a;
    a.x4 = x4; a.w = w; a.scale = scale; a.bias = bias; a.out = out; a.n = n; a.S = size; a.OH = oh; a.OW = ow; a.C0 = c0; a.pad = pad; a.act = act;
    a.tiles_x = (ow + 15) / 16; a.tiles_y = (oh + 7) / 8;
#ifdef EF_TRACE
    a.trace = ef_trace_k == -1 ? ef_trace_buf : nullptr;         // (adaf_ef_set_trace(buf, 0, -1, 0) traces the stem)
#endif
    const dim3 grid((unsigned)((size_t)n * a.tiles_y));
    if (dtype == ADAF_DTYPE_F16) hipLaunchKernelGGL((ef_stem_kernel<_Float16>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((ef_stem_kernel<float>), grid, block, 0, s, a);
    return true;
}

void adaf_launch_pack_dw_kxk(const float* w, int c, int k, float* o, hipStream_t s) {
    hipLaunchKernelGGL(pack_dw_kxk_kernel, dim3((unsigned)((k * k * c + 255) / 256)), dim3(256), 0, s, w, c, k * k, o);
}

// ======================================================================================================================
// The network object
// ======================================================================================================================
struct EfConv {
    std::string name;     // "stem", "b3.expand", "b3.dw", "b3.project", "head"
    int cin, cout, k, stride;
    bool dw;
    int cin_pad;
    float* w = nullptr;   // dense: [cout][k][k][cin_pad] fp32; depthwise: [k*k][c]
    void* w16 = nullptr;  // dense 1x1 filters in fp16 (ADAF_DTYPE_F16)
    float* scale = nullptr;
    float* bias = nullptr;
};
struct EfBlock {
    int k, stride, expand_ratio, cin, cout, hid, sq;
    int expand, dwc, project;       // indices into convs (expand = -1 when the ratio is 1)
    float* se_wr = nullptr;         // [sq][hid]
    float* se_br = nullptr;         // [sq]
    float* se_wet = nullptr;        // [sq][hid] (transposed _se_expand.weight)
    float* se_be = nullptr;         // [hid]
    void* wef = nullptr;            // fp16 storage: expand / project filters in MFMA B-fragment order for the whole-block kernel
    void* wpf = nullptr;            //  (mbconv_whole.hip)
    float* wdl = nullptr;           //  ... and the depthwise taps + folded BN as one row per channel
};

struct adaf_effnet {
    adaf_handle* h = nullptr;
    float width = 1.f, depth = 1.f;
    std::map<std::string, std::pair<const float*, size_t>> params;
    std::vector<EfConv> convs;
    std::vector<EfBlock> blocks;
    int stem = 0, head = 0, feat = 1280;
    int dtype = ADAF_DTYPE_F32;
    // MBConv blocks whose map is small enough for a workgroup to own whole images run as ONE launch (mbconv_whole.hip; fp16 storage,
    // stride 1, maps up to 9 x 9).  On by default; adaf_effnet_set_fusion(net, 0) restores the four-launch plan (tests, A/B).
    bool fuse = true;
    bool finalized = false;
    float* stem_bank = nullptr;      // the stem filter in the packed kernel's lane order (C0 <= 48; ef_stem_packed_kernel)
    // Two patch chunks travel through the network side by side (ADAF_EF_PLAN_PAIR_CHUNKS; the second on a library-owned stream forked from
    // and joined to the caller's stream by events, one helper per caller stream -- as adaf_mobilenetv2 does): the launches of the 9 x 9 and
    // 5 x 5 stages, the SE gates and every launch's ramp and tail leave room that a neighbour fills.
    AdafAuxPool aux;        // (adaf_internal.h: LRU over caller streams, mutex-guarded)
    // Every derived weight buffer (packed filters, folded BN, SE matrices, B fragments) is carved out of a few large slabs: a
    // launch of the whole-block kernel reads ~14 of them, and as separate small hipMalloc()s each sat in pages of its own.
    std::vector<void*> slabs;
    char* slab_cur = nullptr;
    size_t slab_left = 0;
    void* carve(size_t bytes) {
        bytes = (bytes + 255) & ~(size_t)255;
        if (bytes > slab_left) {
            const size_t sz = bytes > ((size_t)32 << 20) ? bytes : ((size_t)32 << 20);
            void* p = nullptr;
            if (hipMalloc(&p, sz) != hipSuccess) return nullptr;
            slabs.push_back(p);
            slab_cur = static_cast<char*>(p);
            slab_left = sz;
        }
        void* r = slab_cur;
        slab_cur += bytes;
        slab_left -= bytes;
        return r;
    }
};

namespace {

int efail(adaf_handle* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    return code;
}

// efficientnet_pytorch utils.py round_filters / round_repeats (depth_divisor 8)
int round_filters(int filters, float width) {
    const double f = (double)filters * (double)width;
    int nf = (int)(f + 4.0) / 8 * 8;
    if (nf < 8) nf = 8;
    if ((double)nf < 0.9 * f) nf += 8;
    return nf;
}
int round_repeats(int r, float depth) { return (int)std::ceil((double)depth * r - 1e-9); }

// blocks_args of utils.py efficientnet(): repeats, kernel, stride, expand, in, out (se_ratio 0.25 everywhere)
const int kBase[7][6] = {{1, 3, 1, 1, 32, 16}, {2, 3, 2, 6, 16, 24}, {2, 5, 2, 6, 24, 40}, {3, 3, 2, 6, 40, 80},
                         {3, 5, 1, 6, 80, 112}, {4, 5, 2, 6, 112, 192}, {1, 3, 1, 6, 192, 320}};

void build(adaf_effnet* net) {
    net->convs.clear();
    net->blocks.clear();
    const int c0 = round_filters(32, net->width);
    net->stem = 0;
    net->convs.push_back({"stem", 3, c0, 3, 2, false, 4});
    int last = c0;
    for (auto& s : kBase) {
        const int i0 = round_filters(s[4], net->width), o = round_filters(s[5], net->width), rep = round_repeats(s[0], net->depth);
        for (int j = 0; j < rep; ++j) {
            EfBlock b;
            b.k = s[1]; b.stride = j == 0 ? s[2] : 1; b.expand_ratio = s[3];
            b.cin = j == 0 ? i0 : o; b.cout = o; b.hid = b.cin * s[3];
            b.sq = (int)(b.cin * 0.25); if (b.sq < 1) b.sq = 1;
            char nm[40];
            const int bi = (int)net->blocks.size();
            b.expand = -1;
            if (s[3] != 1) {
                snprintf(nm, sizeof(nm), "b%d.expand", bi);
                b.expand = (int)net->convs.size();
                net->convs.push_back({nm, b.cin, b.hid, 1, 1, false, b.cin});
            }
            snprintf(nm, sizeof(nm), "b%d.dw", bi);
            b.dwc = (int)net->convs.size();
            net->convs.push_back({nm, b.hid, b.hid, b.k, b.stride, true, b.hid});
            snprintf(nm, sizeof(nm), "b%d.project", bi);
            b.project = (int)net->convs.size();
            net->convs.push_back({nm, b.hid, b.cout, 1, 1, false, b.hid});
            net->blocks.push_back(b);
            last = o;
        }
    }
    net->feat = round_filters(1280, net->width);
    net->head = (int)net->convs.size();
    net->convs.push_back({"head", last, net->feat, 1, 1, false, last});
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
// Conv2dStaticSamePadding (utils.py): padding for an input of `size`; returns pad_before, writes the total
inline int same_pad(int size, int k, int stride, int* total) {
    const int out = ceil_div(size, stride);
    int t = (out - 1) * stride + k - size;
    if (t < 0) t = 0;
    *total = t;
    return t / 2;
}
inline int conv_out_len(int in, int k, int stride, int pad_total) { return (in + pad_total - k) / stride + 1; }

// elements per frame of the chunk buffers: io (block input / output, ping-pong), ex (expanded), dw (depthwise output);
// pc = the largest tiles * hid of a squeeze partial buffer
void slab_sizes(const adaf_effnet* net, int size, int pad_size, int dtype, size_t* io, size_t* ex, size_t* dw, size_t* pc, size_t* gc) {
    int tot;
    (void)same_pad(pad_size, 3, 2, &tot);
    int hw = conv_out_len(size, 3, 2, tot), ps = ceil_div(pad_size, 2);
    *io = (size_t)hw * hw * net->convs[net->stem].cout;
    *ex = *dw = *pc = *gc = 0;
    for (auto& b : net->blocks) {
        (void)same_pad(ps, b.k, b.stride, &tot);
        const int ohw = conv_out_len(hw, b.k, b.stride, tot);
        if (b.expand >= 0 && (size_t)hw * hw * b.hid > *ex) *ex = (size_t)hw * hw * b.hid;
        if ((size_t)ohw * ohw * b.hid > *dw) *dw = (size_t)ohw * ohw * b.hid;
        if ((size_t)ohw * ohw * b.cout > *io) *io = (size_t)ohw * ohw * b.cout;
        int tiles = adaf_effnet_dw_tiles(b.hid, ohw, ohw, b.k, b.stride, dtype);
        if (dtype == ADAF_DTYPE_F16) {                       // (the fused expand + depthwise launch has a tile plan of its own)
            const int tf = adaf_effnet_dw_tiles_fused(b.hid, ohw, ohw, b.k, b.stride);
            if (tf > tiles) tiles = tf;
        }
        if ((size_t)(tiles > 0 ? tiles : 1) * b.hid > *pc) *pc = (size_t)(tiles > 0 ? tiles : 1) * b.hid;
        if ((size_t)b.hid > *gc) *gc = b.hid;
        hw = ohw;
        ps = ceil_div(ps, b.stride);
    }
    // the head's fp32 map is parked in the expanded-map slab when the caller does not want it
    const size_t headf = (size_t)hw * hw * net->feat * 4 / (dtype == ADAF_DTYPE_F16 ? 2 : 4);
    if (headf > *ex) *ex = headf;
}

int chunk_frames(int n) {
    const int c = adaf_options().effnet_chunk > 0 ? adaf_options().effnet_chunk : 1024;
    // a batch that fits one chunk but holds >= 512 patches travels as a pair of half chunks (see adaf_effnet::Aux): 1024 patches of
    // 144^2 6.21 -> 5.95 ms (tools/effnet_streams_probe.py); a patch's arithmetic does not depend on the chunk it travels in
    if ((adaf_options().effnet_plan & ADAF_EF_PLAN_PAIR_CHUNKS) && n <= c && n >= 512) return (n + 1) / 2;
    return n < c ? n : c;
}

int run_dense(adaf_effnet* net, const EfConv& L, const void* in, bool in16, int n, int hh, int ww, int oh, int ow, int pad, int act,
              void* out, bool out16, hipStream_t st) {
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.x = static_cast<const float*>(in); a.w = in16 ? static_cast<const float*>(L.w16) : L.w;
    a.scale = L.scale; a.bias = L.bias; a.res = nullptr; a.out = static_cast<float*>(out);
    a.M = n * oh * ow; a.N = L.cout; a.K = L.k * L.k * L.cin_pad;
    a.cin = L.cin_pad; a.H = hh; a.W = ww; a.OH = oh; a.OW = ow; a.KH = a.KW = L.k; a.stride = L.stride; a.pad = pad;
    a.ldx = L.cin_pad; a.ldo = L.cout; a.ldr = L.cout; a.act = act;
    a.zeros = net->h->zeros;
    a.vec_epi = (L.cout % 4 == 0) ? 1 : 0;
    a.in16 = in16; a.out16 = out16; a.res16 = 0;
    return adaf_launch_conv_gemm(a, 0, net->h->cus, st) > 0 ? ADAF_OK : ADAF_E_LAUNCH;
}

}  // namespace

extern "C" {

int adaf_effnet_create(adaf_handle* h, float width_coefficient, float depth_coefficient, adaf_effnet** out) {
    if (!h || !out) return ADAF_E_BADARG;
    if (!(width_coefficient >= 0.5f && width_coefficient <= 4.f && depth_coefficient >= 0.5f && depth_coefficient <= 8.f))
        return efail(h, ADAF_E_BADARG, "effnet: width / depth coefficients out of range");
    adaf_effnet* net = new adaf_effnet();
    net->h = h;
    net->width = width_coefficient;
    net->depth = depth_coefficient;
    build(net);
    *out = net;
    return ADAF_OK;
}

int adaf_effnet_destroy(adaf_effnet* net) {
    if (!net) return ADAF_OK;
    for (void* p : net->slabs) (void)hipFree(p);
    net->aux.destroy();
    delete net;
    return ADAF_OK;
}

// Does the whole-block kernel take this block?  Its geometry: the SAME padding of the map as it is (pad rows in front as the kernel
// computes them, output = ceil(hw / stride)): stride 1 everywhere it is instantiated, stride 2 for the 9 x 9 -> 5 x 5 block.
static bool ef_whole_geometry(const EfBlock& b, int hw, int ohw, int pbd) {
    return b.expand >= 0 && (b.stride == 1 || b.stride == 2) && ohw == (hw + b.stride - 1) / b.stride && pbd == adaf_mbw_pad_before(hw, b.k, b.stride) &&
           adaf_mbw_eligible(hw, b.k, b.stride, b.cin, b.hid, b.cout, b.sq);
}

int adaf_effnet_feature_dim(const adaf_effnet* net) { return net ? net->feat : 0; }
int adaf_effnet_block_count(const adaf_effnet* net) { return net ? (int)net->blocks.size() : 0; }

int adaf_effnet_block_info(const adaf_effnet* net, int block, int* info8) {
    if (!net || !info8 || block < 0 || block >= (int)net->blocks.size()) return ADAF_E_BADARG;
    const EfBlock& b = net->blocks[block];
    info8[0] = b.k; info8[1] = b.stride; info8[2] = b.expand_ratio; info8[3] = b.cin; info8[4] = b.cout; info8[5] = b.hid;
    info8[6] = b.sq; info8[7] = net->convs[net->stem].cout;
    return ADAF_OK;
}

int adaf_effnet_set_fusion(adaf_effnet* net, int on) {
    if (!net) return ADAF_E_BADARG;
    net->fuse = on != 0;
    return ADAF_OK;
}

int adaf_effnet_whole_blocks(const adaf_effnet* net, int size, int pad_size) {
    if (!net || size < 32) return 0;
    if (pad_size <= 0) pad_size = size;
    if (net->dtype != ADAF_DTYPE_F16 || !net->fuse || !(adaf_options().effnet_plan & ADAF_EF_PLAN_WHOLE_BLOCK)) return 0;
    int tot, count = 0;
    (void)same_pad(pad_size, 3, 2, &tot);
    int hw = conv_out_len(size, 3, 2, tot), ps = ceil_div(pad_size, 2);
    for (auto& b : net->blocks) {
        const int pbd = same_pad(ps, b.k, b.stride, &tot);
        const int ohw = conv_out_len(hw, b.k, b.stride, tot);
        if (ef_whole_geometry(b, hw, ohw, pbd)) ++count;
        hw = ohw;
        ps = ceil_div(ps, b.stride);
    }
    return count;
}

int adaf_effnet_fused_expand_blocks(const adaf_effnet* net, int size, int pad_size) {
    if (!net || size < 32) return 0;
    if (pad_size <= 0) pad_size = size;
    const unsigned plan = adaf_options().effnet_plan;
    if (net->dtype != ADAF_DTYPE_F16 || !(plan & ADAF_EF_PLAN_FUSED_EXPAND)) return 0;
    const bool whole_on = net->fuse && (plan & ADAF_EF_PLAN_WHOLE_BLOCK);
    int tot, count = 0;
    (void)same_pad(pad_size, 3, 2, &tot);
    int hw = conv_out_len(size, 3, 2, tot), ps = ceil_div(pad_size, 2);
    for (size_t bi = 0; bi < net->blocks.size(); ++bi) {
        const EfBlock& b = net->blocks[bi];
        const int pbd = same_pad(ps, b.k, b.stride, &tot);
        const int ohw = conv_out_len(hw, b.k, b.stride, tot);
        const bool whole = whole_on && ef_whole_geometry(b, hw, ohw, pbd);
        if (!whole && b.expand >= 0 && bi < 32 && ((adaf_options().effnet_fused_blocks >> bi) & 1u) && b.cin % 8 == 0 && b.cin <= 64 &&
            adaf_effnet_dw_tiles_fused(b.hid, ohw, ohw, b.k, b.stride) > 0)
            ++count;
        hw = ohw;
        ps = ceil_div(ps, b.stride);
    }
    return count;
}

int adaf_effnet_set_dtype(adaf_effnet* net, int dtype) {
    if (!net) return ADAF_E_BADARG;
    if (dtype != ADAF_DTYPE_F32 && dtype != ADAF_DTYPE_F16) return efail(net->h, ADAF_E_BADARG, "effnet: unknown dtype %d", dtype);
    if (dtype != net->dtype) net->finalized = false;
    net->dtype = dtype;
    return ADAF_OK;
}

int adaf_effnet_set_param(adaf_effnet* net, const char* name, const float* dev_ptr, size_t numel) {
    if (!net || !name || !dev_ptr) return ADAF_E_BADARG;
    net->params[name] = std::make_pair(dev_ptr, numel);
    net->finalized = false;
    return ADAF_OK;
}

int adaf_effnet_finalize(adaf_effnet* net, void* stream) {
    if (!net) return ADAF_E_BADARG;
    adaf_handle* h = net->h;
    hipStream_t st = (hipStream_t)stream;
    auto get = [&](const std::string& key, size_t numel, const float** p) -> int {
        auto it = net->params.find(key);
        if (it == net->params.end()) return efail(h, ADAF_E_STATE, "effnet: missing parameter '%s'", key.c_str());
        if (it->second.second != numel)
            return efail(h, ADAF_E_BADARG, "effnet: '%s' has %zu elements, expected %zu", key.c_str(), it->second.second, numel);
        *p = it->second.first;
        return ADAF_OK;
    };
    auto alloc = [&](float** p, size_t count) -> int {
        if (!*p && !(*p = static_cast<float*>(net->carve(count * sizeof(float))))) return efail(h, ADAF_E_NOMEM, "effnet: hipMalloc");
        return ADAF_OK;
    };
    for (auto& L : net->convs) {
        const float *w, *g, *b, *m, *v;
        int rc;
        const size_t wn_in = L.dw ? (size_t)L.cout * L.k * L.k : (size_t)L.cout * L.cin * L.k * L.k;
        if ((rc = get(L.name + ".weight", wn_in, &w))) return rc;
        if ((rc = get(L.name + ".bn.weight", L.cout, &g))) return rc;
        if ((rc = get(L.name + ".bn.bias", L.cout, &b))) return rc;
        if ((rc = get(L.name + ".bn.running_mean", L.cout, &m))) return rc;
        if ((rc = get(L.name + ".bn.running_var", L.cout, &v))) return rc;
        const size_t wn = L.dw ? (size_t)L.k * L.k * L.cout : (size_t)L.cout * L.k * L.k * L.cin_pad;
        if ((rc = alloc(&L.w, wn)) || (rc = alloc(&L.scale, L.cout)) || (rc = alloc(&L.bias, L.cout))) return rc;
        if (L.dw) adaf_launch_pack_dw_kxk(w, L.cout, L.k, L.w, st);
        else adaf_launch_pack_weight(w, L.cout, L.cin, L.k, L.k, L.cin_pad, L.w, st);
        if (!L.dw && L.k == 1 && net->dtype == ADAF_DTYPE_F16) {
            if (!L.w16 && !(L.w16 = net->carve(wn * sizeof(unsigned short)))) return efail(h, ADAF_E_NOMEM, "effnet: hipMalloc");
            adaf_launch_pack_weight_f16(w, L.cout, L.cin, 1, 1, L.cin_pad, L.w16, st);
        }
        adaf_launch_fold_bn(g, b, m, v, 1e-3f, L.cout, L.scale, L.bias, st);     // utils.py: batch_norm_epsilon = 1e-3
    }
    {
        const EfConv& S = net->convs[net->stem];
        if (S.cout <= 48 && S.k == 3 && S.cin_pad == 4) {
            if (!net->stem_bank && !(net->stem_bank = static_cast<float*>(net->carve(64 * 24 * 4)))) return efail(h, ADAF_E_NOMEM, "effnet: hipMalloc");
            adaf_launch_pack_stem_bank(S.w, S.cout, net->stem_bank, st);
        }
    }
    for (size_t bi = 0; bi < net->blocks.size(); ++bi) {
        EfBlock& b = net->blocks[bi];
        char nm[48];
        const float *wr, *br, *we, *be;
        int rc;
        snprintf(nm, sizeof(nm), "b%zu.se_reduce", bi);
        if ((rc = get(std::string(nm) + ".weight", (size_t)b.sq * b.hid, &wr)) || (rc = get(std::string(nm) + ".bias", b.sq, &br))) return rc;
        snprintf(nm, sizeof(nm), "b%zu.se_expand", bi);
        if ((rc = get(std::string(nm) + ".weight", (size_t)b.sq * b.hid, &we)) || (rc = get(std::string(nm) + ".bias", b.hid, &be))) return rc;
        if ((rc = alloc(&b.se_wr, (size_t)b.sq * b.hid)) || (rc = alloc(&b.se_br, b.sq)) || (rc = alloc(&b.se_wet, (size_t)b.sq * b.hid)) ||
            (rc = alloc(&b.se_be, b.hid)))
            return rc;
        (void)hipMemcpyAsync(b.se_wr, wr, (size_t)b.sq * b.hid * 4, hipMemcpyDeviceToDevice, st);
        (void)hipMemcpyAsync(b.se_br, br, (size_t)b.sq * 4, hipMemcpyDeviceToDevice, st);
        (void)hipMemcpyAsync(b.se_be, be, (size_t)b.hid * 4, hipMemcpyDeviceToDevice, st);
        hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((b.sq * b.hid + 255) / 256)), dim3(256), 0, st, we, b.hid, b.sq, b.se_wet);
        if (net->dtype == ADAF_DTYPE_F16 && b.expand >= 0 && (b.stride == 1 || b.stride == 2) && b.cin % 8 == 0 && b.hid % 16 == 0) {
            // the whole-block kernel streams its filters as ready-made B fragments (one coalesced 1 KB load per wave instruction)
            const float *wexp, *wproj;
            if ((rc = get(net->convs[b.expand].name + ".weight", (size_t)b.hid * b.cin, &wexp)) ||
                (rc = get(net->convs[b.project].name + ".weight", (size_t)b.cout * b.hid, &wproj)))
                return rc;
            if (!b.wef && !(b.wef = net->carve(adaf_mbw_bfrag_halfs(b.hid, b.cin, true) * 2))) return efail(h, ADAF_E_NOMEM, "effnet: hipMalloc");
            if (!b.wpf && !(b.wpf = net->carve(adaf_mbw_bfrag_halfs(b.cout, b.hid, false) * 2))) return efail(h, ADAF_E_NOMEM, "effnet: hipMalloc");
            adaf_launch_pack_bfrag_f16(wexp, b.hid, b.cin, true, b.wef, st);
            adaf_launch_pack_bfrag_f16(wproj, b.cout, b.hid, false, b.wpf, st);
            if (!b.wdl && !(b.wdl = static_cast<float*>(net->carve((size_t)b.hid * adaf_mbw_tap_row(b.k) * 4)))) return efail(h, ADAF_E_NOMEM, "effnet: hipMalloc");
            const EfConv& D = net->convs[b.dwc];
            adaf_launch_pack_dw_rows(D.w, D.scale, D.bias, b.hid, b.k, b.wdl, st);
        }
    }
    hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess) return efail(h, ADAF_E_LAUNCH, "effnet finalize: %s", hipGetErrorString(e));
    net->aux.prepare(4);        // helper streams exist before the first forward (which may be captured into a HIP graph)
    net->finalized = true;
    return ADAF_OK;
}

size_t adaf_effnet_workspace_bytes(const adaf_effnet* net, int n, int size, int pad_size) {
    if (!net || n <= 0 || size < 32) return 0;
    size_t io, ex, dw, pc, gc;
    slab_sizes(net, size, pad_size > 0 ? pad_size : size, net->dtype, &io, &ex, &dw, &pc, &gc);
    const size_t es = net->dtype == ADAF_DTYPE_F16 ? 2 : 4;
    const int chunk = chunk_frames(n);
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t per_chunk = al((size_t)chunk * io * es) * 2 + al((size_t)chunk * ex * es) + al((size_t)chunk * dw * es) + al((size_t)chunk * pc * 4) +
                             al((size_t)chunk * gc * 4);
    return per_chunk * (n > chunk ? 2 : 1);      // two chunks in flight (whether or not the pairing is switched on)
}

int adaf_effnet_forward(adaf_effnet* net, const float* frames_nhwc4, int n, int size, int pad_size, int upto_block, void* block_out,
                        float* featmap, float* featvec, int ldvec, void* ws, size_t ws_bytes, void* stream) {
    if (!net) return ADAF_E_BADARG;
    adaf_handle* h = net->h;
    if (!net->finalized) return efail(h, ADAF_E_STATE, "effnet: finalize() has not been called");
    if (!frames_nhwc4 || !ws) return efail(h, ADAF_E_BADARG, "effnet: null pointer");
    if (n <= 0 || size < 32) return efail(h, ADAF_E_BADARG, "effnet: need n > 0 and size >= 32");
    if (pad_size <= 0) pad_size = size;
    if (featvec && ldvec < net->feat) return efail(h, ADAF_E_LAYOUT, "effnet: ldvec >= %d required", net->feat);
    if (upto_block >= 0 && !block_out) return efail(h, ADAF_E_BADARG, "effnet: upto_block needs block_out");
    if (ws_bytes < adaf_effnet_workspace_bytes(net, n, size, pad_size)) return efail(h, ADAF_E_NOMEM, "effnet: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    const bool f16 = net->dtype == ADAF_DTYPE_F16;
    const size_t es = f16 ? 2 : 4;
    size_t io, ex, dws, pc, gc;
    slab_sizes(net, size, pad_size, net->dtype, &io, &ex, &dws, &pc, &gc);
    const int chunk = chunk_frames(n);
    auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
    const size_t per_chunk = al((size_t)chunk * io * es) * 2 + al((size_t)chunk * ex * es) + al((size_t)chunk * dws * es) + al((size_t)chunk * pc * 4) +
                             al((size_t)chunk * gc * 4);
    // one chunk of patches through the whole network on stream `st` with its own part of the workspace
    auto run_chunk = [&](int f0, int nc, char* base, hipStream_t st) -> int {
        char* bufA = base;
        char* bufB = bufA + al((size_t)chunk * io * es);
        char* bufE = bufB + al((size_t)chunk * io * es);
        char* bufD = bufE + al((size_t)chunk * ex * es);
        float* part = reinterpret_cast<float*>(bufD + al((size_t)chunk * dws * es));
        float* gate = reinterpret_cast<float*>(reinterpret_cast<char*>(part) + al((size_t)chunk * pc * 4));
        int rc, tot;
        const int pb0 = same_pad(pad_size, 3, 2, &tot);
        int hw = conv_out_len(size, 3, 2, tot), ps = ceil_div(pad_size, 2);
        char* cur = bufA;
        char* nxt = bufB;
        const EfConv& S = net->convs[net->stem];
        const unsigned plan = adaf_options().effnet_plan;
        const bool own_stem = (plan & ADAF_EF_PLAN_OWN_STEM) != 0;       // off = the generic engine (A/B)
        if (!(own_stem && adaf_launch_ef_stem(frames_nhwc4 + (size_t)f0 * size * size * 4, net->dtype, nc, size, hw, hw, pb0, S.w, net->stem_bank, S.scale, S.bias,
                                              S.cout, ADAF_ACT_SWISH, cur, st)) &&
            (rc = run_dense(net, S, frames_nhwc4 + (size_t)f0 * size * size * 4, false, nc, size, size, hw, hw, pb0, ADAF_ACT_SWISH, cur, f16, st)))
            return efail(h, rc, "effnet: stem launch");
        size_t out_elems = (size_t)hw * hw * S.cout;
        for (size_t bi = 0; bi < net->blocks.size(); ++bi) {
            if (upto_block >= 0 && (int)bi >= upto_block) break;
            const EfBlock& b = net->blocks[bi];
            const void* dw_in = cur;
            const int pbd = same_pad(ps, b.k, b.stride, &tot);
            const int ohw = conv_out_len(hw, b.k, b.stride, tot);
            const EfConv& D = net->convs[b.dwc];
            const EfConv& P = net->convs[b.project];
            const bool skip = b.stride == 1 && b.cin == b.cout;
            if (f16 && net->fuse && (plan & ADAF_EF_PLAN_WHOLE_BLOCK) && b.wef && ef_whole_geometry(b, hw, ohw, pbd)) {
                // maps small enough for a workgroup to own whole images: the block as ONE launch (mbconv_whole.hip)
                const EfConv& E = net->convs[b.expand];
                if (adaf_launch_mbconv_whole(cur, nc, hw, b.stride, b.cin, b.hid, b.cout, b.sq, b.k, b.wef, E.scale, E.bias, b.wdl, b.se_wr,
                                             b.se_br, b.se_wet, b.se_be, b.wpf, P.scale, P.bias, skip, nxt, st)) {
                    char* t = cur; cur = nxt; nxt = t;
                    out_elems = (size_t)ohw * ohw * b.cout;
                    hw = ohw;
                    ps = ceil_div(ps, b.stride);
                    continue;
                }
            }
            int tiles = -1;
            if (tiles <= 0) {
                if (b.expand >= 0 && f16 && (plan & ADAF_EF_PLAN_FUSED_EXPAND) && bi < 32 && ((adaf_options().effnet_fused_blocks >> bi) & 1u)) {
                    // expand computed inside the depthwise launch's staging step: no expanded map in HBM
                    const EfConv& E = net->convs[b.expand];
                    tiles = adaf_launch_dw_expand(cur, nc, hw, hw, b.cin, E.w16, E.scale, E.bias, b.hid, b.k, b.stride, pbd, pbd, ohw, ohw, D.w,
                                                  D.scale, D.bias, ADAF_ACT_SWISH, bufD, part, st);
                }
                if (tiles <= 0 && b.expand >= 0) {
                    // (the expand GEMM stays on the conv engine: routed through gated_project_kernel -- 128 x 128 tiles, swish in its
                    // 16-byte epilogue -- the fp16 network measured 12.6 instead of 11.5 ms)
                    const EfConv& E = net->convs[b.expand];
                    const bool own_expand = (plan & ADAF_EF_PLAN_STRIP_EXPAND) != 0;   // off = the conv engine for every expand (A/B)
                    if (!(own_expand && adaf_launch_ef_expand(cur, net->dtype, (long long)nc * hw * hw, b.cin, f16 ? E.w16 : static_cast<const void*>(E.w),
                                                              E.scale, E.bias, b.hid, ADAF_ACT_SWISH, bufE, net->h->cus, st)) &&
                        (rc = run_dense(net, E, cur, f16, nc, hw, hw, hw, hw, 0, ADAF_ACT_SWISH, bufE, f16, st)))
                        return efail(h, rc, "effnet: expand launch (block %zu)", bi);
                    dw_in = bufE;
                }
                if (tiles <= 0)
                    tiles = adaf_launch_dw_same(dw_in, net->dtype, nc, hw, hw, b.hid, b.k, b.stride, pbd, pbd, ohw, ohw, D.w, D.scale,
                                                D.bias, ADAF_ACT_SWISH, bufD, part, st);
            }
            if (tiles <= 0) return efail(h, ADAF_E_LAUNCH, "effnet: depthwise launch (block %zu)", bi);
            adaf_launch_se_gate(part, tiles, ohw * ohw, nc, b.hid, b.se_wr, b.se_br, b.sq, b.se_wet, 1, b.hid, b.se_be, gate, st);
            if (adaf_launch_gated_project(bufD, net->dtype, nc * ohw * ohw, ohw * ohw, b.hid, gate, f16 ? P.w16 : static_cast<const void*>(P.w),
                                          b.cout, P.scale, P.bias, skip ? cur : nullptr, nxt, st) < 0)
                return efail(h, ADAF_E_LAUNCH, "effnet: project launch (block %zu)", bi);
            char* t = cur; cur = nxt; nxt = t;
            hw = ohw;
            ps = ceil_div(ps, b.stride);
            out_elems = (size_t)hw * hw * b.cout;
        }
        if (upto_block >= 0) {
            (void)hipMemcpyAsync(static_cast<char*>(block_out) + (size_t)f0 * out_elems * es, cur, (size_t)nc * out_elems * es,
                                 hipMemcpyDeviceToDevice, st);
            return ADAF_OK;
        }
        float* fm = featmap ? featmap + (size_t)f0 * hw * hw * net->feat : reinterpret_cast<float*>(bufE);
        // fp16 storage, pooled features only: the head conv with the global average pool in its epilogue (conv_gemm.hip
        // adaf_launch_conv_pool16: no fp32 map -- 157 MB per 1024 patches of 144^2 -- and one launch; the bits of conv + avgpool_kernel)
        if (f16 && featvec && !featmap && (plan & ADAF_EF_PLAN_HEAD_POOL) && net->feat % 4 == 0 && ldvec % 4 == 0) {
            const EfConv& Hc = net->convs[net->head];
            ConvArgs a;
            memset(&a, 0, sizeof(a));
            a.x = reinterpret_cast<const float*>(cur); a.w = reinterpret_cast<const float*>(Hc.w16);
            a.scale = Hc.scale; a.bias = Hc.bias;
            a.M = nc * hw * hw; a.N = Hc.cout; a.K = Hc.cin_pad;
            a.cin = Hc.cin_pad; a.H = a.OH = hw; a.W = a.OW = hw; a.KH = a.KW = 1; a.stride = 1;
            a.ldx = Hc.cin_pad; a.ldo = Hc.cout; a.ldr = Hc.cout; a.act = ADAF_ACT_SWISH;
            a.zeros = net->h->zeros; a.vec_epi = 1; a.in16 = 1;
            if (adaf_launch_conv_pool16(a, hw * hw, featvec + (size_t)f0 * ldvec, ldvec, st)) return ADAF_OK;
        }
        if ((rc = run_dense(net, net->convs[net->head], cur, f16, nc, hw, hw, hw, hw, 0, ADAF_ACT_SWISH, fm, false, st)))
            return efail(h, rc, "effnet: head launch");
        if (featvec) {
            if (net->feat % 4 == 0 && ldvec % 4 == 0) adaf_launch_avgpool(fm, nc, hw * hw, net->feat, featvec + (size_t)f0 * ldvec, ldvec, st);
            else hipLaunchKernelGGL((avgpool_any_kernel<float>), dim3((unsigned)(((size_t)nc * net->feat + 255) / 256)), dim3(256), 0, st, fm, nc,
                                    hw * hw, net->feat, featvec + (size_t)f0 * ldvec, ldvec);
        }
        return ADAF_OK;
    };
    AdafAuxPool::Aux* ax = ((adaf_options().effnet_plan & ADAF_EF_PLAN_PAIR_CHUNKS) != 0 && n > chunk) ? net->aux.get(st) : nullptr;
    const bool pair = ax != nullptr;
    char* base0 = static_cast<char*>(ws);
    for (int f0 = 0; f0 < n; f0 += chunk) {
        const int nc = (n - f0) < chunk ? (n - f0) : chunk;
        int rc;
        if (pair && f0 + chunk < n) {
            const int f1 = f0 + chunk;
            const int nc1 = (n - f1) < chunk ? (n - f1) : chunk;
            (void)hipEventRecord(ax->ev_fork, st);
            (void)hipStreamWaitEvent(ax->stream, ax->ev_fork, 0);
            rc = run_chunk(f0, nc, base0, st);
            if (!rc) rc = run_chunk(f1, nc1, base0 + per_chunk, ax->stream);
            // ALWAYS join, also when a launch failed after the fork: whatever the helper stream still has queued writes the caller's
            // workspace / outputs, and the caller may reuse them as soon as this call returns
            (void)hipEventRecord(ax->ev_join, ax->stream);
            (void)hipStreamWaitEvent(st, ax->ev_join, 0);
            if (rc) return rc;
            f0 = f1;
        } else if ((rc = run_chunk(f0, nc, base0, st))) return rc;
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ADAF_OK : efail(h, ADAF_E_LAUNCH, "effnet forward: %s", hipGetErrorString(e));
}

// ---- stand-alone ops (tests, and building blocks for other MBConv networks) ---------------------------------------------
int adaf_pack_dw_weight_kxk_f32(adaf_handle* h, const float* w_c1kk, int channels, int k, float* w_kkc, void* stream) {
    if (!h || !w_c1kk || !w_kkc || channels <= 0 || k <= 0) return efail(h, ADAF_E_BADARG, "pack_dw_kxk: bad argument");
    adaf_launch_pack_dw_kxk(w_c1kk, channels, k, w_kkc, (hipStream_t)stream);
    return ADAF_OK;
}

size_t adaf_dwconv_same_workspace_bytes(int n, int hh, int ww, int c, int k, int stride, int dtype) {
    const int oh = ceil_div(hh, stride), ow = ceil_div(ww, stride);
    const int tiles = adaf_effnet_dw_tiles(c, oh, ow, k, stride, dtype);
    // (at least one partial per (image, channel): the tiny-map kernel writes n * c sums whatever the tile planner says -- ADVICE r3)
    return (size_t)n * (tiles > 0 ? tiles : 1) * c * 4;
}

int adaf_dwconv_same_bn_act(adaf_handle* h, const void* x, int dtype, int n, int hh, int ww, int c, int k, int stride, const float* w_kkc,
                            const float* scale, const float* bias, int act, void* out, float* pool_mean, void* ws, size_t ws_bytes,
                            void* stream) {
    if (!h || !x || !w_kkc || !scale || !bias || !out) return efail(h, ADAF_E_BADARG, "dwconv_same: null pointer");
    if (n <= 0 || hh <= 0 || ww <= 0 || c <= 0 || (k != 3 && k != 5) || (stride != 1 && stride != 2))
        return efail(h, ADAF_E_BADARG, "dwconv_same: k in {3, 5}, stride in {1, 2}");
    if (dtype != ADAF_DTYPE_F32 && dtype != ADAF_DTYPE_F16) return efail(h, ADAF_E_BADARG, "dwconv_same: dtype");
    if (c % (dtype == ADAF_DTYPE_F16 ? 8 : 4)) return efail(h, ADAF_E_LAYOUT, "dwconv_same: channels must fill 16-byte chunks");
    if (act < ADAF_ACT_NONE || act > ADAF_ACT_SWISH) return efail(h, ADAF_E_BADARG, "dwconv_same: activation");
    int ty, tx;
    const int pt = same_pad(hh, k, stride, &ty), pl = same_pad(ww, k, stride, &tx);
    const int oh = ceil_div(hh, stride), ow = ceil_div(ww, stride);
    float* part = nullptr;
    if (pool_mean) {
        if (!ws || ws_bytes < adaf_dwconv_same_workspace_bytes(n, hh, ww, c, k, stride, dtype)) return efail(h, ADAF_E_NOMEM, "dwconv_same: workspace");
        part = static_cast<float*>(ws);
    }
    const int tiles = adaf_launch_dw_same(x, dtype, n, hh, ww, c, k, stride, pt, pl, oh, ow, w_kkc, scale, bias, act, out, part, (hipStream_t)stream);
    if (tiles <= 0) return efail(h, ADAF_E_LAYOUT, "dwconv_same: shape not supported");
    if (pool_mean) adaf_launch_pool_finish(part, n, tiles, c, oh * ow, pool_mean, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ADAF_OK : efail(h, ADAF_E_LAUNCH, "dwconv_same: %s", hipGetErrorString(e));
}

int adaf_se_gate_f32(adaf_handle* h, const float* pool_mean, int n, int c, const float* w_reduce, const float* b_reduce, int squeezed,
                     const float* w_expand, const float* b_expand, float* gate, void* stream) {
    if (!h || !pool_mean || !w_reduce || !b_reduce || !w_expand || !b_expand || !gate) return efail(h, ADAF_E_BADARG, "se_gate: null pointer");
    if (n <= 0 || c <= 0 || c % 4 || squeezed <= 0 || (size_t)(c + squeezed) * 16 > 60 * 1024) return efail(h, ADAF_E_BADARG, "se_gate: extents (c %% 4 == 0)");
    adaf_launch_se_gate(pool_mean, 1, 1, n, c, w_reduce, b_reduce, squeezed, w_expand, squeezed, 1, b_expand, gate, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ADAF_OK : efail(h, ADAF_E_LAUNCH, "se_gate: %s", hipGetErrorString(e));
}

int adaf_conv1x1_gated_bn(adaf_handle* h, const void* x, int dtype, int n_images, int hw, int cin, const float* gate, const void* w,
                          int cout, const float* scale, const float* bias, const void* residual, void* out, void* stream) {
    if (!h || !x || !w || !out) return efail(h, ADAF_E_BADARG, "conv1x1_gated: null pointer");
    if (dtype != ADAF_DTYPE_F32 && dtype != ADAF_DTYPE_F16) return efail(h, ADAF_E_BADARG, "conv1x1_gated: dtype");
    if (n_images <= 0 || hw <= 0 || cin <= 0 || cout <= 0 || cin % (dtype == ADAF_DTYPE_F16 ? 8 : 4))
        return efail(h, ADAF_E_LAYOUT, "conv1x1_gated: cin must fill 16-byte chunks");
    if ((long long)n_images * hw > 0x7fffffffLL) return efail(h, ADAF_E_BADARG, "conv1x1_gated: too many rows");
    if (adaf_launch_gated_project(x, dtype, n_images * hw, hw, cin, gate, w, cout, scale, bias, residual, out, (hipStream_t)stream) < 0)
        return efail(h, ADAF_E_LAYOUT, "conv1x1_gated: shape not supported");
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ADAF_OK : efail(h, ADAF_E_LAUNCH, "conv1x1_gated: %s", hipGetErrorString(e));
}

}  // extern "C"
