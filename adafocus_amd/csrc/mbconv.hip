// Fused expand (1x1 conv + BN + ReLU6) -> depthwise 3x3 (+ BN + ReLU6) of a MobileNetV2 inverted-residual
// block (ACT/models/mobilenet.py:42-68; STH/models/mobilenetv2.py): the 6x-expanded feature map never goes
// to HBM.  For the glancer's high-resolution blocks (b2..b7, 112^2..28^2 maps) that tensor is the largest
// transfer of the whole network (4.8 MB per 224^2 frame for b2 alone, written once and read once).
//
// One block = one spatial tile of one frame:
//   stride 1: 8x8 outputs <- 10x10 input halo;  stride 2: 3x8 outputs <- 7x17 input halo  (<= 128 halo pixels)
//   1. the halo pixels' Cin input channels go to LDS once;
//   2. per chunk of 32 hidden channels:  E[halo][32] = X[halo][Cin] * We[32][Cin]^T on the fp32 matrix pipe
//      (one 32x32 tile per wave, same k order as the conv engine -> bit-identical to the unfused expand),
//      BN + ReLU6, halo pixels outside the image forced to 0 (the depthwise conv pads the EXPANDED map),
//      E to LDS;  then the 3x3 depthwise taps on the VALU straight from LDS (same tap order as
//      dwconv3x3_kernel), BN + ReLU6, 16-byte stores.
// The halo is recomputed by neighbouring tiles (1.56x / 1.24x of the expand FLOPs, which are ~1/10 of the
// block's traffic-equivalent cost at these channel counts).
#include <cstdlib>
#include "adaf_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

template <int S>
__global__ __launch_bounds__(256) void mb_expand_dw_kernel(const MbFuseArgs a) {
    constexpr int TH = S == 1 ? 8 : 3, TW = 8;
    constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
    constexpr int HP = IH * IW;     // 100 / 119 halo pixels
    constexpr int MP = 128;         // padded to four 32-row bands, one per wave
    constexpr int EP = 36;          // E row pitch in floats
    static_assert(HP <= MP, "halo must fit four bands");
    extern __shared__ __attribute__((aligned(16))) float smem[];   // X[MP][cin+4] | We chunk [32][cin+4] | E[MP][EP]
    float* Xs = smem;
    float* Ws = smem + MP * (a.cin + 4);
    float* Es = Ws + 32 * (a.cin + 4);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bid = blockIdx.x;
    const int tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int img = bid / a.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
    const int cq = a.cin >> 2;      // 16-byte chunks per pixel
    const int xp = a.cin + 4;       // LDS pitch of X and We rows: conflict-free b128 fragment reads for cin = 16, 24, 32
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    for (int idx = tid; idx < MP * cq; idx += 256) {
        const int p = idx / cq, c = idx - p * cq;
        const int iy = iy0 + p / IW, ix = ix0 + p % IW;
        const bool ok = p < HP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const f32x4 v = *reinterpret_cast<const f32x4*>(ok ? a.x + (((size_t)img * a.H + iy) * a.W + ix) * a.cin + 4 * c : a.zeros);
        *reinterpret_cast<f32x4*>(&Xs[p * xp + 4 * c]) = v;
    }
    // which of this lane's 16 accumulator rows are halo pixels inside the image
    unsigned emask = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int p = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int iy = iy0 + p / IW, ix = ix0 + p % IW;
        if (p < HP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) emask |= 1u << r;
    }
    const int c4 = tid & 7, pg = tid >> 3;
    const int fr = (32 * wave + (lane & 31)) * xp + 4 * (lane >> 5);
    const int fw = (lane & 31) * xp + 4 * (lane >> 5);

    // expand weights of a chunk: 32 rows x cq chunks <= 256 -> at most one 16-byte piece per thread
    const int wr = tid / cq, wc = tid - wr * cq;
    const bool wmine = tid < 32 * cq;
    auto wload = [&](int ch0) -> f32x4 {
        return (wmine && ch0 + wr < a.hid) ? *reinterpret_cast<const f32x4*>(a.we + (size_t)(ch0 + wr) * a.cin + 4 * wc) : zero4;
    };
    if (wmine) *reinterpret_cast<f32x4*>(&Ws[wr * xp + 4 * wc]) = wload(0);

    for (int ch0 = 0; ch0 < a.hid; ch0 += 32) {
        // everything this chunk and the next need from global memory is requested before the first barrier, so its
        // latency runs under the expand GEMM instead of in front of the depthwise phase / the next chunk
        const f32x4 wnext = wload(ch0 + 32);
        const int ch = ch0 + 4 * c4;
        const int chs = ch < a.hid ? ch : 0;
        f32x4 k[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) k[t] = *reinterpret_cast<const f32x4*>(a.wd + (size_t)t * a.hid + chs);
        const f32x4 dsc = *reinterpret_cast<const f32x4*>(a.sd + chs);
        const f32x4 dbi = *reinterpret_cast<const f32x4*>(a.bd + chs);
        const int nch = ch0 + (lane & 31);
        const bool nv = nch < a.hid;
        const float esc = nv ? a.se[nch] : 0.f, ebi = nv ? a.be[nch] : 0.f;
        __syncthreads();
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int kk = 0; kk < a.cin / 8; ++kk) {
            const f32x4 af = *reinterpret_cast<const f32x4*>(&Xs[fr + 8 * kk]);
            const f32x4 bf = *reinterpret_cast<const f32x4*>(&Ws[fw + 8 * kk]);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc, 0, 0, 0);
        }
        {
            const float sc = esc, bi = ebi;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float e = fminf(fmaxf(fmaf(acc[r], sc, bi) + 0.f, 0.f), 6.f);
                Es[p * EP + (lane & 31)] = ((emask >> r) & 1u) ? e : 0.f;
            }
        }
        __syncthreads();
        if (wmine) *reinterpret_cast<f32x4*>(&Ws[wr * xp + 4 * wc]) = wnext;   // every wave is past its reads of this chunk's weights
        if (ch < a.hid) {
            const f32x4 sc = dsc, bi = dbi;
#pragma unroll
            for (int o = pg; o < TH * TW; o += 32) {
                const int oy = o / TW, ox = o - oy * TW;
                const int gy = oy0 + oy, gx = ox0 + ox;
                f32x4 s = zero4;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(&Es[((oy * S + ky) * IW + ox * S + kx) * EP + 4 * c4]);
                        const f32x4 w = k[ky * 3 + kx];
                        s.x = fmaf(v.x, w.x, s.x);
                        s.y = fmaf(v.y, w.y, s.y);
                        s.z = fmaf(v.z, w.z, s.z);
                        s.w = fmaf(v.w, w.w, s.w);
                    }
                if (gy < a.OH && gx < a.OW) {
                    f32x4 r;
                    r.x = fminf(fmaxf(fmaf(s.x, sc.x, bi.x), 0.f), 6.f);
                    r.y = fminf(fmaxf(fmaf(s.y, sc.y, bi.y), 0.f), 6.f);
                    r.z = fminf(fmaxf(fmaf(s.z, sc.z, bi.z), 0.f), 6.f);
                    r.w = fminf(fmaxf(fmaf(s.w, sc.w, bi.w), 0.f), 6.f);
                    *reinterpret_cast<f32x4*>(a.out + (((size_t)img * a.OH + gy) * a.OW + gx) * a.hid + ch) = r;
                }
            }
        }
        // no barrier here: the next chunk's barrier (after its weight load) orders these reads of E before the next writes
    }
}

// ---------------------------------------------------------------------------------------------------------
// The same fused block with WAVE-PRIVATE tiles: every wave owns a small output tile (6x6 at stride 1, 3x4 at stride 2: a halo of
// 64 / 63 pixels = two MFMA row bands) and runs expand -> BN/ReLU6 -> E in its own 9 KB of LDS -> depthwise taps -> stores on
// its own, with no block-level barrier anywhere.  The kernel above is bound by the latency of its barrier-separated phase
// chain; here the chain is private to a wave and the other waves of the CU fill its gaps.  The halo pixels'
// input channels are loaded straight into MFMA A fragments (no X image in LDS), the expand filter rows into B fragments.
// Costs: 1.8x (stride 1) / 1.3x (stride 2) of the expand products are halo recomputation.  Same products in the same order
// per output as the kernels above: bit-identical.
template <int S, int CIN>
__global__ __launch_bounds__(256, 3) void mb_expand_dw_w_kernel(const MbFuseArgs a) {
    constexpr int OTH = S == 1 ? 6 : 3, OTW = S == 1 ? 6 : 4;
    constexpr int IH = (OTH - 1) * S + 3, IW = (OTW - 1) * S + 3;
    constexpr int HP = IH * IW;             // 64 / 63 halo pixels
    constexpr int EP = 36, KK = CIN / 8;
    constexpr int NIT = OTH * OTW * 8;      // (output pixel, 4-channel group) items per chunk
    constexpr int NR = (NIT + 63) / 64;
    static_assert(HP <= 64, "halo = two row bands");
    constexpr int HMAX = 192;               // hidden channels the LDS-resident taps / affines are sized for (launcher checks)
    __shared__ __attribute__((aligned(16))) float Eall[4][64 * EP];
    __shared__ __attribute__((aligned(16))) float Wd[9 * HMAX];
    __shared__ __attribute__((aligned(16))) float Bn[4 * HMAX];       // expand scale | expand bias | depthwise scale | depthwise bias
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the block's only cooperative step: depthwise taps and both BN affines to LDS (a chunk then reads them at LDS latency
    // instead of waiting for L2 once per chunk); zero past `hid` so the last, partial chunk needs no selects
    for (int idx = tid; idx < 9 * HMAX; idx += 256) {
        const int t = idx / HMAX, c = idx - t * HMAX;
        Wd[idx] = c < a.hid ? a.wd[(size_t)t * a.hid + c] : 0.f;
    }
    for (int idx = tid; idx < HMAX; idx += 256) {
        const bool v = idx < a.hid;
        Bn[idx] = v ? a.se[idx] : 0.f;
        Bn[HMAX + idx] = v ? a.be[idx] : 0.f;
        Bn[2 * HMAX + idx] = v ? a.sd[idx] : 0.f;
        Bn[3 * HMAX + idx] = v ? a.bd[idx] : 0.f;
    }
    __syncthreads();
    float* Ew = Eall[wave];
    const int per_img = a.tiles_y * a.tiles_x;
    const int half = lane >> 5, nl = lane & 31;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int c4 = lane & 7, og = lane >> 3;
    float* e0 = Ew + (4 * half) * EP + nl;
    // one tile per wave (a loop over several tiles per wave, to amortise the staging above, was measured: the compiler keeps
    // tile-invariant values live across it, 168 VGPRs with spills, and every block shape got slower)
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= a.n * per_img) return;
    const int img = tile / per_img;
    const int rem = tile - img * per_img;
    const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    const int oy0 = ty * OTH, ox0 = tx * OTW;
    const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
    const bool inner = iy0 >= 0 && ix0 >= 0 && iy0 + IH <= a.H && ix0 + IW <= a.W;

    // A fragments of the two row bands: halo pixel p = 32 b + nl, k chunk 8 kk + 4 half
    f32x4 af[2][KK];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int p = 32 * b + nl;
        const int iy = iy0 + p / IW, ix = ix0 + p % IW;
        const bool ok = p < HP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const float* src = a.x + (((size_t)img * a.H + (ok ? iy : 0)) * a.W + (ok ? ix : 0)) * CIN + 4 * half;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) af[b][kk] = ok ? *reinterpret_cast<const f32x4*>(src + 8 * kk) : zero4;
    }
    // which accumulator rows are halo pixels inside the image (border tiles only)
    unsigned emask = 0xffffffffu;
    if (!inner) {
        emask = 0;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = 32 * b + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int iy = iy0 + p / IW, ix = ix0 + p % IW;
                if (p < HP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) emask |= 1u << (16 * b + r);
            }
    }

    // expand filter rows as B fragments, one chunk ahead of their use (from L2)
    auto wfrag = [&](int ch0, f32x4 (&w)[KK]) {
        const int nch = ch0 + nl;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
            w[kk] = nch < a.hid ? *reinterpret_cast<const f32x4*>(a.we + (size_t)nch * CIN + 8 * kk + 4 * half) : zero4;
    };
    f32x4 bf[KK], bnext[KK];
    wfrag(0, bf);
    for (int ch0 = 0; ch0 < a.hid; ch0 += 32) {
        wfrag(ch0 + 32, bnext);             // (past the last chunk: zeros, no loads)
        const float esc = Bn[ch0 + nl], ebi = Bn[HMAX + ch0 + nl];
        const int ch = ch0 + 4 * c4;
        const bool cv = ch < a.hid;
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][kk][s4], bf[kk][s4], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][kk][s4], bf[kk][s4], acc1, 0, 0, 0);
            }
        }
        __builtin_amdgcn_wave_barrier();        // (the previous chunk's depthwise reads of E are issued: same-wave LDS ops run in order)
        {
            const f32x2 sc2 = {esc, esc}, bi2 = {ebi, ebi};
            if (inner) {
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 v0 = __builtin_elementwise_fma(f32x2{acc0[r], acc0[r + 1]}, sc2, bi2);
                    const f32x2 v1 = __builtin_elementwise_fma(f32x2{acc1[r], acc1[r + 1]}, sc2, bi2);
                    const int o0 = ((r & 3) + 8 * (r >> 2)) * EP, o1 = (((r + 1) & 3) + 8 * ((r + 1) >> 2)) * EP;
                    e0[o0] = __builtin_amdgcn_fmed3f(v0.x, 0.f, 6.f);
                    e0[o1] = __builtin_amdgcn_fmed3f(v0.y, 0.f, 6.f);
                    e0[32 * EP + o0] = __builtin_amdgcn_fmed3f(v1.x, 0.f, 6.f);
                    e0[32 * EP + o1] = __builtin_amdgcn_fmed3f(v1.y, 0.f, 6.f);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = ((r & 3) + 8 * (r >> 2)) * EP;
                    const float v0 = __builtin_amdgcn_fmed3f(fmaf(acc0[r], esc, ebi), 0.f, 6.f);
                    const float v1 = __builtin_amdgcn_fmed3f(fmaf(acc1[r], esc, ebi), 0.f, 6.f);
                    e0[o] = ((emask >> r) & 1u) ? v0 : 0.f;
                    e0[32 * EP + o] = ((emask >> (16 + r)) & 1u) ? v1 : 0.f;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (cv) {
            f32x2 k0[9], k1[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const f32x4 kq = *reinterpret_cast<const f32x4*>(&Wd[t * HMAX + ch]);
                k0[t] = f32x2{kq.x, kq.y};
                k1[t] = f32x2{kq.z, kq.w};
            }
            const f32x4 dsc = *reinterpret_cast<const f32x4*>(&Bn[2 * HMAX + ch]);
            const f32x4 dbi = *reinterpret_cast<const f32x4*>(&Bn[3 * HMAX + ch]);
            const f32x2 sc0 = {dsc.x, dsc.y}, sc1 = {dsc.z, dsc.w}, bi0 = {dbi.x, dbi.y}, bi1 = {dbi.z, dbi.w};
#pragma unroll
            for (int q = 0; q < NR; ++q) {
                const int o = og + 8 * q;
                if (o < OTH * OTW) {
                    const int oy = o / OTW, ox = o - oy * OTW;
                    const int gy = oy0 + oy, gx = ox0 + ox;
                    const float* e = Ew + ((oy * S) * IW + ox * S) * EP + 4 * c4;
                    f32x2 s0 = {0.f, 0.f}, s1 = {0.f, 0.f};
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const f32x4 v = *reinterpret_cast<const f32x4*>(e + (ky * IW + kx) * EP);
                            s0 = __builtin_elementwise_fma(f32x2{v.x, v.y}, k0[ky * 3 + kx], s0);
                            s1 = __builtin_elementwise_fma(f32x2{v.z, v.w}, k1[ky * 3 + kx], s1);
                        }
                    if (gy < a.OH && gx < a.OW) {
                        const f32x2 r0 = __builtin_elementwise_fma(s0, sc0, bi0), r1 = __builtin_elementwise_fma(s1, sc1, bi1);
                        const f32x4 r = {__builtin_amdgcn_fmed3f(r0.x, 0.f, 6.f), __builtin_amdgcn_fmed3f(r0.y, 0.f, 6.f),
                                         __builtin_amdgcn_fmed3f(r1.x, 0.f, 6.f), __builtin_amdgcn_fmed3f(r1.y, 0.f, 6.f)};
                        *reinterpret_cast<f32x4*>(a.out + (((size_t)img * a.OH + gy) * a.OW + gx) * a.hid + ch) = r;
                    }
                }
            }
        }
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) bf[kk] = bnext[kk];
    }
}

// ---------------------------------------------------------------------------------------------------------
// Stem (3x3 stride-2 conv 3->32 + BN + ReLU6) -> block 1 (depthwise 3x3 + BN + ReLU6 -> project 1x1 32->16 + BN)
// in one persistent kernel: reads the 224^2 pixel-major frame, writes the 112^2 x 16 map; the two 112^2 x 32
// intermediates (1.6 MB per frame each, the 2nd and 3rd largest tensors of the network) stay in LDS.
//   tile = 8x8 outputs <- 10x10 stem outputs (halo) <- 21x21 input pixels
//   stem GEMM: the k index of the packed filter is (tap, channel-of-4), so a lane's 16-byte A fragment IS one input
//   pixel of the tile in LDS -- no im2col; K = 36 -> five groups of 8 (the 10th "tap" reads a zero vector), the
//   same group order as the conv engine's two k slices, so the result is bit-identical to the unfused stem.
// (Requesting the next tile's pixels during the current tile's compute was tried and was slower: 14.6 vs 13.9 ms
//  per 1024 frames for the whole glancer; three co-resident blocks already cover the load.)
template <int DUMMY>
__global__ __launch_bounds__(256) void mb_stem_b1_kernel(const MbStemArgs a) {
    constexpr int TW = 8, IW = 10, HP = 100, MP = 128, EP = 36;
    constexpr int XW = 21, XPIX = 21 * 21;
    __shared__ __attribute__((aligned(16))) float Xs[(XPIX + 1) * 4];   // + one zero pixel
    __shared__ __attribute__((aligned(16))) float Wst[32 * 44];
    __shared__ __attribute__((aligned(16))) float Es[MP * EP];
    __shared__ __attribute__((aligned(16))) float Ds[64 * EP];
    __shared__ __attribute__((aligned(16))) float Wps[32 * EP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    for (int idx = tid; idx < 32 * 10; idx += 256) {
        const int nr = idx / 10, c = idx - nr * 10;
        *reinterpret_cast<f32x4*>(&Wst[nr * 44 + 4 * c]) = c < 9 ? *reinterpret_cast<const f32x4*>(a.ws + nr * 36 + 4 * c) : zero4;
    }
    for (int idx = tid; idx < 32 * 8; idx += 256) {
        const int nr = idx >> 3, c = idx & 7;
        *reinterpret_cast<f32x4*>(&Wps[nr * EP + 4 * c]) = nr < 16 ? *reinterpret_cast<const f32x4*>(a.wp + nr * 32 + 4 * c) : zero4;
    }
    if (tid < 4) Xs[XPIX * 4 + tid] = 0.f;
    const int c4 = tid & 7, pg = tid >> 3;
    f32x4 k[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) k[t] = *reinterpret_cast<const f32x4*>(a.wd + t * 32 + 4 * c4);
    const f32x4 dsc = *reinterpret_cast<const f32x4*>(a.sd + 4 * c4);
    const f32x4 dbi = *reinterpret_cast<const f32x4*>(a.bd + 4 * c4);
    const int nl = lane & 31;
    const float ssc = a.ss[nl], sbi = a.bs[nl];
    const float psc = nl < 16 ? a.sp[nl] : 0.f, pbi = nl < 16 ? a.bp[nl] : 0.f;
    // A-fragment offsets of this lane's halo pixel: tap t = 2*kk + half -> input pixel (2*hy + kh, 2*hx + kw) of the tile
    const int p = 32 * wave + nl;
    const int pl = p < HP ? p : 0;
    const int base = ((2 * (pl / IW)) * XW + 2 * (pl % IW)) * 4;
    int aoff[5];
#pragma unroll
    for (int kk = 0; kk < 5; ++kk) {
        const int t = 2 * kk + half;
        aoff[kk] = t < 9 ? base + ((t / 3) * XW + t % 3) * 4 : XPIX * 4;
    }
    const int woff = nl * 44 + 4 * half;

    for (int tile = blockIdx.x; tile < a.total_tiles; tile += gridDim.x) {
        int q = tile;
        const int tx = q % a.tiles_x;
        q /= a.tiles_x;
        const int ty = q % a.tiles_y;
        const int img = q / a.tiles_y;
        const int oy0 = ty * 8, ox0 = tx * 8;
        const int hy0 = oy0 - 1, hx0 = ox0 - 1;
        const int iy0 = 2 * hy0 - 1, ix0 = 2 * hx0 - 1;
        __syncthreads();   // the previous tile's readers are done with Xs / Es / Ds
        for (int idx = tid; idx < XPIX; idx += 256) {
            const int r = idx / XW, c = idx - r * XW;
            const int iy = iy0 + r, ix = ix0 + c;
            const bool ok = (unsigned)iy < (unsigned)a.S && (unsigned)ix < (unsigned)a.S;
            *reinterpret_cast<f32x4*>(&Xs[idx * 4]) =
                *reinterpret_cast<const f32x4*>(ok ? a.x + (((size_t)img * a.S + iy) * a.S + ix) * 4 : a.zeros);
        }
        unsigned emask = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pp = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int hy = hy0 + pp / IW, hx = hx0 + pp % IW;
            if (pp < HP && (unsigned)hy < (unsigned)a.H1 && (unsigned)hx < (unsigned)a.H1) emask |= 1u << r;
        }
        __syncthreads();
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 5; ++kk) {
            const f32x4 af = *reinterpret_cast<const f32x4*>(&Xs[aoff[kk]]);
            const f32x4 bf = *reinterpret_cast<const f32x4*>(&Wst[woff + 8 * kk]);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pp = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * half;
            const float e = fminf(fmaxf(fmaf(acc[r], ssc, sbi) + 0.f, 0.f), 6.f);
            Es[pp * EP + nl] = ((emask >> r) & 1u) ? e : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int o = pg; o < 64; o += 32) {
            const int oy = o / TW, ox = o - oy * TW;
            f32x4 s = zero4;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(&Es[((oy + ky) * IW + ox + kx) * EP + 4 * c4]);
                    const f32x4 w = k[ky * 3 + kx];
                    s.x = fmaf(v.x, w.x, s.x);
                    s.y = fmaf(v.y, w.y, s.y);
                    s.z = fmaf(v.z, w.z, s.z);
                    s.w = fmaf(v.w, w.w, s.w);
                }
            f32x4 r;
            r.x = fminf(fmaxf(fmaf(s.x, dsc.x, dbi.x), 0.f), 6.f);
            r.y = fminf(fmaxf(fmaf(s.y, dsc.y, dbi.y), 0.f), 6.f);
            r.z = fminf(fmaxf(fmaf(s.z, dsc.z, dbi.z), 0.f), 6.f);
            r.w = fminf(fmaxf(fmaf(s.w, dsc.w, dbi.w), 0.f), 6.f);
            *reinterpret_cast<f32x4*>(&Ds[o * EP + 4 * c4]) = r;
        }
        __syncthreads();
        if (wave < 2) {   // project: 64 pixels x 16 channels, K = 32
            f32x16 pa;
#pragma unroll
            for (int r = 0; r < 16; ++r) pa[r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const f32x4 af = *reinterpret_cast<const f32x4*>(&Ds[(32 * wave + nl) * EP + 8 * kk + 4 * half]);
                const f32x4 bf = *reinterpret_cast<const f32x4*>(&Wps[nl * EP + 8 * kk + 4 * half]);
                pa = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, pa, 0, 0, 0);
                pa = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, pa, 0, 0, 0);
                pa = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, pa, 0, 0, 0);
                pa = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, pa, 0, 0, 0);
            }
            if (nl < 16) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const int gy = oy0 + o / TW, gx = ox0 + o % TW;
                    if (gy < a.H1 && gx < a.H1)
                        a.out[(((size_t)img * a.H1 + gy) * a.H1 + gx) * 16 + nl] = fmaf(pa[r], psc, pbi) + 0.f;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Stem -> block 1 with WAVE-PRIVATE tiles (the restructuring that paid for the expand -> depthwise blocks above): a wave owns
// 4 x 8 block-1 outputs <- 6 x 10 stem outputs (60 halo pixels = two MFMA row bands) <- 13 x 21 input pixels, and runs
// window -> stem GEMM -> BN/ReLU6 -> E -> depthwise taps -> D (the project conv's A operand, exactly one row band) -> project
// GEMM -> BN -> 16-byte stores on its own: no block-level barrier.  All filters live in registers (stem and project B
// fragments) or LDS (taps).  112 = 28 x 4 = 14 x 8: the tiles cover the map exactly.  Same products in the same order as
// mb_stem_b1_kernel and the unfused launches: bit-identical.
__global__ __launch_bounds__(256, 4) void mb_stem_b1_w_kernel(const MbStemArgs a) {
    constexpr int OTH = 4, OTW = 8;
    constexpr int HH = OTH + 2, HW = OTW + 2, HP = HH * HW;          // stem-output halo 6 x 10
    constexpr int XH = 2 * HH + 1, XW = 2 * HW + 1, XPIX = XH * XW;   // input window 13 x 21
    constexpr int EP = 16, DP = 36, SP = 20;          // E holds 16 channels at a time
    constexpr int XREG = (XPIX + 1) * 4 > 32 * DP ? (XPIX + 1) * 4 : 32 * DP;   // window (+ one zero pixel), later the D image
    __shared__ __attribute__((aligned(16))) float Xall[4][XREG];
    __shared__ __attribute__((aligned(16))) float Eall[4][64 * EP];               // E, later the output transposition slab
    __shared__ __attribute__((aligned(16))) float Wd[9 * 32 + 64];                // taps | depthwise scale | depthwise bias
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int idx = tid; idx < 9 * 32 + 64; idx += 256)
        Wd[idx] = idx < 288 ? a.wd[idx] : idx < 320 ? a.sd[idx - 288] : a.bd[idx - 320];
    __syncthreads();
    float* Xw = Xall[wave];
    float* Ew = Eall[wave];
    const int per_img = a.tiles_y * a.tiles_x;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= a.n * per_img) return;
    const int img = tile / per_img;
    const int rem = tile - img * per_img;
    const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    const int oy0 = ty * OTH, ox0 = tx * OTW;
    const int hy0 = oy0 - 1, hx0 = ox0 - 1;                 // stem-output coordinates of the halo's first pixel
    const int iy0 = 2 * hy0 - 1, ix0 = 2 * hx0 - 1;         // input coordinates of the window's first pixel
    const bool inner = hy0 >= 0 && hx0 >= 0 && hy0 + HH <= a.H1 && hx0 + HW <= a.H1;
    const int half = lane >> 5, nl = lane & 31;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // input window -> LDS (pixels outside the frame are the stem conv's zero padding).  All five pieces of a lane are requested
    // before the first is written: as `Xw[..] = ok ? *p : zero` per piece hipcc emitted load -> s_waitcnt vmcnt(0) -> ds_write five
    // times in a row -- five dependent memory round trips at the head of every tile (round 5, found in the ISA after the same pattern
    // had cost the small-batch conv form its prefetch).  Unconditional loads from a clamped address, zeroed afterwards.
    {
        constexpr int NP = (XPIX + 63) / 64;
        f32x4 wv[NP];
        bool wok[NP];
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int idx = lane + 64 * u;
            const int r = idx / XW, c = idx - r * XW;
            const int iy = iy0 + r, ix = ix0 + c;
            wok[u] = idx < XPIX && (unsigned)iy < (unsigned)a.S && (unsigned)ix < (unsigned)a.S;
            wv[u] = *reinterpret_cast<const f32x4*>(a.x + (wok[u] ? (((size_t)img * a.S + iy) * a.S + ix) * 4 : (size_t)0));
        }
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            const int idx = lane + 64 * u;
            if (idx < XPIX) *reinterpret_cast<f32x4*>(&Xw[idx * 4]) = wok[u] ? wv[u] : zero4;
        }
    }
    if (lane < 4) Xw[XPIX * 4 + lane] = 0.f;
    // filters: stem rows (k = tap * 4 + channel, 40 with the zero tenth tap) and project rows as B fragments, BN affines
    f32x4 bs[5], bp[4];
#pragma unroll
    for (int kk = 0; kk < 5; ++kk) {
        const int t = 2 * kk + half;
        bs[kk] = t < 9 ? *reinterpret_cast<const f32x4*>(a.ws + nl * 36 + 4 * t) : zero4;
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) bp[kk] = nl < 16 ? *reinterpret_cast<const f32x4*>(a.wp + nl * 32 + 8 * kk + 4 * half) : zero4;
    const float ssc = a.ss[nl], sbi = a.bs[nl];
    const float psc = nl < 16 ? a.sp[nl] : 0.f, pbi = nl < 16 ? a.bp[nl] : 0.f;
    __builtin_amdgcn_wave_barrier();

    // ---- stem GEMM: halo pixel p = 32 b + nl, tap t = 2 kk + half -> one input pixel (4 channels) of the window
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    {
        int base[2];
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int p = 32 * b + nl;
            const int pl = p < HP ? p : 0;
            base[b] = ((2 * (pl / HW)) * XW + 2 * (pl % HW)) * 4;
        }
#pragma unroll
        for (int kk = 0; kk < 5; ++kk) {
            const int t = 2 * kk + half;
            const int toff = ((t / 3) * XW + t % 3) * 4;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(&Xw[t < 9 ? base[0] + toff : XPIX * 4]);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(&Xw[t < 9 ? base[1] + toff : XPIX * 4]);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s4], bs[kk][s4], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s4], bs[kk][s4], acc1, 0, 0, 0);
            }
        }
    }
    // ---- BN + ReLU6 -> E -> depthwise 3x3 + BN + ReLU6 -> D, 16 channels at a time (E is then 60 x 16 floats: four blocks
    // fit a CU instead of three).  Stem outputs outside the map are the depthwise conv's zero padding.
    unsigned emask = 0xffffffffu;
    if (!inner) {
        emask = 0;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = 32 * b + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int hy = hy0 + p / HW, hx = hx0 + p % HW;
                if (p < HP && (unsigned)hy < (unsigned)a.H1 && (unsigned)hx < (unsigned)a.H1) emask |= 1u << (16 * b + r);
            }
    }
#pragma unroll
    for (int r = 0; r < 16; r += 2) {       // the accumulators become the E values in place
        const f32x2 sc2 = {ssc, ssc}, bi2 = {sbi, sbi};
        const f32x2 v0 = __builtin_elementwise_fma(f32x2{acc0[r], acc0[r + 1]}, sc2, bi2);
        const f32x2 v1 = __builtin_elementwise_fma(f32x2{acc1[r], acc1[r + 1]}, sc2, bi2);
        acc0[r] = __builtin_amdgcn_fmed3f(v0.x, 0.f, 6.f);
        acc0[r + 1] = __builtin_amdgcn_fmed3f(v0.y, 0.f, 6.f);
        acc1[r] = __builtin_amdgcn_fmed3f(v1.x, 0.f, 6.f);
        acc1[r + 1] = __builtin_amdgcn_fmed3f(v1.y, 0.f, 6.f);
    }
    if (!inner) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc0[r] = ((emask >> r) & 1u) ? acc0[r] : 0.f;
            acc1[r] = ((emask >> (16 + r)) & 1u) ? acc1[r] : 0.f;
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        __builtin_amdgcn_wave_barrier();        // the previous half's tap reads are issued
        if ((nl >> 4) == h) {
            float* e0 = Ew + (4 * half) * EP + (nl & 15);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = ((r & 3) + 8 * (r >> 2)) * EP;
                e0[o] = acc0[r];
                e0[32 * EP + o] = acc1[r];
            }
        }
        __builtin_amdgcn_wave_barrier();
        // item = (output pixel o = 16 q + lane / 4, channel group lane % 4 of this half)
        const int c4 = lane & 3, ol = lane >> 2;
        const int cg = 16 * h + 4 * c4;
        f32x2 k0[9], k1[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const f32x4 kq = *reinterpret_cast<const f32x4*>(&Wd[t * 32 + cg]);
            k0[t] = f32x2{kq.x, kq.y};
            k1[t] = f32x2{kq.z, kq.w};
        }
        const f32x4 dsc = *reinterpret_cast<const f32x4*>(&Wd[288 + cg]);
        const f32x4 dbi = *reinterpret_cast<const f32x4*>(&Wd[320 + cg]);
        const f32x2 sc0 = {dsc.x, dsc.y}, sc1 = {dsc.z, dsc.w}, bi0 = {dbi.x, dbi.y}, bi1 = {dbi.z, dbi.w};
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int o = 16 * q + ol;
            const int oy = o >> 3, ox = o & 7;
            const float* e = Ew + (oy * HW + ox) * EP + 4 * c4;
            f32x2 s0 = {0.f, 0.f}, s1 = {0.f, 0.f};
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(e + (ky * HW + kx) * EP);
                    s0 = __builtin_elementwise_fma(f32x2{v.x, v.y}, k0[ky * 3 + kx], s0);
                    s1 = __builtin_elementwise_fma(f32x2{v.z, v.w}, k1[ky * 3 + kx], s1);
                }
            const f32x2 r0 = __builtin_elementwise_fma(s0, sc0, bi0), r1 = __builtin_elementwise_fma(s1, sc1, bi1);
            const f32x4 r = {__builtin_amdgcn_fmed3f(r0.x, 0.f, 6.f), __builtin_amdgcn_fmed3f(r0.y, 0.f, 6.f),
                             __builtin_amdgcn_fmed3f(r1.x, 0.f, 6.f), __builtin_amdgcn_fmed3f(r1.y, 0.f, 6.f)};
            *reinterpret_cast<f32x4*>(&Xw[o * DP + cg]) = r;      // (the window is dead: D takes its place)
        }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- project 1x1 (32 -> 16) + BN: one row band of 32 output pixels, K = 32
    f32x16 pa;
#pragma unroll
    for (int r = 0; r < 16; ++r) pa[r] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
        const f32x4 af = *reinterpret_cast<const f32x4*>(&Xw[nl * DP + 8 * kk + 4 * half]);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) pa = __builtin_amdgcn_mfma_f32_32x32x2f32(af[s4], bp[kk][s4], pa, 0, 0, 0);
    }
    if (nl < 16) {
#pragma unroll
        for (int r = 0; r < 16; ++r) Ew[((r & 3) + 8 * (r >> 2) + 4 * half) * SP + nl] = fmaf(pa[r], psc, pbi) + 0.f;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int i = lane + 64 * j;
        const int px = i >> 2, c = i & 3;
        const int gy = oy0 + (px >> 3), gx = ox0 + (px & 7);
        if (gy < a.H1 && gx < a.H1)
            *reinterpret_cast<f32x4*>(a.out + (((size_t)img * a.H1 + gy) * a.H1 + gx) * 16 + 4 * c) =
                *reinterpret_cast<const f32x4*>(&Ew[px * SP + 4 * c]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// A whole stride-1 inverted-residual block per wave: expand 1x1 -> BN/ReLU6 -> depthwise 3x3 -> BN/ReLU6 -> project 1x1 -> BN
// (+ identity), wave-private tiles of 4 x 8 outputs <- 6 x 10 halo pixels.  The 6x-expanded maps never exist in HBM: a block
// reads cin and writes cout channels per pixel (b3: 24 + 24 (+24 identity) instead of 24 + 144 + 144 + 24 (+24)).  Per chunk
// of 32 hidden channels: the expand products (two row bands), E through LDS 16 channels at a time, the taps on the VALU into
// D = one k slice of the project conv's A operand, and 16 project MFMAs into accumulators that live across the chunks.
// The project conv's k order is the conv engine's (slices of 32, a partial last slice zero-filled): bit-identical to the
// three separate launches.
template <int CIN>
__global__ __launch_bounds__(256, 3) void mb_block_w_kernel(const MbFuseArgs a) {
    constexpr int OTH = 4, OTW = 8, HH = OTH + 2, HW = OTW + 2, HP = HH * HW;
    constexpr int EP = 16, DP = 36, KK = CIN / 8, HMAX = 192, SP = 36;
    __shared__ __attribute__((aligned(16))) float Eall[4][64 * EP];
    __shared__ __attribute__((aligned(16))) float Dall[4][32 * DP];     // D chunk; at the end the output transposition slab
    __shared__ __attribute__((aligned(16))) float Wd[9 * HMAX];
    __shared__ __attribute__((aligned(16))) float Bn[4 * HMAX];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int idx = tid; idx < 9 * HMAX; idx += 256) {
        const int t = idx / HMAX, c = idx - t * HMAX;
        Wd[idx] = c < a.hid ? a.wd[(size_t)t * a.hid + c] : 0.f;
    }
    for (int idx = tid; idx < HMAX; idx += 256) {
        const bool v = idx < a.hid;
        Bn[idx] = v ? a.se[idx] : 0.f;
        Bn[HMAX + idx] = v ? a.be[idx] : 0.f;
        Bn[2 * HMAX + idx] = v ? a.sd[idx] : 0.f;
        Bn[3 * HMAX + idx] = v ? a.bd[idx] : 0.f;
    }
    __syncthreads();
    float* Ew = Eall[wave];
    float* Dw = Dall[wave];
    const int per_img = a.tiles_y * a.tiles_x;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= a.n * per_img) return;
    const int img = tile / per_img;
    const int rem = tile - img * per_img;
    const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
    const int oy0 = ty * OTH, ox0 = tx * OTW;
    const int iy0 = oy0 - 1, ix0 = ox0 - 1;
    const bool inner = iy0 >= 0 && ix0 >= 0 && iy0 + HH <= a.H && ix0 + HW <= a.W;
    const int half = lane >> 5, nl = lane & 31;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    f32x4 af[2][KK];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int p = 32 * b + nl;
        const int iy = iy0 + p / HW, ix = ix0 + p % HW;
        const bool ok = p < HP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const float* src = a.x + (((size_t)img * a.H + (ok ? iy : 0)) * a.W + (ok ? ix : 0)) * CIN + 4 * half;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) af[b][kk] = ok ? *reinterpret_cast<const f32x4*>(src + 8 * kk) : zero4;
    }
    unsigned emask = 0xffffffffu;
    if (!inner) {
        emask = 0;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = 32 * b + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int iy = iy0 + p / HW, ix = ix0 + p % HW;
                if (p < HP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) emask |= 1u << (16 * b + r);
            }
    }
    f32x16 pa;
#pragma unroll
    for (int r = 0; r < 16; ++r) pa[r] = 0.f;

    for (int ch0 = 0; ch0 < a.hid; ch0 += 32) {
        // filter rows of this chunk: expand (B fragments, row = hidden channel) and project (row = output channel, k = hidden)
        const int nch = ch0 + nl;
        f32x4 bf[KK], bp[4];
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
            bf[kk] = nch < a.hid ? *reinterpret_cast<const f32x4*>(a.we + (size_t)nch * CIN + 8 * kk + 4 * half) : zero4;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int k = ch0 + 8 * kk + 4 * half;
            bp[kk] = (nl < a.cout && k < a.hid) ? *reinterpret_cast<const f32x4*>(a.wp + (size_t)nl * a.hid + k) : zero4;
        }
        const float esc = Bn[ch0 + nl], ebi = Bn[HMAX + ch0 + nl];
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0][kk][s4], bf[kk][s4], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1][kk][s4], bf[kk][s4], acc1, 0, 0, 0);
            }
        {
            const f32x2 sc2 = {esc, esc}, bi2 = {ebi, ebi};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {       // the accumulators become the E values in place
                const f32x2 v0 = __builtin_elementwise_fma(f32x2{acc0[r], acc0[r + 1]}, sc2, bi2);
                const f32x2 v1 = __builtin_elementwise_fma(f32x2{acc1[r], acc1[r + 1]}, sc2, bi2);
                acc0[r] = __builtin_amdgcn_fmed3f(v0.x, 0.f, 6.f);
                acc0[r + 1] = __builtin_amdgcn_fmed3f(v0.y, 0.f, 6.f);
                acc1[r] = __builtin_amdgcn_fmed3f(v1.x, 0.f, 6.f);
                acc1[r + 1] = __builtin_amdgcn_fmed3f(v1.y, 0.f, 6.f);
            }
            if (!inner) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    acc0[r] = ((emask >> r) & 1u) ? acc0[r] : 0.f;
                    acc1[r] = ((emask >> (16 + r)) & 1u) ? acc1[r] : 0.f;
                }
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            __builtin_amdgcn_wave_barrier();        // earlier reads of E (and, for h = 0, of D) are issued
            if ((nl >> 4) == h) {
                float* e0 = Ew + (4 * half) * EP + (nl & 15);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = ((r & 3) + 8 * (r >> 2)) * EP;
                    e0[o] = acc0[r];
                    e0[32 * EP + o] = acc1[r];
                }
            }
            __builtin_amdgcn_wave_barrier();
            const int c4 = lane & 3, ol = lane >> 2;
            const int cg = 16 * h + 4 * c4;           // channel group inside the chunk
            f32x2 k0[9], k1[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const f32x4 kq = *reinterpret_cast<const f32x4*>(&Wd[t * HMAX + ch0 + cg]);
                k0[t] = f32x2{kq.x, kq.y};
                k1[t] = f32x2{kq.z, kq.w};
            }
            const f32x4 dsc = *reinterpret_cast<const f32x4*>(&Bn[2 * HMAX + ch0 + cg]);
            const f32x4 dbi = *reinterpret_cast<const f32x4*>(&Bn[3 * HMAX + ch0 + cg]);
            const f32x2 sc0 = {dsc.x, dsc.y}, sc1 = {dsc.z, dsc.w}, bi0 = {dbi.x, dbi.y}, bi1 = {dbi.z, dbi.w};
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int o = 16 * q + ol;
                const int oy = o >> 3, ox = o & 7;
                const float* e = Ew + (oy * HW + ox) * EP + 4 * c4;
                f32x2 s0 = {0.f, 0.f}, s1 = {0.f, 0.f};
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(e + (ky * HW + kx) * EP);
                        s0 = __builtin_elementwise_fma(f32x2{v.x, v.y}, k0[ky * 3 + kx], s0);
                        s1 = __builtin_elementwise_fma(f32x2{v.z, v.w}, k1[ky * 3 + kx], s1);
                    }
                const f32x2 r0 = __builtin_elementwise_fma(s0, sc0, bi0), r1 = __builtin_elementwise_fma(s1, sc1, bi1);
                const f32x4 r = {__builtin_amdgcn_fmed3f(r0.x, 0.f, 6.f), __builtin_amdgcn_fmed3f(r0.y, 0.f, 6.f),
                                 __builtin_amdgcn_fmed3f(r1.x, 0.f, 6.f), __builtin_amdgcn_fmed3f(r1.y, 0.f, 6.f)};
                *reinterpret_cast<f32x4*>(&Dw[o * DP + cg]) = r;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // project: pa += D[32 px x 32] * Wp[:, ch0 .. ch0 + 32]^T  (channels past `hid` are zeros on both sides)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const f32x4 ad = *reinterpret_cast<const f32x4*>(&Dw[nl * DP + 8 * kk + 4 * half]);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) pa = __builtin_amdgcn_mfma_f32_32x32x2f32(ad[s4], bp[kk][s4], pa, 0, 0, 0);
        }
    }
    // ---- project BN (+ identity): through the slab so that stores (and identity loads) are 16 bytes per lane
    __builtin_amdgcn_wave_barrier();
    {
        const float psc = nl < a.cout ? a.sp[nl] : 0.f, pbi = nl < a.cout ? a.bp[nl] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) Dw[((r & 3) + 8 * (r >> 2) + 4 * half) * SP + nl] = fmaf(pa[r], psc, pbi);
    }
    __builtin_amdgcn_wave_barrier();
    const int cq = a.cout >> 2;             // 16-byte pieces per output pixel (cout % 4 == 0)
    for (int i = lane; i < 32 * cq; i += 64) {
        const int px = i / cq, c = i - px * cq;
        const int gy = oy0 + (px >> 3), gx = ox0 + (px & 7);
        if (gy < a.OH && gx < a.OW) {
            const size_t g = (((size_t)img * a.OH + gy) * a.OW + gx) * a.cout + 4 * c;
            f32x4 v = *reinterpret_cast<const f32x4*>(&Dw[px * SP + 4 * c]);
            if (a.res) {
                const f32x4 rr = *reinterpret_cast<const f32x4*>(a.res + g);
                v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
            }
            *reinterpret_cast<f32x4*>(a.out2 + g) = v;
        }
    }
}

}  // namespace

bool adaf_mb_expand_dw_ok(int cin, int hid, int hw) { return cin % 8 == 0 && cin <= 32 && hid % 4 == 0 && hw >= 28; }

template <int S, int CIN>
static void launch_expand_dw_w(MbFuseArgs a, hipStream_t s) {
    constexpr int OTH = S == 1 ? 6 : 3, OTW = S == 1 ? 6 : 4;
    a.tiles_x = (a.OW + OTW - 1) / OTW;
    a.tiles_y = (a.OH + OTH - 1) / OTH;
    const long long tiles = (long long)a.n * a.tiles_x * a.tiles_y;
    hipLaunchKernelGGL((mb_expand_dw_w_kernel<S, CIN>), dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, s, a);
}

// (the wave-private kernels were an option, "mb_wave", while they were measured against the block-cooperative ones, rounds 2-5; the latter remain
//  as the fallback for hidden widths beyond the LDS-resident tap table)
static constexpr bool mb_wave_enabled() { return true; }

void adaf_launch_mb_expand_dw(MbFuseArgs a, int stride, hipStream_t s) {
    if (mb_wave_enabled() && a.hid <= 192) {
        if (stride == 1 && a.cin == 16) return launch_expand_dw_w<1, 16>(a, s);
        if (stride == 2 && a.cin == 16) return launch_expand_dw_w<2, 16>(a, s);
        if (stride == 1 && a.cin == 24) return launch_expand_dw_w<1, 24>(a, s);
        if (stride == 2 && a.cin == 24) return launch_expand_dw_w<2, 24>(a, s);
        if (stride == 1 && a.cin == 32) return launch_expand_dw_w<1, 32>(a, s);
        if (stride == 2 && a.cin == 32) return launch_expand_dw_w<2, 32>(a, s);
    }
    const int th = stride == 1 ? 8 : 3, tw = 8;
    a.tiles_x = (a.OW + tw - 1) / tw;
    a.tiles_y = (a.OH + th - 1) / th;
    const unsigned blocks = (unsigned)a.n * a.tiles_x * a.tiles_y;
    const size_t smem = sizeof(float) * ((size_t)160 * (a.cin + 4) + 128 * 36);
    if (stride == 1) hipLaunchKernelGGL((mb_expand_dw_kernel<1>), dim3(blocks), dim3(256), smem, s, a);
    else hipLaunchKernelGGL((mb_expand_dw_kernel<2>), dim3(blocks), dim3(256), smem, s, a);
}

// whole-block kernel: stride-1 blocks with up to 32 output channels (b3, b5, b6 of MobileNetV2 1.0)
bool adaf_mb_block_ok(int cin, int hid, int cout, int stride, int hw) {
    if (mb_wave_enabled() && stride == 2) return adaf_mb_block_strip_ok(cin, hid, cout, stride, hw, hw);     // (stride 2: the strip form only)
    return mb_wave_enabled() && stride == 1 && (cin == 16 || cin == 24 || cin == 32) && hid % 4 == 0 && hid <= 192 && cout % 4 == 0 &&
           cout <= 32 && hw >= 28;
}

void adaf_launch_mb_block(MbFuseArgs a, hipStream_t s) {
    if (adaf_mb_block_strip_ok(a.cin, a.hid, a.cout, a.OH == a.H ? 1 : 2, a.H, a.W)) return adaf_launch_mb_block_strip(a, s);
    a.tiles_x = (a.OW + 7) / 8;
    a.tiles_y = (a.OH + 3) / 4;
    const long long tiles = (long long)a.n * a.tiles_x * a.tiles_y;
    const dim3 grid((unsigned)((tiles + 3) / 4)), block(256);
    if (a.cin == 16) hipLaunchKernelGGL((mb_block_w_kernel<16>), grid, block, 0, s, a);
    else if (a.cin == 24) hipLaunchKernelGGL((mb_block_w_kernel<24>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((mb_block_w_kernel<32>), grid, block, 0, s, a);
}

void adaf_launch_mb_stem_b1(MbStemArgs a, int cus, hipStream_t s) {
    if (mb_wave_enabled() && adaf_mb_stem_b1_strip_ok(a.S, a.H1)) return adaf_launch_mb_stem_b1_strip(a, s);
    if (mb_wave_enabled()) {
        a.tiles_x = (a.H1 + 7) / 8;
        a.tiles_y = (a.H1 + 3) / 4;
        a.total_tiles = a.n * a.tiles_x * a.tiles_y;
        hipLaunchKernelGGL(mb_stem_b1_w_kernel, dim3((unsigned)((a.total_tiles + 3) / 4)), dim3(256), 0, s, a);
        return;
    }
    a.tiles_x = (a.H1 + 7) / 8;
    a.tiles_y = a.tiles_x;
    a.total_tiles = a.n * a.tiles_x * a.tiles_y;
    const int blocks = a.total_tiles < cus * 3 ? a.total_tiles : cus * 3;
    hipLaunchKernelGGL((mb_stem_b1_kernel<0>), dim3(blocks), dim3(256), 0, s, a);
}
