// Strip-walking front of the MobileNetV2 glancer (round 6; ACT/models/mobilenet.py:42-68,71-148).
//
// The wave-private tile kernels of mbconv.hip sit at the sum of their matrix-pipe and vector-pipe time -- on gfx950 an MFMA and a
// VALU instruction never issue side by side on a SIMD, whatever the operand type (tools/exp/mfma_valu_overlap.hip) -- and both terms
// were mostly overhead: 1.9x of the expand / stem products were halo recomputation, half of the project conv's 32-column tile was
// padding, and ~70 % of the vector instructions were tile index arithmetic, bounds selects and scalar LDS traffic (614 VALU
// instructions per 32 outputs of the stem + block-1 kernel, ~175 of them arithmetic).
//
// Here a WAVE owns a column strip of a frame (14 output columns <- 16 columns of the intermediate map) and walks it top to bottom:
//   * every step computes TWO NEW ROWS of the intermediate map (2 x 16 pixels = one 32-row MFMA band: no vertical halo, 16 / 14
//     horizontally), its input pixels loaded straight into MFMA A fragments with buffer loads -- rows outside the frame come back
//     as zeros from the buffer's range check, so the conv's zero padding costs no select;
//   * the last four rows of the intermediate map live in a wave-private LDS ring; the depthwise taps read them once per
//     (row, column offset) for the two output rows of the step;
//   * a lane of the depthwise phase is (output column n = lane & 15, k group g = lane >> 4) and owns exactly the 8 channels that
//     lane supplies as B operand of the project conv's v_mfma_f32_16x16x4_f32 chain (rows = the 16 output channels, no padding),
//     so the depthwise outputs go from registers into the matrix pipe, and the accumulator a lane gets back is one pixel's output
//     channels 4g .. 4g+3: one 16-byte store.  No LDS, no shuffle after the taps.
//   * all addresses are per-wave constants plus a row stride: the loop has no index arithmetic.
// Same products in the same order per output as mb_stem_b1_w_kernel and the unfused launches (the 16x16x4 chain takes its k lanes
// in the order DESIGN 3.8 established: {0,4,1,5} then {2,6,3,7} of every 8-k group = the 32x32x2 chain): bit-identical.
#include <type_traits>

#include "adaf_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Ablation build (tools/exp/build_mbs_abl.sh, never the product library): MBS_ABL bits switch parts of the kernels OFF to see what each costs --
// 1 the expand / stem MFMA chains, 2 the project MFMA chains, 4 the depthwise tap FMAs, 8 the ring reads of the taps, 16 the pixel loads.
#ifndef MBS_ABL
#define MBS_ABL 0
#endif

namespace {
constexpr int ABL = MBS_ABL;

// k offset (inside a 32-channel slice) that lane group g supplies as its q-th product of the 16x16x4 chain
__device__ __forceinline__ int kq(int g, int q) { return 8 * (q >> 1) + (g >> 1) + 4 * (g & 1) + 2 * (q & 1); }
// ... and its inverse: position 8 g + q of channel c in the [g][q] order the depthwise lanes read
__device__ __forceinline__ int pos_of(int c) {
    const int e = c & 7, second = (e >> 1) & 1, f = e - 2 * second;
    return 8 * (((f & 1) << 1) | (f >> 2)) + 2 * (c >> 3) + second;
}

__device__ __forceinline__ f32x4 ringr(const float* p) {      // a 16-byte read of the expanded-row ring
    if (ABL & 8) return f32x4{1.f, 2.f, 3.f, 4.f};
    return *reinterpret_cast<const f32x4*>(p);
}

__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t r, int byte_off) {
    if (ABL & 16) return f32x4{1.f, 0.5f, 0.25f, 0.f};
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
}

// acc += a * b on two packed fp32 lanes.  As inline asm because clang 22 UNPACKS a v_pk_fma_f32 that follows an MFMA into two v_fma_f32 ("to
// co-issue with the MFMA"): on gfx950 nothing co-issues (tools/exp/mfma_valu_overlap.hip: MFMA + VALU time is additive, also inside one wave), so
// the pair costs 9.6 instead of 6.6 cycles -- a third of the tap FMAs of these kernels were unpacked.  Same arithmetic, one rounding per lane.
__device__ __forceinline__ void pkfma(f32x2& acc, f32x2 a, f32x2 b) {
#if MBS_ABL & 4
    acc.x += a.x * 0.f + b.x * 0.f;     // (ablation: keeps the operands alive, costs one add)
#else
    asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#endif
}

constexpr int SW_OW = 14;                    // output columns of a strip
constexpr int SW_EP = 36;                    // floats per pixel of the ring (32 channels + 4: conflict-free 16-byte reads)
constexpr int SW_ROWF = 18 * SW_EP;          // 16 columns + 2 the idle lanes n = 14, 15 read past
constexpr int SW_RING = 4 * SW_ROWF;

// Stem (3x3 / 2, 3 -> 32, BN, ReLU6) -> block 1 (depthwise 3x3, BN, ReLU6 -> project 32 -> 16, BN).  Requires S == 2 * H1, H1 % 14 == 0.
__global__ __launch_bounds__(256, 2) void mb_stem_b1_s_kernel(const MbStemArgs a) {
    __shared__ __attribute__((aligned(16))) float Eall[4][SW_RING];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* Ew = Eall[wave];
    for (int i = lane; i < SW_RING; i += 64) Ew[i] = 0.f;     // (columns 16, 17 are never written: keep what the idle lanes read finite)
    const int nstrips = a.tiles_x;
    const int strip = blockIdx.x * 4 + wave;
    if (strip >= a.n * nstrips) return;
    const int img = strip / nstrips, sx = strip - img * nstrips;
    const int ox0 = sx * SW_OW;
    const bool left = sx == 0, right = sx == nstrips - 1;
    const int S = a.S, H1 = a.H1;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // ---- stem GEMM roles: A row p = lane & 31 = (er = p >> 4, ec = p & 15), k = (tap t = 2 kk + half, channel of 4)
    const int half = lane >> 5, nl = lane & 31;
    const int er = nl >> 4, ec = nl & 15;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + (size_t)img * S * S * 4), 0, S * S * 16, 0x00020000);
    int voff[5];
#pragma unroll
    for (int kk = 0; kk < 5; ++kk) {
        const int t = 2 * kk + half;
        // E row 2 s - 1 + er at step s: input row 2 (2 s - 1 + er) - 1 + t / 3, input column 2 (ox0 - 1 + ec) - 1 + t % 3
        voff[kk] = ((2 * er - 3 + t / 3) * S + 2 * (ox0 + ec) - 3 + t % 3) * 16;
    }
    const bool zl0 = left && ec == 1 && half == 0, zl1 = left && ec == 1 && half == 1;      // taps 0, 6 (half 0) / 3 (half 1) at input column -1
    f32x4 bs[5];
#pragma unroll
    for (int kk = 0; kk < 5; ++kk) {
        const int t = 2 * kk + half;
        bs[kk] = t < 9 ? *reinterpret_cast<const f32x4*>(a.ws + nl * 36 + 4 * t) : zero4;
    }
    const float ssc = a.ss[nl], sbi = a.bs[nl];
    float* ewr = Ew + (4 * half) * SW_EP + pos_of(nl);      // + slot * ROWF + ((r & 3) + 8 ((r >> 2) & 1)) * EP per accumulator row r

    // ---- depthwise / project roles: (output column n, k group g)
    const int n = lane & 15, g = lane >> 4;
    f32x2 tap[9][4];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int q = 0; q < 8; q += 2) tap[t][q >> 1] = f32x2{a.wd[t * 32 + kq(g, q)], a.wd[t * 32 + kq(g, q + 1)]};
    f32x2 dsc[4], dbi[4];
    float wp[8];
#pragma unroll
    for (int q = 0; q < 8; q += 2) {
        dsc[q >> 1] = f32x2{a.sd[kq(g, q)], a.sd[kq(g, q + 1)]};
        dbi[q >> 1] = f32x2{a.bd[kq(g, q)], a.bd[kq(g, q + 1)]};
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) wp[q] = a.wp[n * 32 + kq(g, q)];
    const f32x4 psc = *reinterpret_cast<const f32x4*>(a.sp + 4 * g), pbi = *reinterpret_cast<const f32x4*>(a.bp + 4 * g);
    const float* erd = Ew + n * SW_EP + 8 * g;
    float* const orow0 = a.out + ((size_t)img * H1 * H1 + ox0 + n) * 16 + 4 * g;      // output row 0 of this lane's column
    const bool ostore = n < SW_OW;

    f32x4 af[5];
    auto fetch = [&]() {
#pragma unroll
        for (int kk = 0; kk < 5; ++kk) af[kk] = bload(rsrc, voff[kk]);
#pragma unroll
        for (int kk = 0; kk < 5; ++kk) voff[kk] += 4 * S * 16;
    };
    fetch();
    __builtin_amdgcn_wave_barrier();

    const int nsteps = H1 / 2 + 1;
    // PAR = s & 1: the E rows 2 s - 1, 2 s of step s live in ring slots (3, 0) for even s and (1, 2) for odd s
    auto step = [&](auto PAR, int s) {
        constexpr int par = decltype(PAR)::value;
        constexpr int SL0 = par ? 1 : 3, SL1 = par ? 2 : 0;
        if (left) {       // (wave-uniform: the empty asm keeps it a branch -- if-converted it is 12 selects in every step of every strip)
            asm volatile("");
            af[0] = zl0 ? zero4 : af[0];
            af[1] = zl1 ? zero4 : af[1];
            af[3] = zl0 ? zero4 : af[3];
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 5; ++kk)
#pragma unroll
            for (int s4 = 0; s4 < 3; ++s4)      // (the fourth "channel" of a pixel-major frame has zero filter entries: its five products add +-0 to an
                                                //  accumulator that is +0 or non-zero at that point -- the same bits without them, a quarter of the chain gone)
                if (!(ABL & 1)) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk][s4], bs[kk][s4], acc, 0, 0, 0);
        // BN + ReLU6; stem outputs outside the map are the depthwise conv's zero padding
        {
            const f32x2 sc2 = {ssc, ssc}, bi2 = {sbi, sbi};
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const f32x2 v = __builtin_elementwise_fma(f32x2{acc[r], acc[r + 1]}, sc2, bi2);
                acc[r] = __builtin_amdgcn_fmed3f(v.x, 0.f, 6.f);
                acc[r + 1] = __builtin_amdgcn_fmed3f(v.y, 0.f, 6.f);
            }
        }
        // The next step's pixels are requested only HERE: a returning load is not interlocked against an MFMA that still has to read the
        // register it lands in (DESIGN / LABNOTES "a load may not land in an MFMA's registers", round 5) -- the VALU reads of the chain's
        // result above are, so past this point every MFMA that reads af[] has retired.
        __builtin_amdgcn_sched_barrier(0);
        if (s + 1 < nsteps) fetch();
        __builtin_amdgcn_sched_barrier(0);
        if (s == 0) {
            asm volatile("");
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r] = 0.f;           // E row -1
        }
        if (s == nsteps - 1) {
            asm volatile("");
#pragma unroll
            for (int r = 8; r < 16; ++r) acc[r] = 0.f;          // E row H1
        }
        if (left) { asm volatile(""); if (half == 0) { acc[0] = 0.f; acc[8] = 0.f; } }     // E column -1   (ec = 0: rows r = 0, 8 of half 0)
        if (right) { asm volatile(""); if (half == 1) { acc[7] = 0.f; acc[15] = 0.f; } }   // E column H1   (ec = 15: rows r = 7, 15 of half 1)
        __builtin_amdgcn_wave_barrier();        // the previous step's tap reads are issued (same-wave LDS traffic runs in order)
#pragma unroll
        for (int r = 0; r < 16; ++r) ewr[(r < 8 ? SL0 : SL1) * SW_ROWF + ((r & 3) + 8 * ((r >> 2) & 1)) * SW_EP] = acc[r];
        __builtin_amdgcn_wave_barrier();
        if (s > 0) {
            // ---- depthwise 3x3 of output rows y0 = 2 s - 2, y1 = 2 s - 1 from E rows 2 s - 3 .. 2 s
            constexpr int R0 = par ? 3 : 1, R1 = par ? 0 : 2, R2 = SL0, R3 = SL1;
            constexpr int slot[4] = {R0, R1, R2, R3};
            f32x2 s0[4], s1[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { s0[i] = f32x2{0.f, 0.f}; s1[i] = f32x2{0.f, 0.f}; }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const float* p = erd + slot[e] * SW_ROWF + kx * SW_EP;
                    const f32x4 va = ringr(p), vb = ringr(p + 4);
                    const f32x2 v[4] = {{va.x, va.y}, {va.z, va.w}, {vb.x, vb.y}, {vb.z, vb.w}};
                    if (e < 3) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) pkfma(s0[i], v[i], tap[e * 3 + kx][i]);
                    }
                    if (e > 0) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) pkfma(s1[i], v[i], tap[(e - 1) * 3 + kx][i]);
                    }
                }
            float d0[8], d1[8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x2 r0 = __builtin_elementwise_fma(s0[i], dsc[i], dbi[i]), r1 = __builtin_elementwise_fma(s1[i], dsc[i], dbi[i]);
                d0[2 * i] = __builtin_amdgcn_fmed3f(r0.x, 0.f, 6.f); d0[2 * i + 1] = __builtin_amdgcn_fmed3f(r0.y, 0.f, 6.f);
                d1[2 * i] = __builtin_amdgcn_fmed3f(r1.x, 0.f, 6.f); d1[2 * i + 1] = __builtin_amdgcn_fmed3f(r1.y, 0.f, 6.f);
            }
            // ---- project 32 -> 16 + BN: C[cout][pixel] = sum_k Wp[cout][k] D[pixel][k]; this lane gets couts 4 g .. 4 g + 3 of pixel n
            f32x4 p0 = zero4, p1 = zero4;
#pragma unroll
            for (int q = 0; q < 8 && !(ABL & 2); ++q) {
                p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[q], d0[q], p0, 0, 0, 0);
                p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[q], d1[q], p1, 0, 0, 0);
            }
            if (ostore) {
                f32x4 o0, o1;
#pragma unroll
                for (int i = 0; i < 4; ++i) { o0[i] = fmaf(p0[i], psc[i], pbi[i]) + 0.f; o1[i] = fmaf(p1[i], psc[i], pbi[i]) + 0.f; }
                float* optr = orow0 + (size_t)(2 * s - 2) * H1 * 16;
                *reinterpret_cast<f32x4*>(optr) = o0;
                *reinterpret_cast<f32x4*>(optr + (size_t)H1 * 16) = o1;
            }
        }
    };
    for (int s = 0; s < nsteps; s += 2) {
        step(std::integral_constant<int, 0>{}, s);
        if (s + 1 < nsteps) step(std::integral_constant<int, 1>{}, s + 1);
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// A whole stride-1 inverted-residual block (expand 1x1 -> BN/ReLU6 -> depthwise 3x3 -> BN/ReLU6 -> project 1x1 -> BN [+ identity]) as
// strip segments: a wave owns 14 output columns x SEG output rows of a frame.  The hidden channels are walked in chunks of 32 in the
// OUTER loop -- a chunk's filter rows, taps and affines are fetched once per segment -- and the segment's rows in steps of two
// expanded rows in the inner loop (SEG / 2 + 1 steps: one row of vertical halo above and below the segment is recomputed, 10 / 8),
// exactly as the stem kernel above walks a frame: expanded rows into the four-row ring, the taps of the two output rows they complete
// straight into the B operand of the project conv's 16x16x4 chain.  The project accumulators of all SEG rows stay in registers across
// the chunks (SEG / 2 row pairs x 2 rows x two 16-channel blocks), so the project conv sums its k slices in the conv engine's order
// (slices of 32, a partial last slice zero-filled): bit-identical to mb_block_w_kernel and to the three separate launches.
// The depthwise taps and affines of every chunk sit in LDS in the order the lanes consume them ([chunk][k group][tap][8]): a lane's
// 8 channels x 9 taps would be 72 registers next to 64 of accumulators.  The input needs no padding select at all: expanded pixels
// outside the map are forced to zero (the depthwise conv pads the EXPANDED map), rows outside the frame load as zeros anyway.
constexpr int SW_SEG = 8;                    // output rows of a segment
constexpr int SW_TAPF = 11 * 8;              // floats per (chunk, k group): 9 taps | depthwise scale | depthwise bias, 8 channels each
constexpr int SW_MAXCH = 6;                  // chunks (hid <= 192)

template <int CIN>
__global__ __launch_bounds__(256, 2) void mb_block_s_kernel(const MbFuseArgs a) {
    constexpr int KK = CIN / 8, NST = SW_SEG / 2 + 1;
    __shared__ __attribute__((aligned(16))) float Eall[4][SW_RING];
    __shared__ __attribute__((aligned(16))) float Tall[SW_MAXCH * 4 * SW_TAPF];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hid = a.hid, nchunks = (hid + 31) >> 5;
    for (int idx = tid; idx < nchunks * 4 * SW_TAPF; idx += 256) {
        const int q = idx & 7, t = (idx >> 3) % 11, cg = idx / SW_TAPF;
        const int ch = 32 * (cg >> 2) + kq(cg & 3, q);
        float v = 0.f;
        if (ch < hid) v = t < 9 ? a.wd[(size_t)t * hid + ch] : t == 9 ? a.sd[ch] : a.bd[ch];
        Tall[idx] = v;
    }
    float* Ew = Eall[wave];
    for (int i = lane; i < SW_RING; i += 64) Ew[i] = 0.f;
    __syncthreads();
    const int nstrips = a.tiles_x, nsegs = a.tiles_y;
    const int item = blockIdx.x * 4 + wave;
    if (item >= a.n * nstrips * nsegs) return;
    const int img = item / (nstrips * nsegs), rem = item - img * (nstrips * nsegs);
    const int seg = rem / nstrips, sx = rem - seg * nstrips;        // (the strips of a row segment are neighbours: they share input lines)
    const int ox0 = sx * SW_OW, y0 = seg * SW_SEG;
    const int H = a.H, W = a.W;
    const int rows = H - y0 < SW_SEG ? H - y0 : SW_SEG;             // output rows of this segment (even)
    const int nsteps = rows / 2 + 1;
    const bool left = sx == 0, right = sx == nstrips - 1, top = seg == 0, bottom = y0 + rows == H;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // ---- expand GEMM roles: A row p = lane & 31 = (er = p >> 4, ec = p & 15) is input pixel (y0 - 1 + 2 s + er, ox0 - 1 + ec), k = 8 kk + 4 half ..
    const int half = lane >> 5, nl = lane & 31;
    const int er = nl >> 4, ec = nl & 15;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + (size_t)img * H * W * CIN), 0, H * W * CIN * 4, 0x00020000);
    const int voff0 = (((y0 - 1 + er) * W + ox0 - 1 + ec) * CIN + 4 * half) * 4;
    const int rstep = 2 * W * CIN * 4;
    float* ewr = Ew + (4 * half) * SW_EP + pos_of(nl);

    // ---- depthwise / project roles: (output column n, k group g)
    const int n = lane & 15, g = lane >> 4;
    const float* erd = Ew + n * SW_EP + 8 * g;
    const float* tw = Tall + g * SW_TAPF;

    f32x4 P[SW_SEG / 2][2][2];          // [row pair][row][16-channel block]
#pragma unroll
    for (int i = 0; i < SW_SEG / 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { P[i][j][0] = zero4; P[i][j][1] = zero4; }

    // Chunk operands.  bf / af are read by the expand chain, wp by the project chains: a global load may only target them once a VALU
    // instruction has read the result of an MFMA issued after their last reader (see the stem kernel) -- so the next chunk's bf and its
    // first pixels are requested in the chunk's LAST step behind the epilogue, and wp of a chunk in its step 0 behind the epilogue (by
    // then the previous chunk's project chains have retired; step 0 has no project products of its own).
    f32x4 bf[KK], af[KK];
    float esc, ebi, wp[2][8];
    auto load_bf = [&](int c) {
        const int nch = 32 * c + nl;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
            bf[kk] = nch < hid ? *reinterpret_cast<const f32x4*>(a.we + (size_t)nch * CIN + 8 * kk + 4 * half) : zero4;
    };
    auto load_bn = [&](int c) {
        const int nch = 32 * c + nl;
        esc = nch < hid ? a.se[nch] : 0.f;
        ebi = nch < hid ? a.be[nch] : 0.f;
    };
    const float* pw = a.wp + (size_t)n * hid + kq(g, 0);       // (one lane base + a per-chunk scalar + immediates: kq(g, q) - kq(g, 0) = 8 (q >> 1) + 2 (q & 1))
    auto load_wp = [&](int c) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const float* pm = pw + (size_t)(16 * m) * hid + 32 * c;
            const bool mv = 16 * m + n < a.cout;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k = 32 * c + kq(g, q);
                wp[m][q] = (mv && k < hid) ? pm[8 * (q >> 1) + 2 * (q & 1)] : 0.f;
            }
        }
    };
    auto fetch = [&](int s) {
        const int vo = voff0 + s * rstep;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) af[kk] = bload(rsrc, vo + 32 * kk);
    };
    load_bf(0);
    load_bn(0);
    fetch(0);
    for (int c = 0; c < nchunks; ++c) {
        const float* tc = tw + c * 4 * SW_TAPF;
        auto step = [&](auto SC) {
            constexpr int s = decltype(SC)::value;
            constexpr int par = s & 1;
            constexpr int SL0 = par ? 1 : 3, SL1 = par ? 2 : 0;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) if (!(ABL & 1)) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk][s4], bf[kk][s4], acc, 0, 0, 0);
            {
                const f32x2 sc2 = {esc, esc}, bi2 = {ebi, ebi};
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 v = __builtin_elementwise_fma(f32x2{acc[r], acc[r + 1]}, sc2, bi2);
                    acc[r] = __builtin_amdgcn_fmed3f(v.x, 0.f, 6.f);
                    acc[r + 1] = __builtin_amdgcn_fmed3f(v.y, 0.f, 6.f);
                }
            }
            __builtin_amdgcn_sched_barrier(0);      // (the chain has retired: see above)
            if (s == 0) load_wp(c);
            if (s + 1 < NST && s + 1 < nsteps) fetch(s + 1);
            else if (c + 1 < nchunks) {
                load_bf(c + 1);
                load_bn(c + 1);
                fetch(0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (s == 0) {
                if (top) {
                    asm volatile("");
#pragma unroll
                    for (int r = 0; r < 8; ++r) acc[r] = 0.f;       // expanded row -1
                }
            } else if (bottom && s == nsteps - 1) {
                asm volatile("");
#pragma unroll
                for (int r = 8; r < 16; ++r) acc[r] = 0.f;          // expanded row H
            }
            if (left) { asm volatile(""); if (half == 0) { acc[0] = 0.f; acc[8] = 0.f; } }      // expanded column -1
            if (right) { asm volatile(""); if (half == 1) { acc[7] = 0.f; acc[15] = 0.f; } }    // expanded column W
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) ewr[(r < 8 ? SL0 : SL1) * SW_ROWF + ((r & 3) + 8 * ((r >> 2) & 1)) * SW_EP] = acc[r];
            __builtin_amdgcn_wave_barrier();
            if constexpr (s > 0) {
                constexpr int slot[4] = {par ? 3 : 1, par ? 0 : 2, SL0, SL1};
                f32x2 s0[4], s1[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { s0[i] = f32x2{0.f, 0.f}; s1[i] = f32x2{0.f, 0.f}; }
                f32x2 tprev[3][4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    f32x2 tcur[3][4];
                    if (e < 3) {
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const f32x4 ta = *reinterpret_cast<const f32x4*>(tc + (e * 3 + kx) * 8), tb = *reinterpret_cast<const f32x4*>(tc + (e * 3 + kx) * 8 + 4);
                            tcur[kx][0] = f32x2{ta.x, ta.y}; tcur[kx][1] = f32x2{ta.z, ta.w}; tcur[kx][2] = f32x2{tb.x, tb.y}; tcur[kx][3] = f32x2{tb.z, tb.w};
                        }
                    }
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float* p = erd + slot[e] * SW_ROWF + kx * SW_EP;
                        const f32x4 va = ringr(p), vb = ringr(p + 4);
                        const f32x2 v[4] = {{va.x, va.y}, {va.z, va.w}, {vb.x, vb.y}, {vb.z, vb.w}};
                        if (e < 3) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) pkfma(s0[i], v[i], tcur[kx][i]);
                        }
                        if (e > 0) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) pkfma(s1[i], v[i], tprev[kx][i]);
                        }
                    }
                    if (e < 3) {
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                            for (int i = 0; i < 4; ++i) tprev[kx][i] = tcur[kx][i];
                    }
                }
                const f32x4 sa = *reinterpret_cast<const f32x4*>(tc + 72), sb = *reinterpret_cast<const f32x4*>(tc + 76);
                const f32x4 ba = *reinterpret_cast<const f32x4*>(tc + 80), bb = *reinterpret_cast<const f32x4*>(tc + 84);
                const f32x2 dsc[4] = {{sa.x, sa.y}, {sa.z, sa.w}, {sb.x, sb.y}, {sb.z, sb.w}}, dbi[4] = {{ba.x, ba.y}, {ba.z, ba.w}, {bb.x, bb.y}, {bb.z, bb.w}};
                float d0[8], d1[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x2 r0 = __builtin_elementwise_fma(s0[i], dsc[i], dbi[i]), r1 = __builtin_elementwise_fma(s1[i], dsc[i], dbi[i]);
                    d0[2 * i] = __builtin_amdgcn_fmed3f(r0.x, 0.f, 6.f); d0[2 * i + 1] = __builtin_amdgcn_fmed3f(r0.y, 0.f, 6.f);
                    d1[2 * i] = __builtin_amdgcn_fmed3f(r1.x, 0.f, 6.f); d1[2 * i + 1] = __builtin_amdgcn_fmed3f(r1.y, 0.f, 6.f);
                }
#pragma unroll
                for (int q = 0; q < 8 && !(ABL & 2); ++q) {
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        P[s - 1][0][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[m][q], d0[q], P[s - 1][0][m], 0, 0, 0);
                        P[s - 1][1][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[m][q], d1[q], P[s - 1][1][m], 0, 0, 0);
                    }
                }
            }
        };
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        if (nsteps > 2) step(std::integral_constant<int, 2>{});
        if (nsteps > 3) step(std::integral_constant<int, 3>{});
        if (nsteps > 4) step(std::integral_constant<int, 4>{});
        __builtin_amdgcn_wave_barrier();
    }
    // ---- project BN (+ identity), 16-byte stores: this lane holds output channels 16 m + 4 g .. + 3 of pixel (row, ox0 + n)
    if (n < SW_OW) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int co = 16 * m + 4 * g;
            if (co < a.cout) {
                const f32x4 psc = *reinterpret_cast<const f32x4*>(a.sp + co), pbi = *reinterpret_cast<const f32x4*>(a.bp + co);
#pragma unroll
                for (int i = 0; i < SW_SEG / 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const int y = y0 + 2 * i + j;
                        if (2 * i + j < rows) {
                            const size_t gi = (((size_t)img * H + y) * W + ox0 + n) * a.cout + co;
                            f32x4 v;
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] = fmaf(P[i][j][m][r], psc[r], pbi[r]);
                            if (a.res) {
                                const f32x4 rr = *reinterpret_cast<const f32x4*>(a.res + gi);
                                v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                            }
                            *reinterpret_cast<f32x4*>(a.out2 + gi) = v;
                        }
                    }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// A whole STRIDE-2 inverted-residual block (expand 1x1 -> BN/ReLU6 -> depthwise 3x3 / 2 -> BN/ReLU6 -> project 1x1 -> BN) as strip
// segments: a wave owns 14 output columns x 4 output rows <- 29 (computed: 32) x 9 pixels of the expanded map.  An expanded ROW of the
// strip is one 32-row MFMA band; a step computes two of them (rows 2 y - 2, 2 y - 1 for output row y; the first step of a segment only
// the one row the segment's first output needs), the row shared with the previous output stays in a three-slot ring.  Columns are
// stored even | odd (position (c & 1) * 17 + (c >> 1)) so that the taps' stride-2 reads are unit-stride across lanes.  One output row
// per step: the taps of a chunk fit in registers beside the 32 accumulators of the segment (filled from the LDS table once per chunk).
// The project conv rides in the same launch (b2: 96 -> 24, b4: 144 -> 32 at 224^2 frames): neither expanded map reaches HBM, and the
// project launch and its read of the depthwise map are gone.  Same products in the same order as mb_expand_dw_w_kernel + the engine's
// 1x1 conv: bit-identical.
constexpr int S2_SEG = 4;
constexpr int S2_ROWF = 34 * SW_EP;          // 32 columns + 2 the idle lanes read past, even | odd halves of 17
constexpr int S2_RING = 3 * S2_ROWF;

template <int CIN>
__global__ __launch_bounds__(256, 2) void mb_block_s2_kernel(const MbFuseArgs a) {
    constexpr int KK = CIN / 8, NST = S2_SEG + 1;
    __shared__ __attribute__((aligned(16))) float Eall[4][S2_RING];
    __shared__ __attribute__((aligned(16))) float Tall[SW_MAXCH * 4 * SW_TAPF];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hid = a.hid, nchunks = (hid + 31) >> 5;
    for (int idx = tid; idx < nchunks * 4 * SW_TAPF; idx += 256) {
        const int q = idx & 7, t = (idx >> 3) % 11, cg = idx / SW_TAPF;
        const int ch = 32 * (cg >> 2) + kq(cg & 3, q);
        float v = 0.f;
        if (ch < hid) v = t < 9 ? a.wd[(size_t)t * hid + ch] : t == 9 ? a.sd[ch] : a.bd[ch];
        Tall[idx] = v;
    }
    float* Ew = Eall[wave];
    for (int i = lane; i < S2_RING; i += 64) Ew[i] = 0.f;
    __syncthreads();
    const int nstrips = a.tiles_x, nsegs = a.tiles_y;
    const int item = blockIdx.x * 4 + wave;
    if (item >= a.n * nstrips * nsegs) return;
    const int img = item / (nstrips * nsegs), rem = item - img * (nstrips * nsegs);
    const int seg = rem / nstrips, sx = rem - seg * nstrips;
    const int ox0 = sx * SW_OW, y0 = seg * S2_SEG;
    const int H = a.H, W = a.W, OH = a.OH, OW = a.OW;
    const int rows = OH - y0 < S2_SEG ? OH - y0 : S2_SEG;
    const int nsteps = rows + 1;
    const bool left = sx == 0, top = seg == 0;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // ---- expand GEMM roles: A row p = lane & 31 = expanded column 2 ox0 - 1 + p of expanded row 2 y0 - 2 + 2 s + band
    const int half = lane >> 5, nl = lane & 31;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + (size_t)img * H * W * CIN), 0, H * W * CIN * 4, 0x00020000);
    const int voff0 = (((2 * y0 - 2) * W + 2 * ox0 - 1 + nl) * CIN + 4 * half) * 4;
    const int rband = W * CIN * 4, rstep = 2 * rband;
    float* ewr = Ew + (2 * half) * SW_EP + pos_of(nl);      // accumulator row r: column c = (r & 3) + 8 (r >> 2) + 4 half -> position (c & 1) * 17 + (c >> 1)

    // ---- depthwise / project roles: (output column n, k group g)
    const int n = lane & 15, g = lane >> 4;
    const float* erd = Ew + n * SW_EP + 8 * g;              // tap kx of output column n: expanded column 2 n + kx -> position n | 17 + n | n + 1
    const float* tw = Tall + g * SW_TAPF;

    f32x4 P[S2_SEG][2];
#pragma unroll
    for (int i = 0; i < S2_SEG; ++i) { P[i][0] = zero4; P[i][1] = zero4; }

    // chunk operands: requested behind VALU reads of MFMA results only (see the kernels above)
    f32x4 bf[KK], afa[KK], afb[KK];
    float esc, ebi, wp[2][8];
    auto load_bf = [&](int c) {
        const int nch = 32 * c + nl;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk)
            bf[kk] = nch < hid ? *reinterpret_cast<const f32x4*>(a.we + (size_t)nch * CIN + 8 * kk + 4 * half) : zero4;
    };
    auto load_bn = [&](int c) {
        const int nch = 32 * c + nl;
        esc = nch < hid ? a.se[nch] : 0.f;
        ebi = nch < hid ? a.be[nch] : 0.f;
    };
    const float* pw = a.wp + (size_t)n * hid + kq(g, 0);       // (one lane base + a per-chunk scalar + immediates: kq(g, q) - kq(g, 0) = 8 (q >> 1) + 2 (q & 1))
    auto load_wp = [&](int c) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const float* pm = pw + (size_t)(16 * m) * hid + 32 * c;
            const bool mv = 16 * m + n < a.cout;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k = 32 * c + kq(g, q);
                wp[m][q] = (mv && k < hid) ? pm[8 * (q >> 1) + 2 * (q & 1)] : 0.f;
            }
        }
    };
    auto fetch = [&](int s) {       // (step 0 needs its second row only)
        int vo = voff0 + s * rstep;
        asm volatile("" : "+v"(vo));        // (recomputed per use: hoisted out of the chunk loop the ten step offsets are spilled)
        if (s > 0) {
#pragma unroll
            for (int kk = 0; kk < KK; ++kk) afa[kk] = bload(rsrc, vo + 32 * kk);
        }
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) afb[kk] = bload(rsrc, vo + rband + 32 * kk);
    };
    load_bf(0);
    load_bn(0);
    fetch(0);
    for (int c = 0; c < nchunks; ++c) {
        const float* tc = tw + c * 4 * SW_TAPF;
        // the chunk's taps and depthwise affine: 8 channels x (9 + 2) of this lane's k group, from the table
        f32x2 tap[9][4];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const f32x4 ta = *reinterpret_cast<const f32x4*>(tc + t * 8), tb = *reinterpret_cast<const f32x4*>(tc + t * 8 + 4);
            tap[t][0] = f32x2{ta.x, ta.y}; tap[t][1] = f32x2{ta.z, ta.w}; tap[t][2] = f32x2{tb.x, tb.y}; tap[t][3] = f32x2{tb.z, tb.w};
        }
        auto step = [&](auto SC) {
            constexpr int s = decltype(SC)::value;
            constexpr int SLA = (2 * s) % 3, SLB = (2 * s + 1) % 3, SLP = (2 * s + 2) % 3;      // (2 s - 1) % 3
            // one expanded row = one band: chain -> BN / ReLU6 -> ring, row A (absent in step 0) then row B
            auto band = [&](const f32x4 (&af)[KK], int slotf, bool zero_row) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) if (!(ABL & 1)) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk][s4], bf[kk][s4], acc, 0, 0, 0);
                const f32x2 sc2 = {esc, esc}, bi2 = {ebi, ebi};
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 v = __builtin_elementwise_fma(f32x2{acc[r], acc[r + 1]}, sc2, bi2);
                    acc[r] = __builtin_amdgcn_fmed3f(v.x, 0.f, 6.f);
                    acc[r + 1] = __builtin_amdgcn_fmed3f(v.y, 0.f, 6.f);
                }
                if (zero_row) {
                    asm volatile("");
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;          // expanded row -1
                }
                if (left) { asm volatile(""); if (half == 0) acc[0] = 0.f; }     // expanded column -1
#pragma unroll
                for (int r = 0; r < 16; ++r) ewr[slotf + ((r & 1) * 17 + ((r & 3) >> 1) + 4 * (r >> 2)) * SW_EP] = acc[r];
            };
            __builtin_amdgcn_wave_barrier();        // the previous step's tap reads are issued
            if (s > 0) band(afa, SLA * S2_ROWF, false);
            band(afb, SLB * S2_ROWF, s == 0 && top);
            __builtin_amdgcn_sched_barrier(0);      // (both chains have retired: their results were read)
            if (s == 0) load_wp(c);
            if (s + 1 < NST && s + 1 < nsteps) fetch(s + 1);
            else if (c + 1 < nchunks) {
                load_bf(c + 1);
                load_bn(c + 1);
                fetch(0);
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_wave_barrier();
            if constexpr (s > 0) {
                constexpr int slot[3] = {SLP, SLA, SLB};
                f32x2 s0[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) s0[i] = f32x2{0.f, 0.f};
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float* p = erd + slot[ky] * S2_ROWF + (kx == 1 ? 17 : kx == 2 ? 1 : 0) * SW_EP;
                        const f32x4 va = ringr(p), vb = ringr(p + 4);
                        const f32x2 v[4] = {{va.x, va.y}, {va.z, va.w}, {vb.x, vb.y}, {vb.z, vb.w}};
#pragma unroll
                        for (int i = 0; i < 4; ++i) pkfma(s0[i], v[i], tap[ky * 3 + kx][i]);
                    }
                    __builtin_amdgcn_sched_barrier(0);      // (one expanded row's reads in flight at a time: 24 registers, not 72)
                }
                const f32x4 sa = *reinterpret_cast<const f32x4*>(tc + 72), sb = *reinterpret_cast<const f32x4*>(tc + 76);
                const f32x4 ba = *reinterpret_cast<const f32x4*>(tc + 80), bb = *reinterpret_cast<const f32x4*>(tc + 84);
                const f32x2 dsc[4] = {{sa.x, sa.y}, {sa.z, sa.w}, {sb.x, sb.y}, {sb.z, sb.w}}, dbi[4] = {{ba.x, ba.y}, {ba.z, ba.w}, {bb.x, bb.y}, {bb.z, bb.w}};
                float d0[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x2 r0 = __builtin_elementwise_fma(s0[i], dsc[i], dbi[i]);
                    d0[2 * i] = __builtin_amdgcn_fmed3f(r0.x, 0.f, 6.f); d0[2 * i + 1] = __builtin_amdgcn_fmed3f(r0.y, 0.f, 6.f);
                }
#pragma unroll
                for (int q = 0; q < 8 && !(ABL & 2); ++q) {
                    P[s - 1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[0][q], d0[q], P[s - 1][0], 0, 0, 0);
                    P[s - 1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[1][q], d0[q], P[s - 1][1], 0, 0, 0);
                }
            }
        };
        step(std::integral_constant<int, 0>{});
        step(std::integral_constant<int, 1>{});
        if (nsteps > 2) step(std::integral_constant<int, 2>{});
        if (nsteps > 3) step(std::integral_constant<int, 3>{});
        if (nsteps > 4) step(std::integral_constant<int, 4>{});
        __builtin_amdgcn_wave_barrier();
    }
    if (n < SW_OW) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const int co = 16 * m + 4 * g;
            if (co < a.cout) {
                const f32x4 psc = *reinterpret_cast<const f32x4*>(a.sp + co), pbi = *reinterpret_cast<const f32x4*>(a.bp + co);
#pragma unroll
                for (int i = 0; i < S2_SEG; ++i)
                    if (i < rows) {
                        const size_t gi = (((size_t)img * OH + y0 + i) * OW + ox0 + n) * a.cout + co;
                        f32x4 v;
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = fmaf(P[i][m][r], psc[r], pbi[r]);
                        if (a.res) {
                            const f32x4 rr = *reinterpret_cast<const f32x4*>(a.res + gi);
                            v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                        }
                        *reinterpret_cast<f32x4*>(a.out2 + gi) = v;
                    }
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Expand 1x1 (+ BN + ReLU6) -> depthwise 3x3 (+ BN + ReLU6) of the stride-1 blocks on 14 x 14 maps (b8-b13 of MobileNetV2 at 224^2: 64 / 96 input
// channels, 384 / 576 hidden): the 6x-expanded map -- 44 % of such a block's HBM bytes, written by a short-K GEMM the conv engine runs at a
// third of its rate (K = 64: two slices per tile, the tile's prologue and epilogue are most of its time) -- never exists.  A wave walks a
// whole 14-column strip of a frame top to bottom, once per chunk of 32 hidden channels (chunk loop outside: the chunk's filter rows stay in
// registers for the 8 steps); there is no project conv behind the taps here (its 64 / 96 x 384 / 576 accumulators would not fit a wave: it
// stays on the engine), so a depthwise lane is (column n, channel octet cq = lane >> 4) with 8 CONSECUTIVE channels -- natural channel order
// in the ring, two 16-byte stores per output row -- and nothing lives across the chunks.  Same products in the same order as the engine's
// 1x1 conv (k slices of 32, a lane's 16-byte fragment) and dwconv3x3_kernel: bit-identical to the two launches.
constexpr int XD_MAXCH = 18;                 // chunks (hid <= 576)

template <int CIN, int STRIDE = 1>
__global__ __launch_bounds__(256, 2) void mb_expand_dw_s_kernel(const MbFuseArgs a) {
    constexpr int KK = CIN / 8;
    __shared__ __attribute__((aligned(16))) float Eall[4][SW_RING];
    __shared__ __attribute__((aligned(16))) float Tall[XD_MAXCH * 4 * SW_TAPF];      // [chunk][octet][9 taps | scale | bias][8]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hid = a.hid, nchunks = hid >> 5;
    for (int idx = tid; idx < nchunks * 4 * SW_TAPF; idx += 256) {
        const int q = idx & 7, t = (idx >> 3) % 11, cg = idx / SW_TAPF;
        const int ch = 8 * cg + q;                       // (32 (cg >> 2) + 8 (cg & 3) + q)
        Tall[idx] = t < 9 ? a.wd[(size_t)t * hid + ch] : t == 9 ? a.sd[ch] : a.bd[ch];
    }
    float* Ew = Eall[wave];
    for (int i = lane; i < SW_RING; i += 64) Ew[i] = 0.f;
    __syncthreads();
    // a work item = (strip, chunk group): nothing lives across the chunks, so the hidden channels are also cut over the waves -- a 14 x 14 map is ONE
    // strip per frame, 512 waves for a 512-frame chunk would leave three quarters of the wave slots empty.  The waves of a block take the groups
    // of one strip: they read the same pixels at about the same time.
    const int nstrips = a.tiles_x, ngroups = a.tiles_y;
    int strip = blockIdx.x * 4 + wave;
    if (strip >= a.n * nstrips * ngroups) return;
    const int cgrp = strip % ngroups;
    strip /= ngroups;
    const int c_lo = (nchunks * cgrp) / ngroups, c_hi = (nchunks * (cgrp + 1)) / ngroups;
    const int img = strip / nstrips, sx = strip - img * nstrips;
    const int ox0 = sx * SW_OW;
    const bool left = sx == 0, right = sx == nstrips - 1;
    const int H = a.H, W = a.W;
    const int nsteps = H / 2 + 1;

    // ---- expand GEMM roles: A row p = lane & 31 = (er = p >> 4, ec = p & 15) is input pixel (2 s - 1 + er, ox0 - 1 + ec), k = 8 kk + 4 half ..
    const int half = lane >> 5, nl = lane & 31;
    const int er = nl >> 4, ec = nl & 15;
    // (temporal shift of the block's input, MbFuseArgs::tsm_T: the descriptor spans the frame's CLIP, a lane's k group reads its own frame, the next or
    //  the previous one, and a neighbour outside the clip is out of range = zeros.  Out-of-image pixels then land in a neighbouring frame instead of out
    //  of range: every one of them is overwritten by the row / column masks below.)
    const int tsmT = a.tsm_T, tfr = tsmT > 0 ? img % tsmT : 0, fbytes = H * W * CIN * 4;
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + (size_t)(img - tfr) * H * W * CIN), 0,
                                                                     (tsmT > 0 ? tsmT : 1) * fbytes, 0x00020000);
    const int voff0 = (((er - 1) * W + ox0 - 1 + ec) * CIN + 4 * half) * 4 + tfr * fbytes;
    constexpr int KS = (KK + 3) / 4;         // the two shifted folds are the first CIN / 4 channels: only the first KK / 4 loads of a pixel carry an offset
    int dsh[KS];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
        const int ch = 8 * kk + 4 * half;
        dsh[kk] = tsmT > 0 ? (ch < a.tsm_fold ? fbytes : ch < 2 * a.tsm_fold ? -fbytes : 0) : 0;
    }
    const int rstep = 2 * W * CIN * 4;
    float* ewr = Ew + (4 * half) * SW_EP + nl;              // natural channel order in the ring

    // ---- depthwise roles: (output column n, channel octet cq)
    // (STRIDE 2 -- block 14, 14 x 14 -> 7 x 7: output column n of the strip's seven reads ring positions 2 n .. 2 n + 2, one output row per step)
    const int n = lane & 15, cq = lane >> 4;
    const float* erd = Ew + STRIDE * n * SW_EP + 8 * cq;
    const float* tw = Tall + cq * SW_TAPF;
    const int OWs = STRIDE == 1 ? W : W / 2;
    float* const obase = a.out + ((size_t)img * (STRIDE == 1 ? H : H / 2) * OWs + ox0 / STRIDE + n) * hid + 8 * cq;       // output row 0 of this lane's column
    const bool ostore = n < SW_OW / STRIDE;

    f32x4 bf[KK], af[KK];
    float esc, ebi;
    auto load_bf = [&](int c) {
        const int nch = 32 * c + nl;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) bf[kk] = *reinterpret_cast<const f32x4*>(a.we + (size_t)nch * CIN + 8 * kk + 4 * half);
        esc = a.se[nch];
        ebi = a.be[nch];
    };
    auto fetch = [&](int s) {
        const int vo = voff0 + s * rstep;
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) af[kk] = bload(rsrc, kk < KS ? vo + 32 * kk + dsh[kk] : vo + 32 * kk);
    };
    load_bf(c_lo);
    fetch(0);
    for (int c = c_lo; c < c_hi; ++c) {
        const float* tc = tw + c * 4 * SW_TAPF;
        float* optr = obase + 32 * c;
        auto step = [&](auto PAR, int s) {
            constexpr int par = decltype(PAR)::value;
            constexpr int SL0 = par ? 1 : 3, SL1 = par ? 2 : 0;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < KK; ++kk)
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) if (!(ABL & 1)) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk][s4], bf[kk][s4], acc, 0, 0, 0);
            {
                const f32x2 sc2 = {esc, esc}, bi2 = {ebi, ebi};
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const f32x2 v = __builtin_elementwise_fma(f32x2{acc[r], acc[r + 1]}, sc2, bi2);
                    acc[r] = __builtin_amdgcn_fmed3f(v.x, 0.f, 6.f);
                    acc[r + 1] = __builtin_amdgcn_fmed3f(v.y, 0.f, 6.f);
                }
            }
            // (the chain has retired -- its result was read: the landing registers of the next requests are free; see the stem kernel)
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < nsteps) fetch(s + 1);
            else if (c + 1 < c_hi) {
                load_bf(c + 1);
                fetch(0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (s == 0) {
                asm volatile("");
#pragma unroll
                for (int r = 0; r < 8; ++r) acc[r] = 0.f;           // expanded row -1
            }
            if (s == nsteps - 1) {
                asm volatile("");
#pragma unroll
                for (int r = 8; r < 16; ++r) acc[r] = 0.f;          // expanded row H
            }
            if (left) { asm volatile(""); if (half == 0) { acc[0] = 0.f; acc[8] = 0.f; } }      // expanded column -1
            if (right) { asm volatile(""); if (half == 1) { acc[7] = 0.f; acc[15] = 0.f; } }    // expanded column W
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < 16; ++r) ewr[(r < 8 ? SL0 : SL1) * SW_ROWF + ((r & 3) + 8 * ((r >> 2) & 1)) * SW_EP] = acc[r];
            __builtin_amdgcn_wave_barrier();
            if (s > 0 && STRIDE == 2) {
                // output row s - 1 from expanded rows 2 s - 3, 2 s - 2 (the previous step's) and 2 s - 1 (this step's first)
                constexpr int slot[3] = {par ? 3 : 1, par ? 0 : 2, SL0};
                f32x2 s0[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) s0[i] = f32x2{0.f, 0.f};
#pragma unroll
                for (int e = 0; e < 3; ++e) {
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const f32x4 ta = *reinterpret_cast<const f32x4*>(tc + (e * 3 + kx) * 8), tb = *reinterpret_cast<const f32x4*>(tc + (e * 3 + kx) * 8 + 4);
                        const f32x2 t4[4] = {{ta.x, ta.y}, {ta.z, ta.w}, {tb.x, tb.y}, {tb.z, tb.w}};
                        const float* p = erd + slot[e] * SW_ROWF + kx * SW_EP;
                        const f32x4 va = ringr(p), vb = ringr(p + 4);
                        const f32x2 v[4] = {{va.x, va.y}, {va.z, va.w}, {vb.x, vb.y}, {vb.z, vb.w}};
#pragma unroll
                        for (int i = 0; i < 4; ++i) pkfma(s0[i], v[i], t4[i]);
                    }
                }
                const f32x4 sa = *reinterpret_cast<const f32x4*>(tc + 72), sb = *reinterpret_cast<const f32x4*>(tc + 76);
                const f32x4 ba = *reinterpret_cast<const f32x4*>(tc + 80), bb = *reinterpret_cast<const f32x4*>(tc + 84);
                const f32x2 dsc[4] = {{sa.x, sa.y}, {sa.z, sa.w}, {sb.x, sb.y}, {sb.z, sb.w}}, dbi[4] = {{ba.x, ba.y}, {ba.z, ba.w}, {bb.x, bb.y}, {bb.z, bb.w}};
                float d0[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x2 r0 = __builtin_elementwise_fma(s0[i], dsc[i], dbi[i]);
                    d0[2 * i] = __builtin_amdgcn_fmed3f(r0.x, 0.f, 6.f); d0[2 * i + 1] = __builtin_amdgcn_fmed3f(r0.y, 0.f, 6.f);
                }
                if (ostore) {
                    float* o = optr + (size_t)(s - 1) * OWs * hid;
                    *reinterpret_cast<f32x4*>(o) = f32x4{d0[0], d0[1], d0[2], d0[3]};
                    *reinterpret_cast<f32x4*>(o + 4) = f32x4{d0[4], d0[5], d0[6], d0[7]};
                }
            }
            if (s > 0 && STRIDE == 1) {
                constexpr int slot[4] = {par ? 3 : 1, par ? 0 : 2, SL0, SL1};
                f32x2 s0[4], s1[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { s0[i] = f32x2{0.f, 0.f}; s1[i] = f32x2{0.f, 0.f}; }
                f32x2 tprev[3][4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    f32x2 tcur[3][4];
                    if (e < 3) {
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const f32x4 ta = *reinterpret_cast<const f32x4*>(tc + (e * 3 + kx) * 8), tb = *reinterpret_cast<const f32x4*>(tc + (e * 3 + kx) * 8 + 4);
                            tcur[kx][0] = f32x2{ta.x, ta.y}; tcur[kx][1] = f32x2{ta.z, ta.w}; tcur[kx][2] = f32x2{tb.x, tb.y}; tcur[kx][3] = f32x2{tb.z, tb.w};
                        }
                    }
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const float* p = erd + slot[e] * SW_ROWF + kx * SW_EP;
                        const f32x4 va = ringr(p), vb = ringr(p + 4);
                        const f32x2 v[4] = {{va.x, va.y}, {va.z, va.w}, {vb.x, vb.y}, {vb.z, vb.w}};
                        if (e < 3) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) pkfma(s0[i], v[i], tcur[kx][i]);
                        }
                        if (e > 0) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) pkfma(s1[i], v[i], tprev[kx][i]);
                        }
                    }
                    if (e < 3) {
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                            for (int i = 0; i < 4; ++i) tprev[kx][i] = tcur[kx][i];
                    }
                }
                const f32x4 sa = *reinterpret_cast<const f32x4*>(tc + 72), sb = *reinterpret_cast<const f32x4*>(tc + 76);
                const f32x4 ba = *reinterpret_cast<const f32x4*>(tc + 80), bb = *reinterpret_cast<const f32x4*>(tc + 84);
                const f32x2 dsc[4] = {{sa.x, sa.y}, {sa.z, sa.w}, {sb.x, sb.y}, {sb.z, sb.w}}, dbi[4] = {{ba.x, ba.y}, {ba.z, ba.w}, {bb.x, bb.y}, {bb.z, bb.w}};
                float d0[8], d1[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x2 r0 = __builtin_elementwise_fma(s0[i], dsc[i], dbi[i]), r1 = __builtin_elementwise_fma(s1[i], dsc[i], dbi[i]);
                    d0[2 * i] = __builtin_amdgcn_fmed3f(r0.x, 0.f, 6.f); d0[2 * i + 1] = __builtin_amdgcn_fmed3f(r0.y, 0.f, 6.f);
                    d1[2 * i] = __builtin_amdgcn_fmed3f(r1.x, 0.f, 6.f); d1[2 * i + 1] = __builtin_amdgcn_fmed3f(r1.y, 0.f, 6.f);
                }
                if (ostore) {
                    float* o = optr + (size_t)(2 * s - 2) * W * hid;
                    *reinterpret_cast<f32x4*>(o) = f32x4{d0[0], d0[1], d0[2], d0[3]};
                    *reinterpret_cast<f32x4*>(o + 4) = f32x4{d0[4], d0[5], d0[6], d0[7]};
                    *reinterpret_cast<f32x4*>(o + (size_t)W * hid) = f32x4{d1[0], d1[1], d1[2], d1[3]};
                    *reinterpret_cast<f32x4*>(o + (size_t)W * hid + 4) = f32x4{d1[4], d1[5], d1[6], d1[7]};
                }
            }
        };
        for (int s = 0; s < nsteps; s += 2) {
            step(std::integral_constant<int, 0>{}, s);
            if (s + 1 < nsteps) step(std::integral_constant<int, 1>{}, s + 1);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

}  // namespace

bool adaf_mb_stem_b1_strip_ok(int S, int H1) { return adaf_options().mb_strip != 0 && S == 2 * H1 && H1 % SW_OW == 0 && H1 >= SW_OW; }

void adaf_launch_mb_stem_b1_strip(MbStemArgs a, hipStream_t s) {
    a.tiles_x = a.H1 / SW_OW;
    a.tiles_y = 1;
    a.total_tiles = a.n * a.tiles_x;
    hipLaunchKernelGGL(mb_stem_b1_s_kernel, dim3((unsigned)((a.total_tiles + 3) / 4)), dim3(256), 0, s, a);
}

// whole stride-1 blocks as strip segments (b3, b5, b6 of MobileNetV2 1.0 at 224^2): square maps whose side is a multiple of 14
bool adaf_mb_block_strip_ok(int cin, int hid, int cout, int stride, int h, int w) {
    if (adaf_options().mb_strip == 0 || hid > 32 * SW_MAXCH || hid % 4 || cout % 4 || cout > 32 || h != w) return false;
    if (stride == 1) return (cin == 24 || cin == 32) && w % SW_OW == 0 && h % 2 == 0;
    // stride 2 (b2, b4 at 224^2): even maps whose half is a multiple of 14
    return stride == 2 && (cin == 16 || cin == 24) && w % 2 == 0 && (w / 2) % SW_OW == 0;
}

void adaf_launch_mb_block_strip(MbFuseArgs a, hipStream_t s) {
    if (a.OH != a.H) {      // stride 2
        a.tiles_x = a.OW / SW_OW;
        a.tiles_y = (a.OH + S2_SEG - 1) / S2_SEG;
        const long long items = (long long)a.n * a.tiles_x * a.tiles_y;
        const dim3 grid((unsigned)((items + 3) / 4)), block(256);
        if (a.cin == 16) hipLaunchKernelGGL((mb_block_s2_kernel<16>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((mb_block_s2_kernel<24>), grid, block, 0, s, a);
        return;
    }
    a.tiles_x = a.W / SW_OW;
    a.tiles_y = (a.H + SW_SEG - 1) / SW_SEG;
    const long long items = (long long)a.n * a.tiles_x * a.tiles_y;
    const dim3 grid((unsigned)((items + 3) / 4)), block(256);
    if (a.cin == 24) hipLaunchKernelGGL((mb_block_s_kernel<24>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((mb_block_s_kernel<32>), grid, block, 0, s, a);
}

// expand -> depthwise of the stride-1 blocks with 64 / 96 input channels on maps whose side is a multiple of 14 (b8-b13 at 224^2 frames),
// and of the stride-2 block with 96 (b14: 14 x 14 -> 7 x 7)
bool adaf_mb_expand_dw_strip_ok(int cin, int hid, int stride, int h, int w) {
    return adaf_options().mb_strip != 0 && (stride == 1 || (stride == 2 && cin == 96)) && (cin == 64 || cin == 96) && hid % 32 == 0 && hid <= 32 * XD_MAXCH &&
           h == w && w % SW_OW == 0 && h % 2 == 0;
}

void adaf_launch_mb_expand_dw_strip(MbFuseArgs a, hipStream_t s) {
    a.tiles_x = a.W / SW_OW;
    // chunk groups: enough waves to fill the device's 2048 wave slots (two per SIMD) once -- 512 frames: four groups (12 chunks: 3 each; 18: 5, 4, 5, 4);
    // fewer frames (the Something-Something glancer's 256-frame half chunks, small batches): more, smaller groups, down to one chunk per wave
    const int nchunks = a.hid / 32;
    int groups = nchunks >= 4 ? 4 : 1;
    while ((long long)a.n * a.tiles_x * groups < 2048 && groups < nchunks) {
        int g = groups + 1;
        while (g < nchunks && nchunks % g) ++g;      // next divisor of the chunk count: equal groups
        groups = g;
    }
    a.tiles_y = groups;
    const long long items = (long long)a.n * a.tiles_x * a.tiles_y;
    const dim3 grid((unsigned)((items + 3) / 4)), block(256);
    if (a.OH != a.H) hipLaunchKernelGGL((mb_expand_dw_s_kernel<96, 2>), grid, block, 0, s, a);      // stride 2 (block 14)
    else if (a.cin == 64) hipLaunchKernelGGL((mb_expand_dw_s_kernel<64>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((mb_expand_dw_s_kernel<96>), grid, block, 0, s, a);
}
