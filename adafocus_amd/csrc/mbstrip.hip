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

namespace {

// k offset (inside a 32-channel slice) that lane group g supplies as its q-th product of the 16x16x4 chain
__device__ __forceinline__ int kq(int g, int q) { return 8 * (q >> 1) + (g >> 1) + 4 * (g & 1) + 2 * (q & 1); }
// ... and its inverse: position 8 g + q of channel c in the [g][q] order the depthwise lanes read
__device__ __forceinline__ int pos_of(int c) {
    const int e = c & 7, second = (e >> 1) & 1, f = e - 2 * second;
    return 8 * (((f & 1) << 1) | (f >> 2)) + 2 * (c >> 3) + second;
}

__device__ __forceinline__ f32x4 bload(__amdgpu_buffer_rsrc_t r, int byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 0));
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
    float* optr = a.out + (((size_t)img * H1 - 2) * H1 + ox0 + n) * 16 + 4 * g;       // row 2 s - 2 at step s
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
        if (left) {
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
            for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk][s4], bs[kk][s4], acc, 0, 0, 0);
        if (s + 1 < nsteps) fetch();
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
        if (s == 0) {
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r] = 0.f;           // E row -1
        }
        if (s == nsteps - 1) {
#pragma unroll
            for (int r = 8; r < 16; ++r) acc[r] = 0.f;          // E row H1
        }
        if (left && half == 0) { acc[0] = 0.f; acc[8] = 0.f; }     // E column -1   (ec = 0: rows r = 0, 8 of half 0)
        if (right && half == 1) { acc[7] = 0.f; acc[15] = 0.f; }   // E column H1   (ec = 15: rows r = 7, 15 of half 1)
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
                    const f32x4 va = *reinterpret_cast<const f32x4*>(p), vb = *reinterpret_cast<const f32x4*>(p + 4);
                    const f32x2 v[4] = {{va.x, va.y}, {va.z, va.w}, {vb.x, vb.y}, {vb.z, vb.w}};
                    if (e < 3) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) s0[i] = __builtin_elementwise_fma(v[i], tap[e * 3 + kx][i], s0[i]);
                    }
                    if (e > 0) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) s1[i] = __builtin_elementwise_fma(v[i], tap[(e - 1) * 3 + kx][i], s1[i]);
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
            for (int q = 0; q < 8; ++q) {
                p0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[q], d0[q], p0, 0, 0, 0);
                p1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wp[q], d1[q], p1, 0, 0, 0);
            }
            if (ostore) {
                f32x4 o0, o1;
#pragma unroll
                for (int i = 0; i < 4; ++i) { o0[i] = fmaf(p0[i], psc[i], pbi[i]) + 0.f; o1[i] = fmaf(p1[i], psc[i], pbi[i]) + 0.f; }
                *reinterpret_cast<f32x4*>(optr) = o0;
                *reinterpret_cast<f32x4*>(optr + (size_t)H1 * 16) = o1;
            }
        }
        optr += (size_t)2 * H1 * 16;
    };
    for (int s = 0; s < nsteps; s += 2) {
        step(std::integral_constant<int, 0>{}, s);
        if (s + 1 < nsteps) step(std::integral_constant<int, 1>{}, s + 1);
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
