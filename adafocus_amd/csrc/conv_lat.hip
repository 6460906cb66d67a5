// Convolutions for SMALL batches (BASELINE config 1 is B = 2, T = 8: 16 patches): the latency form of the conv engine.
//
// At 16 patches a ResNet-50 step is ~50 dependent launches of a few dozen blocks each, and a launch lasts as long as ONE
// accumulator chain: conv_gemm.hip's waves own 32x32 (or larger) tiles and v_mfma_f32_32x32x2_f32 retires 2 k per 64 cycles,
// so a K = 4608 conv is 147 k cycles = 61 us however many CUs idle next to it (stage 4's 3x3: 107-115 us measured, 24 blocks).
// Splitting K would change the summation order, and with it the bits: a clip's logits must not depend on the batch it came in.
// v_mfma_f32_16x16x4_f32 retires 4 k per 40 cycles of dependent latency -- a chain 3.2x shorter -- and it IS the same arithmetic:
// both instructions are exact fp32 fma chains over their k lanes in ascending lane-group order, so with the lane groups of the
// 16x16x4 taking k offsets {0, 4, 1, 5} and then {2, 6, 3, 7} of every 8-k group (the order in which the engine's four products
// x, y, z, w of a 16-byte fragment visit them) the results are BIT-IDENTICAL to the engine's -- checked instruction against
// instruction and against a host fmaf chain before this file was written, and per layer by tests/test_hip_parity_r3.py.
//
// Block = 4 waves = a 32 x 32 output tile (2 x 2 sub-tiles of 16 x 16), K walked in slices of KC = 64 | 128 floats inside one
// filter tap, double-buffered in LDS with the next slice's global loads in flight during the products (one barrier per slice).
// Inside LDS every 8-k group is stored as [k0 k2 k4 k6 k1 k3 k5 k7], so lane group g reads its two values as ONE ds_read_b64.
// Same epilogue arithmetic as the engine (BN affine, + identity, clamp).  fp32, no temporal shift, K % 64 == 0.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "adaf_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

template <int KC>
__global__ __launch_bounds__(256) void conv_lat_kernel(const ConvArgs a) {
    constexpr int PITCH = KC + 4;                 // floats per staged row (+16 bytes: the 32 lanes of a half-wave ds_read_b64 hit the 64 banks once each)
    constexpr int PPT = KC / 32;                  // 16-byte pieces per thread per operand per slice (32 rows x KC / 4 pieces / 256 threads)
    __shared__ __attribute__((aligned(16))) float smem[2][2][32 * PITCH];   // [buffer][A | B][row][k]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r16 = lane & 15, kg = lane >> 4;
    const int tiles_n = (a.N + 31) >> 5;
    const int m0 = (blockIdx.x / tiles_n) * 32, n0 = (blockIdx.x % tiles_n) * 32;
    // ---- the row / filter row this thread stages: row = tid / 8, pieces (tid % 8) + 8 j
    const int srow = tid >> 3, sp = tid & 7;
    const int m = m0 + srow;
    const bool m_ok = m < a.M;
    long long boff = 0;
    unsigned tapmask = 0;
    {
        const int mm = m_ok ? m : 0;
        const int ohw = a.OH * a.OW;
        const int img = mm / ohw, rem = mm - img * ohw;
        const int oy = rem / a.OW, ox = rem - oy * a.OW;
        const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
        boff = ((long long)img * a.H * a.W + (long long)iy0 * a.W + ix0) * a.ldx;
        for (int kh = 0; kh < a.KH; ++kh)
            for (int kw = 0; kw < a.KW; ++kw)
                if (m_ok && (unsigned)(iy0 + kh) < (unsigned)a.H && (unsigned)(ix0 + kw) < (unsigned)a.W) tapmask |= 1u << (kh * a.KW + kw);
    }
    const bool n_ok = n0 + srow < a.N;
    const float* wrow = a.w + (size_t)(n_ok ? n0 + srow : 0) * a.K;
    const int cslices = a.cin / KC;               // slices per tap
    const int nslices = a.KH * a.KW * cslices;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // The slices of one block are a dependent chain (one accumulator).  Round 3's loop measured 1.4-1.5 us per 128-k slice against
    // ~0.5 us of chain, and 1, 2 or 4 slices of prefetch all the same; round 5 read the ISA and found why: (i) every load was written
    // `ok ? *p : zero` -- its own exec-masked branch behind a v_mov of zeros into its destination -- and (ii) the prefetch sets were
    // an array indexed by `s % DEPTH` inside a loop with a break, which hipcc compiled as ONE loop body that loads into scratch
    // registers, waits s_waitcnt vmcnt(0) and MOVES them into the set: every slice drained the loads it had just issued, i.e. paid a
    // full memory round trip.  Now: NSET explicit register sets (slice j lives in set j % NSET; the loop body is NSET steps with
    // compile-time sets, no moves), UNCONDITIONAL loads from a selected address (the handle's block of zeros for padding taps, rows
    // past M, filters past N; past the last slice the last slice again), so a slice's loads have NSET - 1 steps to land and the only
    // wait is the counted one in front of the LDS hand-over; and the hand-over of slice s+1 is issued BEFORE the products of slice s
    // (its buffer was released by the previous barrier), so the chain of 2 KC / 8 dependent MFMAs runs while the writes complete.
    // Same products in the same order: bit-identical (tests/test_hip_parity_r3.py).
    constexpr int NSET = 4;
    f32x4 pa[NSET][PPT], pb[NSET][PPT];
    auto gload = [&](int s_req, f32x4 (&ra)[PPT], f32x4 (&rb)[PPT]) {
        const int s = s_req < nslices ? s_req : nslices - 1;
        const int tap = s / cslices, c0 = (s - tap * cslices) * KC;
        const int kh = tap / a.KW, kw = tap - kh * a.KW;
        const bool ok = (tapmask >> tap) & 1u;
        const float* src = a.x + boff + ((long long)kh * a.W + kw) * a.ldx + c0;
        const float* wsrc = wrow + (size_t)tap * a.cin + c0;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int p = sp + 8 * j;
            const float* pa_ = ok ? src + 4 * p : a.zeros;
            const float* pb_ = n_ok ? wsrc + 4 * p : a.zeros;
            ra[j] = *reinterpret_cast<const f32x4*>(pa_);
            rb[j] = *reinterpret_cast<const f32x4*>(pb_);
        }
    };
    auto lstore = [&](int buf, const f32x4 (&ra)[PPT], const f32x4 (&rb)[PPT]) {
        float* As = smem[buf][0] + srow * PITCH;
        float* Bs = smem[buf][1] + srow * PITCH;
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int p = sp + 8 * j;             // piece p holds k = 4p .. 4p+3 of the slice: group p / 2, half p % 2
            const int base = (p >> 1) * 8 + (p & 1) * 2;          // [k0 k2 k4 k6 | k1 k3 k5 k7]: (x, z) at base, (y, w) at base + 4
            *reinterpret_cast<f32x2*>(As + base) = f32x2{ra[j].x, ra[j].z};
            *reinterpret_cast<f32x2*>(As + base + 4) = f32x2{ra[j].y, ra[j].w};
            *reinterpret_cast<f32x2*>(Bs + base) = f32x2{rb[j].x, rb[j].z};
            *reinterpret_cast<f32x2*>(Bs + base + 4) = f32x2{rb[j].y, rb[j].w};
        }
    };
#pragma unroll
    for (int d = 0; d < NSET; ++d) gload(d, pa[d], pb[d]);
    // the epilogue's operands travel under the K loop too
    const int n = n0 + wn * 16 + r16;
    const bool col_ok = n < a.N;
    const float sc = (a.scale && col_ok) ? a.scale[n] : 1.f, bi = (a.bias && col_ok) ? a.bias[n] : 0.f;
    float rv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int mo = m0 + wm * 16 + 4 * kg + i;
        rv[i] = (a.res && col_ok && mo < a.M) ? a.res[(size_t)mo * a.ldr + n] : 0.f;
    }
    f32x4 acc = zero4;
    lstore(0, pa[0], pb[0]);
    __syncthreads();
    const float* Ar = smem[0][0] + (wm * 16 + r16) * PITCH + 2 * kg;
    const float* Br = smem[0][1] + (wn * 16 + r16) * PITCH + 2 * kg;
    constexpr int BUFSTRIDE = 2 * 32 * PITCH;     // floats between the two buffers
    // one step: slice s (in LDS buffer BUF = s & 1).  LOAD: set `rs` (which held slice s, handed to LDS one step ago) is refilled
    // with slice s + NSET; the hand-over of slice s + 1 (set `ns`) goes out before the products.
    auto step = [&](int s, auto buf_tag, auto load_tag, f32x4 (&rsa)[PPT], f32x4 (&rsb)[PPT], const f32x4 (&nsa)[PPT], const f32x4 (&nsb)[PPT]) {
        constexpr int BUF = decltype(buf_tag)::value;
        constexpr bool LOAD = decltype(load_tag)::value;
        if (LOAD) gload(s + NSET, rsa, rsb);
        // every fragment of the slice is requested before the first product
        f32x2 av[KC / 8], bv[KC / 8];
#pragma unroll
        for (int g = 0; g < KC / 8; ++g) {
            av[g] = *reinterpret_cast<const f32x2*>(Ar + BUF * BUFSTRIDE + 8 * g);
            bv[g] = *reinterpret_cast<const f32x2*>(Br + BUF * BUFSTRIDE + 8 * g);
        }
        if (s + 1 < nslices) lstore(BUF ^ 1, nsa, nsb);       // (that buffer was last read before the previous barrier)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < KC / 8; ++g) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g].x, bv[g].x, acc, 0, 0, 0);     // k offsets {0, 4, 1, 5} of the group
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[g].y, bv[g].y, acc, 0, 0, 0);     // k offsets {2, 6, 3, 7}
        }
        __syncthreads();
    };
    using B0 = std::integral_constant<int, 0>;
    using B1 = std::integral_constant<int, 1>;
    int s0 = 0;
    for (; s0 + NSET <= nslices; s0 += NSET) {      // NSET is even: the buffer of a step is a compile-time constant
        step(s0, B0{}, std::true_type{}, pa[0], pb[0], pa[1], pb[1]);
        step(s0 + 1, B1{}, std::true_type{}, pa[1], pb[1], pa[2], pb[2]);
        step(s0 + 2, B0{}, std::true_type{}, pa[2], pb[2], pa[3], pb[3]);
        step(s0 + 3, B1{}, std::true_type{}, pa[3], pb[3], pa[0], pb[0]);
    }
    // tail (nslices % NSET steps): everything they need is already in the sets
    if (s0 < nslices) step(s0, B0{}, std::false_type{}, pa[0], pb[0], pa[1], pb[1]);
    if (s0 + 1 < nslices) step(s0 + 1, B1{}, std::false_type{}, pa[1], pb[1], pa[2], pb[2]);
    if (s0 + 2 < nslices) step(s0 + 2, B0{}, std::false_type{}, pa[2], pb[2], pa[3], pb[3]);
    // (Tried on top of this and not kept, round 5: a three-buffer LDS ring with the next slice's fragments read and the slice after it
    // handed over INSIDE the chain, one piece of each behind every pair of products, pinned with sched_barrier -- the ISA was exactly the
    // weave intended: stage 4's 3x3 43.5 -> 38.2 us, but the short-K launches 8.6 -> 11-12 us (a longer prologue), B = 1 the same 0.89 ms,
    // and 32-64 patches 8-10 % SLOWER (101 KB of LDS: one block per CU instead of two).  Six sets of prefetch instead of four: no gain
    // either -- a 128-k slice stays at ~1.05 us, so what is left is the chain itself plus the barrier, not memory latency.)
    // ---- epilogue: C[row = 4 kg + i][col = r16] of the wave's 16 x 16 sub-tile
    if (!col_ok) return;
    const float lo = a.act == ADAF_ACT_NONE ? -__builtin_inff() : 0.f;
    const float hi = a.act == ADAF_ACT_RELU6 ? 6.f : __builtin_inff();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int mo = m0 + wm * 16 + 4 * kg + i;
        if (mo >= a.M) continue;
        a.out[(size_t)mo * a.ldo + n] = fminf(fmaxf(fmaf(acc[i], sc, bi) + rv[i], lo), hi);
    }
}

}  // namespace

// 1 = launched, 0 = the shape is not the latency kernel's (the caller uses the engine)
int adaf_launch_conv_lat(const ConvArgs& a, hipStream_t s) {
    if (a.in16 || a.out16 || a.res16 || a.split_n || a.tsm_T > 0) return 0;
    if (a.act != ADAF_ACT_NONE && a.act != ADAF_ACT_RELU && a.act != ADAF_ACT_RELU6) return 0;
    if (a.cin % 64 || a.K != a.KH * a.KW * a.cin || a.KH * a.KW > 32 || (a.ldx & 3)) return 0;
    if ((reinterpret_cast<size_t>(a.x) | reinterpret_cast<size_t>(a.w)) & 15) return 0;
    const long long blocks = (long long)((a.M + 31) / 32) * ((a.N + 31) / 32);
    if (blocks <= 0 || blocks > (1ll << 30)) return 0;
    if (a.cin % 128 == 0) hipLaunchKernelGGL((conv_lat_kernel<128>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((conv_lat_kernel<64>), dim3((unsigned)blocks), dim3(256), 0, s, a);
    return 1;
}
