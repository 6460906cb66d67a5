// Whole-image MBConv blocks of EfficientNet (BASELINE config 5, fp16 storage): expand 1x1 + BN + swish -> depthwise k x k + BN + swish
// -> squeeze-and-excite -> gated project 1x1 + BN (+ identity) as ONE launch for the blocks whose map is small enough that a
// workgroup owns whole images (9 x 9 and 5 x 5 at 144^2 patches: 16 of B3's 26 blocks: 9-17 and 19-24 at stride 1, block 18 -- 9 x 9 -> 5 x 5 --
// at stride 2; block 25's 2304 hidden channels exceed the 2048 the kernel walks).  The 6x-expanded map never exists in HBM
// (it does not even exist in LDS: it goes from the MFMA accumulators straight into the depthwise taps), the depthwise output
// lives in LDS only, the squeeze is an in-block reduction and the two SE matrix products run inside the block -- four launches
// and three HBM round trips of the widest tensors of the block become one launch that reads the block input and writes the block
// output.  PARITY UNPINNED like the rest of config 5 (the reference holds no EfficientNet: effnet.hip's header); the algorithm is
// model.py MBConvBlock.forward of `efficientnet_pytorch` as restated by oracle/ref_effnet.py:mbconv, and every value is produced by
// the arithmetic of the four-launch plan of effnet.hip (same MFMA instruction and k order, same BN / swish expressions, the same
// fp16 roundings of the expanded map, the depthwise output and the gated operand); only the order of the squeeze's fp32 sum differs.
//
// Phases of a workgroup (8 waves, G images; NB = G x ceil(HW^2 / 32) row bands of 32 pixels):
//   0  the block input X (G x HW^2 rows x cin halfs) -> LDS, k padded to whole MFMA steps with zeros
//   1  per PAIR of 32-channel tiles of the hidden dimension, one wave: expand GEMM (v_mfma_f32_32x32x16_f16, A fragments from X,
//      B fragments streamed from L2 in fragment order: one coalesced 1 KB load per wave instruction), BN + swish + fp16 rounding
//      on the accumulators, then v_permlane32_swap hands lane l ALL pixels of channel 64 pair + l (the two tiles' half-rows
//      trade places), and the k x k depthwise conv runs out of registers with every index resolved at compile time -- only the
//      taps inside the map are visited (a skipped tap adds an exact zero) -- + BN + swish; the outputs go to D[image][pixel][hid]
//      in LDS as fp16, the squeeze sums stay in the lane
//   2  squeeze -> reduce FC + swish -> expand FC + sigmoid (the arithmetic of se_gate_kernel), gate in LDS
//   3  D *= gate in place (fp32 product rounded to fp16: the operand gated_project_kernel hands its MFMAs)
//   4  project GEMM: A fragments from D, B fragments streamed like phase 1, BN (+ identity) epilogue, fp16 stores
//
// Two rules of this file, both measured:
//   * NO SCRATCH.  A spilled register is an HBM write, and every later s_waitcnt vmcnt of the wave waits for its acknowledgement: with
//     46 spilled dwords the kernel's L2 round trips measured 8-10 k cycles instead of ~800 (tools/mbw_trace.py, tools/exp/l2_burst_bench.hip).
//   * NO GLOBAL LOAD INTO A REGISTER OF AN MFMA THAT MAY STILL BE IN FLIGHT -- neither one it reads (B fragments loaded straight into the
//     set the previous chunk's products used) nor one it writes (loads that the allocator placed in accumulator registers it knew to be
//     dead): results then differed from run to run in a few hundred values per forward.  B fragments land in sets no MFMA reads and move
//     through VALU copies; loads behind a K loop are issued after a VALU instruction has read the last MFMA's result, pinned by
//     scheduling barriers (tools/exp/effnet_determinism.py, tests/test_effnet.py::test_b3_fp16_forward_is_the_same_from_run_to_run).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "adaf_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int kMbwThreads = 512;
constexpr int kMbwWaves = kMbwThreads / 64;
constexpr int kMbwMaxRounds = 4;          // channel pairs per wave: hid <= 64 * 8 * 4 = 2048 (= 4 x 512 threads in the SE expand FC)

__device__ __forceinline__ float w_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f)); }
// two values at a time on the packed fp32 instructions (v_pk_mul / v_pk_add; a plain VALU instruction of a wave takes ~4 cycles on
// gfx950 and the vector part of this kernel is what bounds it): the same operations in the same order as v * w_sigmoid(v)
__device__ __forceinline__ f32x2 w_swish2(f32x2 t) {
    f32x2 e = t * -1.4426950408889634f;
    e.x = __builtin_amdgcn_exp2f(e.x); e.y = __builtin_amdgcn_exp2f(e.y);
    e = 1.f + e;
    e.x = __builtin_amdgcn_rcpf(e.x); e.y = __builtin_amdgcn_rcpf(e.y);
    return t * e;
}
__device__ __forceinline__ f32x2 w_fma2(f32x2 a, float s, float b) { return __builtin_elementwise_fma(a, f32x2{s, s}, f32x2{b, b}); }

// rows / columns of SAME padding in front of the map (TensorFlow's rule: the odd one falls behind)
constexpr int mbw_pad_before(int hw, int k, int s) {
    const int o = (hw + s - 1) / s, tot = (o - 1) * s + k - hw;
    return tot > 0 ? tot / 2 : 0;
}

struct MbwArgs {
    const _Float16* x;       // [n][HW*HW][cin] block input
    const _Float16* res;     // identity rows (= x) or nullptr
    _Float16* out;           // [n][HW*HW][cout]
    const u32x4* wef;        // expand filter in B-fragment order [NT2][KS][64] x 16 B (NT2 = tiles rounded up to even; zero padded)
    const float* se;         // expand BN [hid]
    const float* be;
    const float* wdl;        // depthwise operands, one row per channel [hid][TP]: K*K taps, BN scale, BN bias, zeros (TP = 12 / 28)
    const float* se_wr;      // [sq][hid]
    const float* se_br;      // [sq]
    const float* se_wet;     // [sq][hid] (transposed _se_expand.weight)
    const float* se_be;      // [hid]
    const u32x4* wpf;        // project filter in B-fragment order [NTP][KSP][64] x 16 B
    const float* sp;         // project BN [cout]
    const float* bp;
    int n, cin, hid, cout, sq;
    int KS, KSP, NTP, NPAIR; // expand k steps (of 16), project k steps, project column tiles, channel pairs (of 64)
    int KSPP;                // project k steps padded to whole chunks of 6 (the packed filter carries zero fragments there)
    int xpitch, dpitch;      // LDS row pitches in bytes
    int d_off;               // byte offset of D in the dynamic LDS (X and, later, mean / gate / squeezed vector come first)
#ifdef MBW_TRACE
    unsigned long long* trace;   // [blocks][waves][16] s_memtime stamps (tools/mbw_trace.py; never compiled into the shipped library)
#endif
};

// Trace build (tools/exp/build_mbw_trace.sh, -DMBW_TRACE): lane 0 of every wave stamps s_memtime at the phase boundaries.
#ifdef MBW_TRACE
#define MBW_STAMP(slot_) do { if (a.trace && lane == 0) a.trace[((size_t)blockIdx.x * kMbwWaves + wave) * 16 + (slot_)] = __builtin_readcyclecounter(); } while (0)
#else
#define MBW_STAMP(slot_) do { } while (0)
#endif

// S = 2 (the block that takes a 9 x 9 map to 5 x 5): the depthwise part visits the HWO x HWO outputs only, D / squeeze / gate / project
// run on those PXO rows, no identity.
template <int HW, int K, int G, int S = 1>
__global__ __launch_bounds__(kMbwThreads) void mbconv_whole_kernel(const MbwArgs a) {
    static_assert(S == 1 || (S == 2 && G == 1), "stride 2: one image per workgroup");
    constexpr int PX = HW * HW;
    constexpr int RB = (PX + 31) / 32;         // row bands per image (input map: the expand GEMM)
    constexpr int NB = G * RB;
    constexpr int HWO = (HW + S - 1) / S, PXO = HWO * HWO;      // output map (D, squeeze, project)
    constexpr int RBO = (PXO + 31) / 32, NBO = G * RBO;
    constexpr int P = mbw_pad_before(HW, K, S);                 // SAME padding: symmetric at stride 1
    constexpr int KC = 3;                      // k steps per prefetched chunk of B fragments (expand)
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nl = lane & 31, half = lane >> 5;
    const int img0 = blockIdx.x * G;
    const int nimg = min(G, a.n - img0);
    char* xl = dsm;
    char* dl = dsm + a.d_off;
    const int hid = a.hid;
    MBW_STAMP(0);

    // ---- phase 0: X -> LDS ----
    // (all global loads of this kernel are UNCONDITIONAL loads from clamped addresses, in straight-line batches: hipcc puts an
    // s_waitcnt vmcnt(0) in front of a load that sits in its own branch, which serialises a batch into a chain of L2 round trips --
    // measured: the residual loads of phase 4 and the filter rows of phase 2 ran 4-5x slower that way)
    {
        const int cpr = a.cin >> 3, cpp = a.KS * 2;
        const int rows = nimg * PX;
        const int total = G * PX * cpp;
        const _Float16* xb = a.x + (size_t)img0 * PX * a.cin;
        for (int i0 = tid; i0 < total; i0 += 4 * kMbwThreads) {
            u32x4 v[4];
            int r[4], c[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * kMbwThreads;
                r[u] = i / cpp; c[u] = i - r[u] * cpp;
                const bool ok = i < total && r[u] < rows && c[u] < cpr;
                v[u] = *reinterpret_cast<const u32x4*>(xb + (ok ? (size_t)r[u] * a.cin + c[u] * 8 : 0));
                if (!ok) v[u] = u32x4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i0 + u * kMbwThreads < total) *reinterpret_cast<u32x4*>(xl + r[u] * a.xpitch + c[u] * 16) = v[u];
        }
    }
    __syncthreads();
    MBW_STAMP(1);

    // row of band b this lane feeds the MFMAs with (rows past the last pixel repeat the last row: their results are never used)
    int xoff[NB], doff[NBO];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        int row = (b / RB) * PX + (b % RB) * 32 + nl;
        row = row < G * PX ? row : G * PX - 1;
        xoff[b] = row * a.xpitch + half * 16;
    }
#pragma unroll
    for (int b = 0; b < NBO; ++b) {
        int row = (b / RBO) * PXO + (b % RBO) * 32 + nl;
        row = row < G * PXO ? row : G * PXO - 1;
        doff[b] = row * a.dpitch + half * 16;
    }

    // ---- phase 1: expand -> depthwise, a pair of 32-channel tiles per wave iteration ----
    float psr[kMbwMaxRounds][G];
#pragma unroll
    for (int q = 0; q < kMbwMaxRounds; ++q)
#pragma unroll
        for (int g = 0; g < G; ++g) psr[q][g] = 0.f;
    {
        const int KS = a.KS;
        // (KS is a whole number of chunks: plan_mbw pads it, the packed filter and the LDS rows carry zeros there)
        auto load_b = [&](u32x4 (&dst)[2][KC], int jp, int k0) {
            const u32x4* p0 = a.wef + ((size_t)(2 * jp) * KS + k0) * 64 + lane;
            const u32x4* p1 = p0 + (size_t)KS * 64;
#pragma unroll
            for (int u = 0; u < KC; ++u) { dst[0][u] = p0[u * 64]; dst[1][u] = p1[u * 64]; }
        };
        // Every workgroup of the launch streams the same filters; they start together, so without a stagger all CUs would ask the L2
        // for the same lines at the same time.  Workgroup b walks the channel pairs starting at pair `rot`.
        const int rot = (int)((blockIdx.x * 5u) % (unsigned)a.NPAIR);
        auto pair_of = [&](int li) { const int j = li + rot; return j >= a.NPAIR ? j - a.NPAIR : j; };
        // B fragments: ONE set feeds the MFMAs (bcur); global loads land in two other sets (bl0 / bl1, alternating) and a chunk reaches
        // bcur through a VALU copy after the products of the chunk before it.  So a chunk is requested two chunks ahead of its use
        // (a chunk's products are shorter than an L2 round trip), and NO LOAD EVER TARGETS A REGISTER THAT AN MFMA READS.  The
        // obvious form -- three sets with rotating roles, loads straight into the set the previous chunk's MFMAs have just read --
        // gave results that differed from run to run on gfx950 (tools/exp/effnet_determinism.py; the compiler's waits were correct on
        // every path): with several independent accumulator chains a wave's MFMAs are still queued when a load issued behind them
        // returns.  VALU writes to MFMA operands are interlocked; returning loads, measurably, are not.
        const int nchunk = KS / KC;
        u32x4 bcur[2][KC], bl0[2][KC], bl1[2][KC];
        auto load_pair_head = [&](int jp) { load_b(bl0, jp, 0); if (nchunk > 1) load_b(bl1, jp, KC); };
        auto take = [&](const u32x4 (&src)[2][KC]) {
#pragma unroll
            for (int u = 0; u < KC; ++u) { bcur[0][u] = src[0][u]; bcur[1][u] = src[1][u]; }
        };
        if (wave < a.NPAIR) load_pair_head(pair_of(wave));
        int rd = 0;
        for (int li = wave; li < a.NPAIR; li += kMbwWaves, ++rd) {
            const int jp = pair_of(li);
            const int c = 64 * jp + lane;            // the channel this lane owns in the depthwise part
            const bool cok = c < hid;
            const int cc = cok ? c : 0;
            // BN of the expanded channels: requested now, needed right after the products
            const int ce0 = 64 * jp + nl, ce1 = ce0 + 32;      // the channels of this lane's accumulator columns
            const float sce0 = ce0 < hid ? a.se[ce0] : 0.f, bie0 = ce0 < hid ? a.be[ce0] : 0.f;
            const float sce1 = ce1 < hid ? a.se[ce1] : 0.f, bie1 = ce1 < hid ? a.be[ce1] : 0.f;

            f32x16 acc[2][NB];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int i = 0; i < 16; ++i) acc[t][b][i] = 0.f;
            // K loop in chunks of KC steps, branch-free inside a chunk: every A fragment of the chunk is requested from LDS before its
            // first product (a read -> wait -> product sequence per step leaves the matrix pipe idle for an LDS round trip each time)
            auto mma_chunk = [&](int k0, const u32x4 (&bb)[2][KC]) {
                f16x8 af[KC][NB];
#pragma unroll
                for (int u = 0; u < KC; ++u)
#pragma unroll
                    for (int b = 0; b < NB; ++b) af[u][b] = *reinterpret_cast<const f16x8*>(xl + xoff[b] + (k0 + u) * 32);
#pragma unroll
                for (int u = 0; u < KC; ++u)
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[u][b], __builtin_bit_cast(f16x8, bb[0][u]), acc[0][b], 0, 0, 0);
                        acc[1][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[u][b], __builtin_bit_cast(f16x8, bb[1][u]), acc[1][b], 0, 0, 0);
                    }
            };
            take(bl0);
            for (int c = 0; c < nchunk; c += 2) {
                if (c + 2 < nchunk) load_b(bl0, jp, (c + 2) * KC);
                mma_chunk(c * KC, bcur);
                if (c + 1 < nchunk) {
                    take(bl1);
                    if (c + 3 < nchunk) load_b(bl1, jp, (c + 3) * KC);
                    mma_chunk((c + 1) * KC, bcur);
                    if (c + 2 < nchunk) take(bl0);
                }
            }
            if (rd < 2) MBW_STAMP(2 + 3 * rd);
            // BN + swish on the accumulators, rounded to the storage type (what the expand launch would have written) -- LAST band
            // first: its first swish reads the result of the last MFMA issued, and MFMAs complete in order, so every load requested
            // behind that point finds no MFMA in flight.  Requested there, pinned by scheduling barriers (hipcc hoists loads as far
            // up as it can, and hands them the registers the products have just freed):
            //  * the next pair's first two chunks of B fragments, which travel under the vector part;
            //  * the operands of the depthwise part (the channel's row: taps, BN scale, BN bias in TP / 4 16-byte loads) -- not
            //    before the products: 27 registers that are live across the K loop are the difference between spilling and not
            //    spilling, and a spill to scratch is an HBM write that every later s_waitcnt vmcnt of the wave waits for.
            constexpr int TP = (K * K + 2 + 3) / 4 * 4;
            float w[TP];
#pragma unroll
            for (int bb = 0; bb < NB; ++bb) {
                const int b = NB - 1 - bb;
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    // (registers i, i + 1 of band b hold rows 32 (b % RB) + (i & 3) + 8 (i >> 2) [+ 1] (+ 4 in the upper half-wave):
                    // a pair whose first row is past the image is never read by the depthwise part)
                    if ((b % RB) * 32 + (i & 3) + 8 * (i >> 2) >= PX) continue;
                    const f32x2 t0 = w_swish2(w_fma2(f32x2{acc[0][b][i], acc[0][b][i + 1]}, sce0, bie0));
                    const f32x2 t1 = w_swish2(w_fma2(f32x2{acc[1][b][i], acc[1][b][i + 1]}, sce1, bie1));
                    acc[0][b][i] = (float)adaf_f16_of(t0.x); acc[0][b][i + 1] = (float)adaf_f16_of(t0.y);
                    acc[1][b][i] = (float)adaf_f16_of(t1.x); acc[1][b][i + 1] = (float)adaf_f16_of(t1.y);
                }
                if (bb == 0) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (li + kMbwWaves < a.NPAIR) load_pair_head(pair_of(li + kMbwWaves));
                    const f32x4* wrow = reinterpret_cast<const f32x4*>(a.wdl + (size_t)cc * TP);
#pragma unroll
                    for (int q = 0; q < TP / 4; ++q) {
                        const f32x4 v = wrow[q];
                        w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            const float sdl = w[K * K], bdl = w[K * K + 1];
            // lanes 0-31 take tile 0's other half-rows, lanes 32-63 give them and take tile 1's: afterwards acc[h][b][i] of lane l is
            // row 32 b + (i & 3) + 8 (i >> 2) + 4 h of channel 64 jp + l
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if ((b % RB) * 32 + (i & 3) + 8 * (i >> 2) >= PX) continue;
                    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[0][b][i]), __float_as_uint(acc[1][b][i]), false, false);
                    acc[0][b][i] = __uint_as_float(r[0]);
                    acc[1][b][i] = __uint_as_float(r[1]);
                }
            if (rd < 2) MBW_STAMP(3 + 3 * rd);
            // depthwise k x k + BN + swish out of registers: the map value of pixel p of image g
#define MBW_M(g_, p_) acc[(((p_) & 31) >> 2) & 1][(g_) * RB + ((p_) >> 5)][((p_) & 3) + 4 * (((p_) & 31) >> 3)]
            if constexpr (G == 2) {
                // two images per workgroup: the packed lanes are the two images (same channel, same taps)
                f32x2 ps = {0.f, 0.f};
                unsigned dofs = (unsigned)a.d_off + (unsigned)cc * 2u;
                const unsigned gofs = (unsigned)(PX * a.dpitch);
#pragma unroll
                for (int oy = 0; oy < HW; ++oy) {
                    f32x2 s[HW];
#pragma unroll
                    for (int ox = 0; ox < HW; ++ox) s[ox] = f32x2{0.f, 0.f};
#pragma unroll
                    for (int ky = 0; ky < K; ++ky)
#pragma unroll
                        for (int kx = 0; kx < K; ++kx)
#pragma unroll
                            for (int ox = 0; ox < HW; ++ox) {
                                const int iy = oy + ky - P, ix = ox + kx - P;
                                if (iy < 0 || iy >= HW || ix < 0 || ix >= HW) continue;      // resolved at compile time
                                const int p = iy * HW + ix;
                                s[ox] = __builtin_elementwise_fma(f32x2{MBW_M(0, p), MBW_M(1, p)}, f32x2{w[ky * K + kx], w[ky * K + kx]}, s[ox]);
                            }
#pragma unroll
                    for (int ox = 0; ox < HW; ++ox) {
                        const f32x2 v = w_swish2(w_fma2(s[ox], sdl, bdl));
                        ps += v;
                        if (cok) {
                            *reinterpret_cast<_Float16*>(dsm + dofs) = (_Float16)v.x;
                            if (nimg > 1) *reinterpret_cast<_Float16*>(dsm + dofs + gofs) = (_Float16)v.y;
                        }
                        dofs += (unsigned)a.dpitch;
                    }
                }
#pragma unroll
                for (int q = 0; q < kMbwMaxRounds; ++q) { psr[q][0] = q == rd ? ps.x : psr[q][0]; psr[q][1] = q == rd ? ps.y : psr[q][1]; }
            } else {
                // one image per workgroup: one product per instruction, in (ky, kx) order like the four-launch plan.  (Measured on
                // gfx950 with two waves per SIMD: v_fma_f32 issues every ~3 cycles, v_pk_fma_f32 every ~5.3 -- pairing two outputs per
                // instruction through two partial sums per output bought nothing and gave up the summation order; tools/exp/valu_rate_bench.hip)
                f32x2 ps = {0.f, 0.f};
                unsigned dofs = (unsigned)a.d_off + (unsigned)cc * 2u;
#pragma unroll
                for (int oy = 0; oy < HWO; ++oy) {
                    float s[HWO + 1];
#pragma unroll
                    for (int ox = 0; ox <= HWO; ++ox) s[ox] = 0.f;
#pragma unroll
                    for (int ky = 0; ky < K; ++ky)
#pragma unroll
                        for (int kx = 0; kx < K; ++kx)
#pragma unroll
                            for (int ox = 0; ox < HWO; ++ox) {
                                const int iy = oy * S + ky - P, ix = ox * S + kx - P;
                                if (iy < 0 || iy >= HW || ix < 0 || ix >= HW) continue;      // resolved at compile time
                                s[ox] = fmaf(MBW_M(0, iy * HW + ix), w[ky * K + kx], s[ox]);
                            }
                    // BN + swish two outputs at a time (the last pair of an odd row carries a dummy)
#pragma unroll
                    for (int ox = 0; ox < HWO; ox += 2) {
                        const f32x2 v = w_swish2(w_fma2(f32x2{s[ox], s[ox + 1]}, sdl, bdl));
                        if (ox + 1 < HWO) ps += v; else ps.x += v.x;
                        if (cok) {
                            *reinterpret_cast<_Float16*>(dsm + dofs) = (_Float16)v.x;
                            if (ox + 1 < HWO) *reinterpret_cast<_Float16*>(dsm + dofs + a.dpitch) = (_Float16)v.y;
                        }
                        dofs += (ox + 1 < HWO ? 2u : 1u) * (unsigned)a.dpitch;
                    }
                }
                const float pst = ps.x + ps.y;
#pragma unroll
                for (int q = 0; q < kMbwMaxRounds; ++q) psr[q][0] = q == rd ? pst : psr[q][0];
            }
#undef MBW_M
            if (rd < 2) MBW_STAMP(4 + 3 * rd);
        }
    }
    MBW_STAMP(8);
    // ---- phase 2: squeeze-and-excite (the arithmetic of se_gate_kernel) ----
    // The filter rows come from L2 and do not depend on this block's data, and every workgroup of the launch streams all of them:
    // the phase is bound by how many bytes a CU keeps in flight.  A batch is SEB rows per wave with every 16-byte piece requested
    // before the first is used, the first batch is requested BEFORE the barrier that closes phase 1 (a wave that ran out of channel
    // pairs waits there anyway), and the expand FC's first rows are requested before the barrier that closes the reduce FC.
    // (rows per wave per batch of the reduce FC and rows per batch of the expand FC are chosen with the piece count, below: B3's
    //  squeeze widths -- 24 / 34 / 58 rows -- then take ONE batch of the reduce FC, and 34 rows one batch of the expand FC; a
    //  second batch of two rows is a whole extra round trip)
    const int C4 = hid >> 2;
    const int SQ = a.sq;
    float* mean = reinterpret_cast<float*>(xl);            // [G][hid], later the gate
    float* sqv = mean + G * hid;                            // [G][sq]
    // reduce FC for NP 16-byte pieces of a filter row per lane (NP = ceil(hid / 256): a compile-time count keeps the batch straight-line)
    auto se_reduce = [&](auto np_tag) {
        constexpr int NP = decltype(np_tag)::value;
        constexpr int SEB = NP == 4 ? 5 : NP == 6 ? 8 : 4;          // rows per wave per batch: SEB * NP 16-byte pieces in flight per lane
        f32x4 wv[SEB][NP];
        float brv[SEB];            // the rows' biases travel with them (as a load inside `if (lane == 0)` each was a round trip of its own)
        auto se_load = [&](int j0) {
#pragma unroll
            for (int r = 0; r < SEB; ++r) {
                const int j = j0 + r * kMbwWaves < SQ ? j0 + r * kMbwWaves : SQ - 1;
                brv[r] = a.se_br[j];
                const float* wrow = a.se_wr + (size_t)j * hid + 4 * (lane < C4 ? lane : 0);      // (hid < 256: lanes past the row read lane 0's piece, never used)
#pragma unroll
                for (int q = 0; q < NP; ++q) wv[r][q] = *reinterpret_cast<const f32x4*>(wrow + (lane + 64 * q < C4 ? 256 * q : 0));
            }
        };
        se_load(wave);
        __syncthreads();           // X is dead, D is complete
        MBW_STAMP(9);
        {
            const float inv_hw = 1.f / (float)PXO;
            const int rot = (int)((blockIdx.x * 5u) % (unsigned)a.NPAIR);
#pragma unroll
            for (int q = 0; q < kMbwMaxRounds; ++q) {
                const int li = wave + q * kMbwWaves;
                const int jp = li + rot >= a.NPAIR ? li + rot - a.NPAIR : li + rot;
                const int c = 64 * jp + lane;
                if (li < a.NPAIR && c < hid) {
#pragma unroll
                    for (int g = 0; g < G; ++g) mean[g * hid + c] = psr[q][g] * inv_hw;
                }
            }
        }
        __syncthreads();
        for (int j0 = wave; j0 < SQ; j0 += SEB * kMbwWaves) {
            if (j0 != wave) se_load(j0);
#pragma unroll
            for (int r = 0; r < SEB; ++r) {
                const int j = j0 + r * kMbwWaves;
                float sg[G];
#pragma unroll
                for (int g = 0; g < G; ++g) sg[g] = 0.f;
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    const int c4 = lane + 64 * q;
                    if (c4 < C4) {
#pragma unroll
                        for (int g = 0; g < G; ++g) {
                            const f32x4 m = *reinterpret_cast<const f32x4*>(mean + g * hid + 4 * c4);
                            sg[g] = fmaf(m.x, wv[r][q].x, fmaf(m.y, wv[r][q].y, fmaf(m.z, wv[r][q].z, fmaf(m.w, wv[r][q].w, sg[g]))));
                        }
                    }
                }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    sg[g] = adaf_wave_sum(sg[g]);
                    if (lane == 0 && j < SQ) {
                        const float v = sg[g] + brv[r];
                        sqv[g * SQ + j] = v * w_sigmoid(v);
                    }
                }
            }
        }
    };
    {
        const int np = (C4 + 63) >> 6;
        if (np <= 2) se_reduce(std::integral_constant<int, 2>());
        else if (np == 3) se_reduce(std::integral_constant<int, 3>());
        else if (np == 4) se_reduce(std::integral_constant<int, 4>());
        else if (np <= 6) se_reduce(std::integral_constant<int, 6>());
        else se_reduce(std::integral_constant<int, 8>());
    }
    auto se_expand = [&](auto gjb_tag) {
        // expand FC: a thread owns 4 consecutive channels; rows in batches of GJB with every load of a batch in flight at once
        constexpr int GJB = decltype(gjb_tag)::value;
        const int c4 = tid < C4 ? tid : 0;            // (C4 <= 512; threads past the end repeat piece 0 and store nothing)
        const float* wp = a.se_wet + 4 * c4;
        f32x4 wj[GJB];
        auto gate_load = [&](int j0) {
#pragma unroll
            for (int u = 0; u < GJB; ++u) wj[u] = *reinterpret_cast<const f32x4*>(wp + (size_t)(j0 + u < SQ ? j0 + u : SQ - 1) * hid);
        };
        gate_load(0);
        MBW_STAMP(10);
        __syncthreads();
        {
            f32x4 sg[G];
            const f32x4 b = *reinterpret_cast<const f32x4*>(a.se_be + 4 * c4);
#pragma unroll
            for (int g = 0; g < G; ++g) sg[g] = b;
            for (int j0 = 0; j0 < SQ; j0 += GJB) {
                if (j0) gate_load(j0);
#pragma unroll
                for (int u = 0; u < GJB; ++u)
                    if (j0 + u < SQ) {
#pragma unroll
                        for (int g = 0; g < G; ++g) {          // (explicit fmaf: the contraction of `sg += w * q` is the compiler's choice per image)
                            const float q = sqv[g * SQ + j0 + u];
                            sg[g].x = fmaf(wj[u].x, q, sg[g].x); sg[g].y = fmaf(wj[u].y, q, sg[g].y);
                            sg[g].z = fmaf(wj[u].z, q, sg[g].z); sg[g].w = fmaf(wj[u].w, q, sg[g].w);
                        }
                    }
            }
            if (tid < C4) {
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const f32x4 o = {w_sigmoid(sg[g].x), w_sigmoid(sg[g].y), w_sigmoid(sg[g].z), w_sigmoid(sg[g].w)};
                    *reinterpret_cast<f32x4*>(mean + g * hid + 4 * c4) = o;      // (every thread is past its reads of `mean`: the barrier above)
                }
            }
        }
    };
    if (SQ > 32 && SQ <= 36) se_expand(std::integral_constant<int, 36>());      // (B3's 34 rows: one batch)
    else se_expand(std::integral_constant<int, 32>());
    MBW_STAMP(11);
    __syncthreads();
    MBW_STAMP(12);

    // ---- the first two chunks of filter fragments of this wave's first column tile of phase 4 are requested HERE, ahead of phase 3:
    // they do not depend on the gate, and phase 3 + its barrier is longer than their round trip ----
    constexpr int KCP = 6;                     // k steps per chunk of B fragments (project); three chunks deep
    const int KSP = a.KSP, KSPP = a.KSPP;      // real k steps / padded to whole chunks (zero fragments in the packed filter)
    const _Float16* __restrict__ resb = a.res ? a.res + (size_t)img0 * PXO * a.cout : nullptr;
    _Float16* __restrict__ outb = a.out + (size_t)img0 * PXO * a.cout;
    const int rotp = (int)((blockIdx.x * 3u) % (unsigned)a.NTP);
    u32x4 pcur[KCP], pl0[KCP], pl1[KCP];       // the MFMAs' set and two landing sets, like the expand loop
    auto tile_of = [&](int lt) { return lt + rotp >= a.NTP ? lt + rotp - a.NTP : lt + rotp; };
    auto load_p = [&](u32x4 (&dst)[KCP], int nt, int k0) {
        const u32x4* bp0 = a.wpf + (size_t)nt * KSPP * 64 + lane;
#pragma unroll
        for (int u = 0; u < KCP; ++u) dst[u] = bp0[(k0 + u < KSPP ? k0 + u : KSPP - 1) * 64];
    };
    auto issue_tile = [&](int lt) {
        const int nt = tile_of(lt);
        load_p(pl0, nt, 0);
        load_p(pl1, nt, KCP);
    };
    if (wave < a.NTP) issue_tile(wave);

    // ---- phase 3: D *= gate (fp32 product, rounded to fp16) ----
    // A thread keeps ONE 16-byte column chunk and walks down the rows, its eight gate values in registers (read once per image): the
    // phase is LDS traffic, and a flat walk over (row, chunk) read 32 bytes of gate for every 16 bytes of D.
    {
        const int cpr = hid >> 3;                     // 16-byte chunks per row
        const int rpar = kMbwThreads / cpr;           // rows in flight (cpr <= 256: hid <= 2048)
        const int r0 = tid / cpr, cq = tid - r0 * cpr;
        if (r0 < rpar) {
#pragma unroll
            for (int g = 0; g < G; ++g) {
                if (g >= nimg) break;
                const float* gp = mean + g * hid + cq * 8;
                const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp), g1 = *reinterpret_cast<const f32x4*>(gp + 4);
                char* p = dl + (g * PXO + r0) * a.dpitch + cq * 16;
                const int pstep = rpar * a.dpitch;
                for (int row = r0; row < PXO; row += rpar, p += pstep) {
                    const f16x8 v = *reinterpret_cast<const f16x8*>(p);
                    f16x8 o;
                    o[0] = adaf_f16_of((float)v[0] * g0.x); o[1] = adaf_f16_of((float)v[1] * g0.y);
                    o[2] = adaf_f16_of((float)v[2] * g0.z); o[3] = adaf_f16_of((float)v[3] * g0.w);
                    o[4] = adaf_f16_of((float)v[4] * g1.x); o[5] = adaf_f16_of((float)v[5] * g1.y);
                    o[6] = adaf_f16_of((float)v[6] * g1.z); o[7] = adaf_f16_of((float)v[7] * g1.w);
                    *reinterpret_cast<f16x8*>(p) = o;
                }
            }
        }
    }
    MBW_STAMP(13);
    __syncthreads();
    MBW_STAMP(14);

    // ---- phase 4: project 1x1 + BN (+ identity), one 32-column tile per wave iteration ----
    // K loop in chunks of KCP steps, a chunk requested two chunks ahead of its products (they are shorter than an L2 round trip);
    // the padding steps multiply real (finite) D columns by zero fragments
    for (int lt = wave; lt < a.NTP; lt += kMbwWaves) {
        const int nt = tile_of(lt);
        const int ncol = nt * 32 + nl;
        const bool nok = ncol < a.cout;
        const int obase = 4 * half * a.cout + (nok ? ncol : 0);
        // the identity rows of this tile travel under the products (one 2-byte load per output of the lane; unconditional loads from
        // clamped addresses -- a load inside a per-lane branch is followed by its own s_waitcnt, and 41 of those in a row were most of
        // this phase's time; values of pixels that do not exist are never stored)
        _Float16 rv[NBO][16];
        if (resb) {
#pragma unroll
            for (int b = 0; b < NBO; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int pr = (b % RBO) * 32 + (i & 3) + 8 * (i >> 2);          // pixel of the h = 0 lanes; h = 1: + 4
                    const bool ok = b / RBO < nimg && pr + 4 * half < PXO;
                    rv[b][i] = resb[ok ? obase + ((b / RBO) * PXO + pr) * a.cout : 0];
                }
        } else {
#pragma unroll
            for (int b = 0; b < NBO; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) rv[b][i] = (_Float16)0.f;
        }
        f32x16 acc[NBO];
#pragma unroll
        for (int b = 0; b < NBO; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
        const int npc = KSPP / KCP;
        auto mma_p = [&](int k0, const u32x4 (&bb)[KCP]) {
#pragma unroll
            for (int h = 0; h < KCP; h += 3) {
                f16x8 af[3][NBO];
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int kk = k0 + h + u < KSP ? k0 + h + u : KSP - 1;
#pragma unroll
                    for (int b = 0; b < NBO; ++b) af[u][b] = *reinterpret_cast<const f16x8*>(dl + doff[b] + kk * 32);
                }
#pragma unroll
                for (int u = 0; u < 3; ++u)
#pragma unroll
                    for (int b = 0; b < NBO; ++b)
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[u][b], __builtin_bit_cast(f16x8, bb[h + u]), acc[b], 0, 0, 0);
            }
        };
        auto ptake = [&](const u32x4 (&src)[KCP]) {
#pragma unroll
            for (int u = 0; u < KCP; ++u) pcur[u] = src[u];
        };
        ptake(pl0);
        for (int c = 0; c < npc; c += 2) {
            if (c + 2 < npc) load_p(pl0, nt, (c + 2) * KCP);
            mma_p(c * KCP, pcur);
            if (c + 1 < npc) {
                ptake(pl1);
                if (c + 3 < npc) load_p(pl1, nt, (c + 3) * KCP);
                mma_p((c + 1) * KCP, pcur);
                if (c + 2 < npc) ptake(pl0);
            }
        }
        if (nok) {
            const float sc = a.sp[ncol], bi = a.bp[ncol];
#pragma unroll
            for (int b = 0; b < NBO; ++b) {
                if (b / RBO >= nimg) continue;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int pr = (b % RBO) * 32 + (i & 3) + 8 * (i >> 2);
                    if (pr + 4 * half < PXO) outb[obase + ((b / RBO) * PXO + pr) * a.cout] = adaf_f16_of(fmaf(acc[b][i], sc, bi) + (float)rv[b][i]);
                }
            }
        }
        if (lt + kMbwWaves < a.NTP) issue_tile(lt + kMbwWaves);
    }
    MBW_STAMP(15);
}

// [N][K] fp32 (a 1x1 conv's OIHW filter) -> the B operand of v_mfma_f32_32x32x16_f16 in fragment order, fp16:
// o[((tile * ks + kk) * 64 + lane) * 8 + e] = w[tile * 32 + (lane & 31)][16 kk + 8 (lane >> 5) + e], zeros outside
__global__ void pack_bfrag_f16_kernel(const float* __restrict__ w, int n, int k, int tiles, int ks, _Float16* __restrict__ o) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)tiles * ks * 512) return;
    const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
    const long long f = idx >> 9;
    const int kk = (int)(f % ks), tile = (int)(f / ks);
    const int row = tile * 32 + (lane & 31), col = 16 * kk + 8 * (lane >> 5) + e;
    o[idx] = (row < n && col < k) ? (_Float16)w[(size_t)row * k + col] : (_Float16)0.f;
}

// k steps of the expand GEMM: whole chunks of 3 (the K loop has no tail; the padding is zeros on both operands)
int mbw_expand_ksteps(int cin) { return ((cin + 15) / 16 + 2) / 3 * 3; }
// ... of the project GEMM's packed filter: whole chunks of 6
int mbw_project_ksteps(int hid) { return ((hid + 15) / 16 + 5) / 6 * 6; }

struct MbwPlan { int G, KS, KSP, NTP, NPAIR, xpitch, dpitch, d_off; size_t lds; };

constexpr size_t kLdsMax = 160 * 1024;

// images per workgroup and LDS layout, or false when the block does not fit
bool plan_mbw(int hw, int stride, int cin, int hid, int cout, int sq, int gmax, MbwPlan* p) {
    if (cin % 8 || hid % 16 || hid > 64 * kMbwWaves * kMbwMaxRounds) return false;
    const int px = hw * hw, hwo = (hw + stride - 1) / stride, pxo = hwo * hwo;
    p->KS = mbw_expand_ksteps(cin);
    p->KSP = hid / 16;
    p->NTP = (cout + 31) / 32;
    p->NPAIR = (hid + 63) / 64;
    for (int g = gmax; g >= 1; --g) {
        for (int pad = 16; pad >= 0; pad -= 16) {
            // (+16: rows start in different banks for the 16-lane groups of a ds_read_b128; dropped when the block does not fit otherwise)
            const int xpitch = p->KS * 32 + pad, dpitch = hid * 2 + pad;
            size_t first = (size_t)g * px * xpitch;
            const size_t alias = (size_t)g * (hid + sq) * 4;
            if (alias > first) first = alias;
            first = (first + 15) & ~(size_t)15;
            const size_t total = first + (size_t)g * pxo * dpitch;
            if (total <= kLdsMax) {
                p->G = g; p->xpitch = xpitch; p->dpitch = dpitch; p->d_off = (int)first; p->lds = total;
                return true;
            }
        }
    }
    return false;
}

template <int HW, int K, int G, int S = 1>
void launch_mbw_one(const MbwArgs& a, size_t lds, hipStream_t s) {
    // dynamic LDS above 64 KB has to be asked for -- per DEVICE (function attributes are per device and a process may hold handles on
    // several), so on every launch like the other launchers of effnet.hip; it is a host-side table update, not a device call
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mbconv_whole_kernel<HW, K, G, S>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax);
    hipLaunchKernelGGL((mbconv_whole_kernel<HW, K, G, S>), dim3((unsigned)((a.n + G - 1) / G)), dim3(kMbwThreads), lds, s, a);
}

// maps with an instantiated kernel: 3 x 3 .. 9 x 9 (two images per workgroup up to 5 x 5: one 32-row band each)
int mbw_gmax(int hw) { return hw >= 3 && hw <= 5 ? 2 : hw >= 6 && hw <= 9 ? 1 : 0; }

template <int HW>
void launch_mbw_hw(const MbwArgs& a, int k, int g, size_t lds, hipStream_t s) {
    if constexpr (HW <= 5) {
        if (g == 2) { if (k == 3) launch_mbw_one<HW, 3, 2>(a, lds, s); else launch_mbw_one<HW, 5, 2>(a, lds, s); return; }
    }
    if (k == 3) launch_mbw_one<HW, 3, 1>(a, lds, s); else launch_mbw_one<HW, 5, 1>(a, lds, s);
}

}  // namespace

#ifdef MBW_TRACE
static unsigned long long* adaf_mbw_trace_buf = nullptr;
static int adaf_mbw_trace_hid = 0, adaf_mbw_trace_k = 0;
extern "C" void adaf_mbw_set_trace(unsigned long long* p, int hid, int k) { adaf_mbw_trace_buf = p; adaf_mbw_trace_hid = hid; adaf_mbw_trace_k = k; }
#endif

// even_tiles: the expand filter (tiles in pairs, k steps in whole chunks); otherwise the project filter
size_t adaf_mbw_bfrag_halfs(int n, int k, bool even_tiles) {
    int tiles = (n + 31) / 32;
    if (even_tiles) tiles = (tiles + 1) & ~1;
    return (size_t)tiles * (even_tiles ? mbw_expand_ksteps(k) : mbw_project_ksteps(k)) * 512;
}

int adaf_mbw_tap_row(int k) { return (k * k + 2 + 3) / 4 * 4; }

// depthwise taps [K*K][hid] + folded BN -> one row per channel [hid][TP]: taps, scale, bias, zeros
__global__ void pack_dw_rows_kernel(const float* __restrict__ wd, const float* __restrict__ sd, const float* __restrict__ bd, int hid, int kk, int tp,
                                    float* __restrict__ o) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= hid * tp) return;
    const int c = idx / tp, t = idx - c * tp;
    o[idx] = t < kk ? wd[(size_t)t * hid + c] : t == kk ? sd[c] : t == kk + 1 ? bd[c] : 0.f;
}

void adaf_launch_pack_dw_rows(const float* wd, const float* sd, const float* bd, int hid, int k, float* o, hipStream_t s) {
    const int tp = adaf_mbw_tap_row(k);
    hipLaunchKernelGGL(pack_dw_rows_kernel, dim3((unsigned)((hid * tp + 255) / 256)), dim3(256), 0, s, wd, sd, bd, hid, k * k, tp, o);
}

void adaf_launch_pack_bfrag_f16(const float* w, int n, int k, bool even_tiles, void* o, hipStream_t s) {
    int tiles = (n + 31) / 32;
    if (even_tiles) tiles = (tiles + 1) & ~1;
    const int ks = even_tiles ? mbw_expand_ksteps(k) : mbw_project_ksteps(k);
    const long long total = (long long)tiles * ks * 512;
    hipLaunchKernelGGL(pack_bfrag_f16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, n, k, tiles, ks, static_cast<_Float16*>(o));
}

// stride 2: the one instantiated shape -- 9 x 9 -> 5 x 5, k = 5, one image per workgroup (B3's block 18 at 144^2 patches)
bool adaf_mbw_eligible(int hw, int k, int stride, int cin, int hid, int cout, int sq) {
    if ((k != 3 && k != 5) || mbw_gmax(hw) == 0) return false;
    if (stride == 2 ? !(hw == 9 && k == 5) : stride != 1) return false;
    MbwPlan p;
    return plan_mbw(hw, stride, cin, hid, cout, sq, stride == 2 ? 1 : mbw_gmax(hw), &p);
}

int adaf_mbw_pad_before(int hw, int k, int stride) { return mbw_pad_before(hw, k, stride); }

// one launch for a whole MBConv block (fp16 storage, map hw x hw, SAME padding with adaf_mbw_pad_before() rows / columns in front);
// false = not eligible, nothing launched
bool adaf_launch_mbconv_whole(const void* x, int n, int hw, int stride, int cin, int hid, int cout, int sq, int k, const void* wef, const float* se,
                              const float* be, const float* wdl, const float* se_wr, const float* se_br,
                              const float* se_wet, const float* se_be, const void* wpf, const float* sp, const float* bp, bool skip, void* out,
                              hipStream_t s) {
    if (!adaf_mbw_eligible(hw, k, stride, cin, hid, cout, sq) || n <= 0 || (stride == 2 && skip)) return false;
    MbwPlan p;
    if (!plan_mbw(hw, stride, cin, hid, cout, sq, stride == 2 ? 1 : mbw_gmax(hw), &p)) return false;
    MbwArgs a;
    memset(&a, 0, sizeof(a));
    a.x = static_cast<const _Float16*>(x); a.res = skip ? a.x : nullptr; a.out = static_cast<_Float16*>(out);
    a.wef = static_cast<const u32x4*>(wef); a.se = se; a.be = be; a.wdl = wdl;
    a.se_wr = se_wr; a.se_br = se_br; a.se_wet = se_wet; a.se_be = se_be;
    a.wpf = static_cast<const u32x4*>(wpf); a.sp = sp; a.bp = bp;
    a.n = n; a.cin = cin; a.hid = hid; a.cout = cout; a.sq = sq;
#ifdef MBW_TRACE
    a.trace = (adaf_mbw_trace_hid == 0 || adaf_mbw_trace_hid == hid) && adaf_mbw_trace_k == k ? adaf_mbw_trace_buf : nullptr;
#endif
    a.KS = p.KS; a.KSP = p.KSP; a.KSPP = mbw_project_ksteps(hid); a.NTP = p.NTP; a.NPAIR = p.NPAIR; a.xpitch = p.xpitch; a.dpitch = p.dpitch; a.d_off = p.d_off;
    if (stride == 2) { launch_mbw_one<9, 5, 1, 2>(a, p.lds, s); return true; }
    switch (hw) {
        case 3: launch_mbw_hw<3>(a, k, p.G, p.lds, s); break;
        case 4: launch_mbw_hw<4>(a, k, p.G, p.lds, s); break;
        case 5: launch_mbw_hw<5>(a, k, p.G, p.lds, s); break;
        case 6: launch_mbw_hw<6>(a, k, p.G, p.lds, s); break;
        case 7: launch_mbw_hw<7>(a, k, p.G, p.lds, s); break;
        case 8: launch_mbw_hw<8>(a, k, p.G, p.lds, s); break;
        case 9: launch_mbw_hw<9>(a, k, p.G, p.lds, s); break;
        default: return false;
    }
    return true;
}
