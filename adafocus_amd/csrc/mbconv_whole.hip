// Whole-image MBConv blocks of EfficientNet (BASELINE config 5, fp16 storage): expand 1x1 + BN + swish -> depthwise k x k + BN + swish
// -> squeeze-and-excite -> gated project 1x1 + BN (+ identity) as ONE launch for the blocks whose map is small enough that a
// workgroup owns whole images (9 x 9 and 5 x 5 at 144^2 patches: 16 of B3's 26 blocks).  The 6x-expanded map never exists in HBM
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
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "adaf_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int kMbwThreads = 512;
constexpr int kMbwWaves = kMbwThreads / 64;
constexpr int kMbwMaxRounds = 5;          // channel pairs per wave: hid <= 64 * 8 * 5

__device__ __forceinline__ float w_sigmoid(float v) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f)); }
__device__ __forceinline__ float w_swish(float v) { return v * w_sigmoid(v); }

struct MbwArgs {
    const _Float16* x;       // [n][HW*HW][cin] block input
    const _Float16* res;     // identity rows (= x) or nullptr
    _Float16* out;           // [n][HW*HW][cout]
    const u32x4* wef;        // expand filter in B-fragment order [NT2][KS][64] x 16 B (NT2 = tiles rounded up to even; zero padded)
    const float* se;         // expand BN [hid]
    const float* be;
    const float* wd;         // depthwise taps [K*K][hid]
    const float* sd;         // depthwise BN [hid]
    const float* bd;
    const float* se_wr;      // [sq][hid]
    const float* se_br;      // [sq]
    const float* se_wet;     // [sq][hid] (transposed _se_expand.weight)
    const float* se_be;      // [hid]
    const u32x4* wpf;        // project filter in B-fragment order [NTP][KSP][64] x 16 B
    const float* sp;         // project BN [cout]
    const float* bp;
    int n, cin, hid, cout, sq;
    int KS, KSP, NTP, NPAIR; // expand k steps (of 16), project k steps, project column tiles, channel pairs (of 64)
    int xpitch, dpitch;      // LDS row pitches in bytes
    int d_off;               // byte offset of D in the dynamic LDS (X and, later, mean / gate / squeezed vector come first)
};

template <int HW, int K, int G>
__global__ __launch_bounds__(kMbwThreads) void mbconv_whole_kernel(const MbwArgs a) {
    constexpr int PX = HW * HW;
    constexpr int RB = (PX + 31) / 32;         // row bands per image
    constexpr int NB = G * RB;
    constexpr int P = (K - 1) / 2;             // SAME padding at stride 1 is symmetric
    constexpr int KC = 3;                      // k steps per prefetched chunk of B fragments (expand)
    constexpr int KCP = 4;                     // ... (project)
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nl = lane & 31, half = lane >> 5;
    const int img0 = blockIdx.x * G;
    const int nimg = min(G, a.n - img0);
    char* xl = dsm;
    char* dl = dsm + a.d_off;
    const int hid = a.hid;

    // ---- phase 0: X -> LDS ----
    {
        const int cpr = a.cin >> 3, cpp = a.KS * 2;
        const int rows = nimg * PX;
        const _Float16* xb = a.x + (size_t)img0 * PX * a.cin;
        for (int i = tid; i < G * PX * cpp; i += kMbwThreads) {
            const int r = i / cpp, c = i - r * cpp;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (r < rows && c < cpr) v = *reinterpret_cast<const u32x4*>(xb + (size_t)r * a.cin + c * 8);
            *reinterpret_cast<u32x4*>(xl + r * a.xpitch + c * 16) = v;
        }
    }
    __syncthreads();

    // row of band b this lane feeds the MFMAs with (rows past the last pixel repeat the last row: their results are never used)
    int xoff[NB], doff[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        int row = (b / RB) * PX + (b % RB) * 32 + nl;
        row = row < G * PX ? row : G * PX - 1;
        xoff[b] = row * a.xpitch + half * 16;
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
        auto load_b = [&](u32x4 (&dst)[2][KC], int jp, int k0) {
            const u32x4* p0 = a.wef + ((size_t)(2 * jp) * KS + k0) * 64 + lane;
            const u32x4* p1 = p0 + (size_t)KS * 64;
#pragma unroll
            for (int u = 0; u < KC; ++u)
                if (k0 + u < KS) { dst[0][u] = p0[u * 64]; dst[1][u] = p1[u * 64]; }
        };
        u32x4 bc[2][KC], bn[2][KC];
        if (wave < a.NPAIR) load_b(bc, wave, 0);
        int rd = 0;
        for (int jp = wave; jp < a.NPAIR; jp += kMbwWaves, ++rd) {
            const int c = 64 * jp + lane;            // the channel this lane owns in the depthwise part
            const bool cok = c < hid;
            const int cc = cok ? c : 0;
            // operands of the vector part: requested now, needed after the products
            float w[K * K];
#pragma unroll
            for (int t = 0; t < K * K; ++t) w[t] = a.wd[(size_t)t * hid + cc];
            const float sdl = a.sd[cc], bdl = a.bd[cc];
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
            for (int k0 = 0; k0 < KS; k0 += KC) {
                if (k0 + KC < KS) load_b(bn, jp, k0 + KC);
#pragma unroll
                for (int u = 0; u < KC; ++u) {
                    if (k0 + u < KS) {
                        f16x8 af[NB];
#pragma unroll
                        for (int b = 0; b < NB; ++b) af[b] = *reinterpret_cast<const f16x8*>(xl + xoff[b] + (k0 + u) * 32);
#pragma unroll
                        for (int b = 0; b < NB; ++b) {
                            acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[b], __builtin_bit_cast(f16x8, bc[0][u]), acc[0][b], 0, 0, 0);
                            acc[1][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[b], __builtin_bit_cast(f16x8, bc[1][u]), acc[1][b], 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < KC; ++u) { bc[0][u] = bn[0][u]; bc[1][u] = bn[1][u]; }
            }
            // the next pair's first fragments travel under the vector part
            if (jp + kMbwWaves < a.NPAIR) load_b(bc, jp + kMbwWaves, 0);

            // BN + swish on the accumulators, rounded to the storage type (what the expand launch would have written)
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    acc[0][b][i] = (float)(_Float16)w_swish(fmaf(acc[0][b][i], sce0, bie0));
                    acc[1][b][i] = (float)(_Float16)w_swish(fmaf(acc[1][b][i], sce1, bie1));
                }
            // lanes 0-31 take tile 0's other half-rows, lanes 32-63 give them and take tile 1's: afterwards acc[h][b][i] of lane l is
            // row 32 b + (i & 3) + 8 (i >> 2) + 4 h of channel 64 jp + l
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[0][b][i]), __float_as_uint(acc[1][b][i]), false, false);
                    acc[0][b][i] = __uint_as_float(r[0]);
                    acc[1][b][i] = __uint_as_float(r[1]);
                }
            // depthwise k x k + BN + swish out of registers
#pragma unroll
            for (int g = 0; g < G; ++g) {
                float ps = 0.f;
                char* dp = dl + (size_t)g * PX * a.dpitch + cc * 2;
#pragma unroll
                for (int oy = 0; oy < HW; ++oy) {
                    float s[HW];
#pragma unroll
                    for (int ox = 0; ox < HW; ++ox) s[ox] = 0.f;
#pragma unroll
                    for (int ky = 0; ky < K; ++ky)
#pragma unroll
                        for (int kx = 0; kx < K; ++kx)
#pragma unroll
                            for (int ox = 0; ox < HW; ++ox) {
                                const int iy = oy + ky - P, ix = ox + kx - P;
                                if (iy < 0 || iy >= HW || ix < 0 || ix >= HW) continue;      // resolved at compile time
                                const int p = iy * HW + ix, r = p & 31;
                                s[ox] = fmaf(acc[(r >> 2) & 1][g * RB + (p >> 5)][(r & 3) + 4 * (r >> 3)], w[ky * K + kx], s[ox]);
                            }
#pragma unroll
                    for (int ox = 0; ox < HW; ++ox) {
                        const float v = w_swish(fmaf(s[ox], sdl, bdl));
                        ps += v;
                        if (cok && g < nimg) *reinterpret_cast<_Float16*>(dp) = (_Float16)v;
                        dp += a.dpitch;
                    }
                }
#pragma unroll
                for (int q = 0; q < kMbwMaxRounds; ++q) psr[q][g] = q == rd ? ps : psr[q][g];
            }
        }
    }
    __syncthreads();           // X is dead, D is complete

    // ---- phase 2: squeeze-and-excite ----
    float* mean = reinterpret_cast<float*>(xl);            // [G][hid], later the gate
    float* sqv = mean + G * hid;                            // [G][sq]
    {
        const float inv_hw = 1.f / (float)PX;
#pragma unroll
        for (int q = 0; q < kMbwMaxRounds; ++q) {
            const int c = 64 * (wave + q * kMbwWaves) + lane;
            if (c < hid) {
#pragma unroll
                for (int g = 0; g < G; ++g) mean[g * hid + c] = psr[q][g] * inv_hw;
            }
        }
    }
    __syncthreads();
    {
        const int C4 = hid >> 2;
        for (int j = wave; j < a.sq; j += kMbwWaves) {
            const float* wrow = a.se_wr + (size_t)j * hid;
            float s[G];
#pragma unroll
            for (int g = 0; g < G; ++g) s[g] = 0.f;
#pragma unroll 4
            for (int c4 = lane; c4 < C4; c4 += 64) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wrow + 4 * c4);
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const f32x4 m = *reinterpret_cast<const f32x4*>(mean + g * hid + 4 * c4);
                    s[g] = fmaf(m.x, wv.x, fmaf(m.y, wv.y, fmaf(m.z, wv.z, fmaf(m.w, wv.w, s[g]))));
                }
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) s[g] += __shfl_xor(s[g], off, 64);
                if (lane == 0) {
                    const float v = s[g] + a.se_br[j];
                    sqv[g * a.sq + j] = v * w_sigmoid(v);
                }
            }
        }
    }
    __syncthreads();
    {
        const int C4 = hid >> 2, SQ = a.sq;
        for (int c4 = tid; c4 < C4; c4 += kMbwThreads) {
            f32x4 s[G];
            const f32x4 b = *reinterpret_cast<const f32x4*>(a.se_be + 4 * c4);
#pragma unroll
            for (int g = 0; g < G; ++g) s[g] = b;
            const float* wp = a.se_wet + 4 * c4;
#pragma unroll 8
            for (int j = 0; j < SQ; ++j) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(wp + (size_t)j * hid);
#pragma unroll
                for (int g = 0; g < G; ++g) s[g] += wv * sqv[g * SQ + j];
            }
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const f32x4 o = {w_sigmoid(s[g].x), w_sigmoid(s[g].y), w_sigmoid(s[g].z), w_sigmoid(s[g].w)};
                *reinterpret_cast<f32x4*>(mean + g * hid + 4 * c4) = o;          // (every thread is past its reads of `mean`: the barrier above)
            }
        }
    }
    __syncthreads();

    // ---- phase 3: D *= gate (fp32 product, rounded to fp16) ----
    {
        const int cpr = hid >> 3;                     // 16-byte chunks per row
        const int total = nimg * PX * cpr;
        const int drow = kMbwThreads / cpr, dcc = kMbwThreads - drow * cpr;
        int row = tid / cpr, cq = tid - row * cpr;
        for (int i = tid; i < total; i += kMbwThreads) {
            char* p = dl + row * a.dpitch + cq * 16;
            const float* gp = mean + (G > 1 && row >= PX ? hid : 0) + cq * 8;
            const f16x8 v = *reinterpret_cast<const f16x8*>(p);
            const f32x4 g0 = *reinterpret_cast<const f32x4*>(gp), g1 = *reinterpret_cast<const f32x4*>(gp + 4);
            f16x8 o;
            o[0] = (_Float16)((float)v[0] * g0.x); o[1] = (_Float16)((float)v[1] * g0.y);
            o[2] = (_Float16)((float)v[2] * g0.z); o[3] = (_Float16)((float)v[3] * g0.w);
            o[4] = (_Float16)((float)v[4] * g1.x); o[5] = (_Float16)((float)v[5] * g1.y);
            o[6] = (_Float16)((float)v[6] * g1.z); o[7] = (_Float16)((float)v[7] * g1.w);
            *reinterpret_cast<f16x8*>(p) = o;
            row += drow; cq += dcc;
            if (cq >= cpr) { cq -= cpr; ++row; }
        }
    }
    __syncthreads();

    // ---- phase 4: project 1x1 + BN (+ identity), one 32-column tile per wave iteration ----
    {
        const int KSP = a.KSP;
        for (int nt = wave; nt < a.NTP; nt += kMbwWaves) {
            const u32x4* bp0 = a.wpf + (size_t)nt * KSP * 64 + lane;
            u32x4 bc[KCP], bn[KCP];
#pragma unroll
            for (int u = 0; u < KCP; ++u)
                if (u < KSP) bc[u] = bp0[u * 64];
            f32x16 acc[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[b][i] = 0.f;
            for (int k0 = 0; k0 < KSP; k0 += KCP) {
#pragma unroll
                for (int u = 0; u < KCP; ++u)
                    if (k0 + KCP + u < KSP) bn[u] = bp0[(k0 + KCP + u) * 64];
#pragma unroll
                for (int u = 0; u < KCP; ++u) {
                    if (k0 + u < KSP) {
#pragma unroll
                        for (int b = 0; b < NB; ++b) {
                            const f16x8 af = *reinterpret_cast<const f16x8*>(dl + doff[b] + (k0 + u) * 32);
                            acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af, __builtin_bit_cast(f16x8, bc[u]), acc[b], 0, 0, 0);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < KCP; ++u) bc[u] = bn[u];
            }
            const int ncol = nt * 32 + nl;
            if (ncol < a.cout) {
                const float sc = a.sp[ncol], bi = a.bp[ncol];
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const int g = b / RB;
                    if (g >= nimg) continue;
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int p = (b % RB) * 32 + (i & 3) + 8 * (i >> 2) + 4 * half;
                        if (p < PX) {
                            const size_t o = ((size_t)(img0 + g) * PX + p) * a.cout + ncol;
                            float v = fmaf(acc[b][i], sc, bi);
                            if (a.res) v += (float)a.res[o];
                            a.out[o] = (_Float16)v;
                        }
                    }
                }
            }
        }
    }
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

struct MbwPlan { int G, KS, KSP, NTP, NPAIR, xpitch, dpitch, d_off; size_t lds; };

constexpr size_t kLdsMax = 160 * 1024;

// images per workgroup and LDS layout, or false when the block does not fit
bool plan_mbw(int hw, int cin, int hid, int cout, int sq, int gmax, MbwPlan* p) {
    if (cin % 8 || hid % 16 || hid > 64 * kMbwWaves * kMbwMaxRounds) return false;
    const int px = hw * hw;
    p->KS = (cin + 15) / 16;
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
            const size_t total = first + (size_t)g * px * dpitch;
            if (total <= kLdsMax) {
                p->G = g; p->xpitch = xpitch; p->dpitch = dpitch; p->d_off = (int)first; p->lds = total;
                return true;
            }
        }
    }
    return false;
}

template <int HW, int K, int G>
void launch_mbw_one(const MbwArgs& a, size_t lds, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {      // dynamic LDS above 64 KB has to be asked for
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mbconv_whole_kernel<HW, K, G>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax);
        attr_set = true;
    }
    hipLaunchKernelGGL((mbconv_whole_kernel<HW, K, G>), dim3((unsigned)((a.n + G - 1) / G)), dim3(kMbwThreads), lds, s, a);
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

size_t adaf_mbw_bfrag_halfs(int n, int k, bool even_tiles) {
    int tiles = (n + 31) / 32;
    if (even_tiles) tiles = (tiles + 1) & ~1;
    return (size_t)tiles * ((k + 15) / 16) * 512;
}

void adaf_launch_pack_bfrag_f16(const float* w, int n, int k, bool even_tiles, void* o, hipStream_t s) {
    int tiles = (n + 31) / 32;
    if (even_tiles) tiles = (tiles + 1) & ~1;
    const int ks = (k + 15) / 16;
    const long long total = (long long)tiles * ks * 512;
    hipLaunchKernelGGL(pack_bfrag_f16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, n, k, tiles, ks, static_cast<_Float16*>(o));
}

bool adaf_mbw_eligible(int hw, int k, int stride, int cin, int hid, int cout, int sq) {
    if (stride != 1 || (k != 3 && k != 5) || mbw_gmax(hw) == 0) return false;
    MbwPlan p;
    return plan_mbw(hw, cin, hid, cout, sq, mbw_gmax(hw), &p);
}

// one launch for a whole MBConv block (fp16 storage, stride 1, map hw x hw); false = not eligible, nothing launched
bool adaf_launch_mbconv_whole(const void* x, int n, int hw, int cin, int hid, int cout, int sq, int k, const void* wef, const float* se,
                              const float* be, const float* wd, const float* sd, const float* bd, const float* se_wr, const float* se_br,
                              const float* se_wet, const float* se_be, const void* wpf, const float* sp, const float* bp, bool skip, void* out,
                              hipStream_t s) {
    if (!adaf_mbw_eligible(hw, k, 1, cin, hid, cout, sq) || n <= 0) return false;
    MbwPlan p;
    if (!plan_mbw(hw, cin, hid, cout, sq, mbw_gmax(hw), &p)) return false;
    MbwArgs a;
    memset(&a, 0, sizeof(a));
    a.x = static_cast<const _Float16*>(x); a.res = skip ? a.x : nullptr; a.out = static_cast<_Float16*>(out);
    a.wef = static_cast<const u32x4*>(wef); a.se = se; a.be = be; a.wd = wd; a.sd = sd; a.bd = bd;
    a.se_wr = se_wr; a.se_br = se_br; a.se_wet = se_wet; a.se_be = se_be;
    a.wpf = static_cast<const u32x4*>(wpf); a.sp = sp; a.bp = bp;
    a.n = n; a.cin = cin; a.hid = hid; a.cout = cout; a.sq = sq;
    a.KS = p.KS; a.KSP = p.KSP; a.NTP = p.NTP; a.NPAIR = p.NPAIR; a.xpitch = p.xpitch; a.dpitch = p.dpitch; a.d_off = p.d_off;
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
