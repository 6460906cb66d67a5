// ResNet stem: conv 7x7 / stride 2 / pad 3, 3 -> 64 channels, + BN affine + ReLU (+ max-pool 3x3 / 2 / pad 1 in the same
// launch) -- ACT/models/resnet.py:138-141, 212-215.  Specialised because the generic engine wastes a third of its MFMAs on
// it (K = 49 taps x 4 padded channels = 196 -> 224 after slice padding, of which 147 are real).
//
// GEMM view per output pixel: K = 7 x 7 x 3 = 147 products.  The two k-lanes of v_mfma_f32_32x32x2_f32 split K in halves:
// lanes 0-31 take k = j, lanes 32-63 k = 74 + j for step j = 0..73 (k = 147 is a zero weight), so 147 of 148 multiplies
// are real.  A block keeps the whole filter bank in LDS in that order and stages the input window of its output tile once
// (the zero 4th lane of the NHWC4 patch is dropped): with k = kh*21 + kw*3 + c the A value of a pixel is one ds_read_b32 at
// a compile-time offset from the pixel's window origin (one offset per k-lane half, picked by a v_cndmask).  Any k order is
// exact in fp32 as long as both operands use it; both kernels below use the same one, so they are bit-identical.
//
//   stem7x7_kernel       8 x 16 output pixels per tile, 4 waves: conv + BN + ReLU only (fusion off / tests).
//   stem7x7_pool_kernel  the pooled map directly: a tile of TPH x 8 POOLED pixels needs (2 TPH + 1) x 17 conv outputs
//                        (one halo row / column), i.e. 5 (TPH = 4) or 7 (TPH = 6) waves of 32 pixels; their BN + ReLU
//                        results go to LDS 32 channels at a time (pixels outside the map as 0 -- every pooling window holds
//                        a real, non-negative value, so 0 is as good as the -inf padding of nn.MaxPool2d) and 256 threads
//                        take the 3 x 3 maxima with 16-byte reads and stores.  The 604 MB conv map (1024 patches of 96^2)
//                        is never written: 151 MB of pooled output instead, one launch instead of two.
// Both are persistent (a block loops over tiles) and software-pipelined: the next tile's window is fetched into registers
// before the current tile's MFMAs and written to LDS after them, so no global latency sits between two tiles.
#include "adaf_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int KS = 74;                    // MFMA steps (k pairs): k = j | 74 + j

// w_oihw [64][3][7][7] -> Wr[h][j][n], k = 74 h + j = kh*21 + kw*3 + c (k = 147: zero)
__global__ void pack_stem_weight_kernel(const float* __restrict__ w, float* __restrict__ o) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 2 * KS * 64) return;
    const int n = idx & 63, j = (idx >> 6) % KS, h = idx / (KS * 64);
    const int k = KS * h + j;
    const int kh = k / 21, r = k % 21, kw = r / 3, c = r % 3;
    o[idx] = k < 147 ? w[((n * 3 + c) * 7 + kh) * 7 + kw] : 0.f;
}

struct StemArgs {
    const float* x;      // [n, P, P, 4]
    const float* w;      // Wr [2][74][64]
    const float* scale;  // [64]
    const float* bias;
    float* out;          // [n, OH, OW, 64] (conv map) or [n, PH, PW, 64] (pooled map)
    int n, P, OH, OW, PH, PW, tiles_y, tiles_x, ntiles;
};

// window-relative offset (floats) of the A value of k-lane half h at step j; RP = floats per window row
template <int RP>
__device__ __forceinline__ constexpr int a_off(int h, int j) {
    const int k = KS * h + j;
    return k < 147 ? (k / 21) * RP + (k % 21) : -1;
}

// The K walk shared by both kernels: `abase` = the lane's pixel origin in the window.  Both k-lane halves' A values are
// read (compile-time offsets -> immediate fields, no address registers: a per-lane SELECTED address made the compiler keep
// 74 loop-invariant addresses live, 214 VGPRs) and the half picks its own with a v_cndmask on the data.  Groups of 8 steps,
// operands of group g+1 read from LDS while group g multiplies.
template <int RP, int G = 8>
__device__ __forceinline__ void stem_mfma(const float* abase, const float* bbase, int h, f32x16& acc0, f32x16& acc1) {
    constexpr int NG = (KS + G - 1) / G;
    float av[2][G], b0[2][G], b1[2][G];
    auto rd = [&](int g, int buf) {
#pragma unroll
        for (int u = 0; u < G; ++u) {
            const int j = g * G + u;
            if (j < KS) {
                const int o0 = a_off<RP>(0, j);
                const int o1 = a_off<RP>(1, j);
                const float a0 = abase[o0];
                const float a1 = o1 < 0 ? 0.f : abase[o1 < 0 ? 0 : o1];      // k = 147: the zero-weight slot
                av[buf][u] = h ? a1 : a0;
                b0[buf][u] = bbase[j * 64];
                b1[buf][u] = bbase[j * 64 + 32];
            }
        }
    };
    rd(0, 0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) rd(g + 1, (g + 1) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < G; ++u) {
            if (g * G + u < KS) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g & 1][u], b0[g & 1][u], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av[g & 1][u], b1[g & 1][u], acc1, 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- conv + BN + ReLU only ------------------------------------------------------------------------------------
constexpr int TH = 8, TW = 16;                       // output tile
constexpr int RH = 2 * TH + 5, RW = 2 * TW + 5;      // staged input rows / cols
constexpr int RP = RW * 3;                           // floats per staged row
constexpr int WPIX = RH * RW;                        // window pixels
constexpr int WPT = (WPIX + 255) / 256;              // window pixels per thread

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void stem7x7_kernel(const StemArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[2 * KS * 64 + RH * RP + 8];
    float* Ws = smem;
    float* reg = smem + 2 * KS * 64;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * KS * 64 / 4; i += 256) reinterpret_cast<f32x4*>(Ws)[i] = reinterpret_cast<const f32x4*>(a.w)[i];

    const int h = lane >> 5, li = lane & 31;
    const int py = 2 * wave + (li >> 4), px = li & 15;          // this lane's output pixel inside the tile (as an A row)
    const float* abase = reg + ((2 * py) * RW + 2 * px) * 3;
    const float* bbase = Ws + h * KS * 64 + li;
    const float sc0 = a.scale[li], sc1 = a.scale[32 + li], bi0 = a.bias[li], bi1 = a.bias[32 + li];
    const int per_img = a.tiles_y * a.tiles_x;

    f32x4 win[WPT];
    auto fetch = [&](int tile) {
        const int img = tile / per_img, rem = tile - img * per_img;
        const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
        const int iy0 = 2 * ty * TH - 3, ix0 = 2 * tx * TW - 3;
#pragma unroll
        for (int u = 0; u < WPT; ++u) {
            const int idx = tid + 256 * u;
            const int r = idx / RW, c = idx - r * RW;
            const int iy = iy0 + r, ix = ix0 + c;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (idx < WPIX && (unsigned)iy < (unsigned)a.P && (unsigned)ix < (unsigned)a.P)
                v = *reinterpret_cast<const f32x4*>(a.x + (((size_t)img * a.P + iy) * a.P + ix) * 4);
            win[u] = v;
        }
    };
    int tile = blockIdx.x;
    if (tile < a.ntiles) fetch(tile);
    for (; tile < a.ntiles; tile += gridDim.x) {
        const int img = tile / per_img, rem = tile - img * per_img;
        const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
        const int oy0 = ty * TH, ox0 = tx * TW;
        __syncthreads();   // previous tile's window fully consumed (and Ws written, first time round)
#pragma unroll
        for (int u = 0; u < WPT; ++u) {
            const int idx = tid + 256 * u;
            if (idx < WPIX) {
                float* d = reg + idx * 3;
                d[0] = win[u].x; d[1] = win[u].y; d[2] = win[u].z;
            }
        }
        if (tile + (int)gridDim.x < a.ntiles) fetch(tile + gridDim.x);     // in flight during this tile's MFMAs
        __syncthreads();

        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        stem_mfma<RP>(abase, bbase, h, acc0, acc1);
        // C layout: col = lane&31 (channel), row = (r&3) + 8(r>>2) + 4(lane>>5) (pixel index within the wave's 32)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pi = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int oy = oy0 + 2 * wave + (pi >> 4), ox = ox0 + (pi & 15);
            if (oy < a.OH && ox < a.OW) {
                float* o = a.out + (((size_t)img * a.OH + oy) * a.OW + ox) * 64;
                o[li] = fmaxf(fmaf(acc0[r], sc0, bi0), 0.f);
                o[32 + li] = fmaxf(fmaf(acc1[r], sc1, bi1), 0.f);
            }
        }
    }
}

// ---- conv + BN + ReLU + max-pool ----------------------------------------------------------------------------------
template <int TPH>
struct PoolCfg {
    static constexpr int TPW = 8;
    static constexpr int SH = 2 * TPH + 1, SW = 2 * TPW + 1;        // conv outputs needed (one halo row / column)
    static constexpr int NPIX = SH * SW;
    static constexpr int NWV = (NPIX + 31) / 32;                      // waves
    static constexpr int NT = 64 * NWV;
    static constexpr int RH = 2 * SH + 5, RW = 2 * SW + 5;           // input window
    static constexpr int RP = RW * 3;
    static constexpr int WPIX = RH * RW;
    static constexpr int WPT = (WPIX + NT - 1) / NT;
    static constexpr int SPITCH = 40;                                 // floats per pixel of the 32-channel staging image
    static constexpr int WIN = RH * RP + 8;
    static constexpr int SIMG = NPIX * SPITCH;
    static constexpr int REGION = WIN > SIMG ? WIN : SIMG;            // the window and the staging image share LDS
    static constexpr int WPE = (2 * NWV + 3) / 4;                     // waves per SIMD with two blocks resident on a CU
};

template <int TPH>
__global__ __launch_bounds__(PoolCfg<TPH>::NT) __attribute__((amdgpu_waves_per_eu(PoolCfg<TPH>::WPE, PoolCfg<TPH>::WPE)))
void stem7x7_pool_kernel(const StemArgs a) {
    using C = PoolCfg<TPH>;
    __shared__ __attribute__((aligned(16))) float smem[2 * KS * 64 + C::REGION];
    float* Ws = smem;
    float* reg = smem + 2 * KS * 64;
    float* S = reg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * KS * 64 / 4; i += C::NT) reinterpret_cast<f32x4*>(Ws)[i] = reinterpret_cast<const f32x4*>(a.w)[i];

    const int h = lane >> 5, li = lane & 31;
    const int pix = 32 * wave + li;                              // this lane's conv pixel of the tile (as an A row)
    const int pixc = pix < C::NPIX ? pix : 0;
    const int sy_l = pixc / C::SW, sx_l = pixc - sy_l * C::SW;
    const int aorg = ((2 * sy_l) * C::RW + 2 * sx_l) * 3;
    const float* abase = reg + aorg;
    const float* bbase = Ws + h * KS * 64 + li;
    const float sc0 = a.scale[li], sc1 = a.scale[32 + li], bi0 = a.bias[li], bi1 = a.bias[32 + li];
    const int per_img = a.tiles_y * a.tiles_x;

    f32x4 win[C::WPT];
    auto fetch = [&](int tile) {
        const int img = tile / per_img, rem = tile - img * per_img;
        const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
        // pooled tile origin (ty*TPH, tx*8) -> conv origin 2*p0 - 1 -> input origin 2*(2*p0 - 1) - 3
        const int iy0 = 4 * ty * TPH - 5, ix0 = 4 * tx * C::TPW - 5;
#pragma unroll
        for (int u = 0; u < C::WPT; ++u) {
            const int idx = tid + C::NT * u;
            const int r = idx / C::RW, c = idx - r * C::RW;
            const int iy = iy0 + r, ix = ix0 + c;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (idx < C::WPIX && (unsigned)iy < (unsigned)a.P && (unsigned)ix < (unsigned)a.P)
                v = *reinterpret_cast<const f32x4*>(a.x + (((size_t)img * a.P + iy) * a.P + ix) * 4);
            win[u] = v;
        }
    };
    int tile = blockIdx.x;
    if (tile < a.ntiles) fetch(tile);
    for (; tile < a.ntiles; tile += gridDim.x) {
        const int img = tile / per_img, rem = tile - img * per_img;
        const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
        const int py0 = ty * TPH, px0 = tx * C::TPW;
        const int sy0 = 2 * py0 - 1, sx0 = 2 * px0 - 1;          // conv coordinates of the tile's first row / column
        __syncthreads();   // previous tile's pooling reads are done (and Ws written, first time round)
#pragma unroll
        for (int u = 0; u < C::WPT; ++u) {
            const int idx = tid + C::NT * u;
            if (idx < C::WPIX) {
                float* d = reg + idx * 3;
                d[0] = win[u].x; d[1] = win[u].y; d[2] = win[u].z;
            }
        }
        if (tile + (int)gridDim.x < a.ntiles) fetch(tile + gridDim.x);     // in flight during this tile's MFMAs
        __syncthreads();

        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        stem_mfma<C::RP, (C::WPE >= 4 ? 4 : 8)>(abase, bbase, h, acc0, acc1);
        __syncthreads();   // every wave is done reading the window: the staging image may overwrite it

#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // BN + ReLU of 32 channels into S[pixel][channel]; conv pixels outside the map contribute 0
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (p < C::NPIX) {
                    const int sy = sy0 + p / C::SW, sx = sx0 + p % C::SW;
                    const bool in = (unsigned)sy < (unsigned)a.OH && (unsigned)sx < (unsigned)a.OW;
                    const float v = half ? fmaxf(fmaf(acc1[r], sc1, bi1), 0.f) : fmaxf(fmaf(acc0[r], sc0, bi0), 0.f);
                    S[p * C::SPITCH + li] = in ? v : 0.f;
                }
            }
            __syncthreads();
            if (tid < TPH * C::TPW * 8) {
                const int pp = tid >> 3, c4 = tid & 7;
                const int py = pp / C::TPW, px = pp - py * C::TPW;
                const float* s0 = S + ((2 * py) * C::SW + 2 * px) * C::SPITCH + 4 * c4;
                f32x4 m = *reinterpret_cast<const f32x4*>(s0);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        if (dy == 0 && dx == 0) continue;
                        const f32x4 v = *reinterpret_cast<const f32x4*>(s0 + (dy * C::SW + dx) * C::SPITCH);
                        m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
                    }
                if (py0 + py < a.PH && px0 + px < a.PW)
                    *reinterpret_cast<f32x4*>(a.out + (((size_t)img * a.PH + py0 + py) * a.PW + px0 + px) * 64 + 32 * half + 4 * c4) = m;
            }
            if (half == 0) __syncthreads();   // the second half overwrites the staging image
        }
    }
}

template <int TPH>
void launch_pool(StemArgs a, int cus, hipStream_t s) {
    using C = PoolCfg<TPH>;
    a.tiles_y = (a.PH + TPH - 1) / TPH; a.tiles_x = (a.PW + C::TPW - 1) / C::TPW;
    a.ntiles = a.n * a.tiles_y * a.tiles_x;
    const int grid = a.ntiles < cus * 2 ? a.ntiles : cus * 2;
    hipLaunchKernelGGL((stem7x7_pool_kernel<TPH>), dim3(grid), dim3(C::NT), 0, s, a);
}

}  // namespace

void adaf_launch_pack_stem_weight(const float* w_oihw, float* wr, hipStream_t s) {
    hipLaunchKernelGGL(pack_stem_weight_kernel, dim3((2 * KS * 64 + 255) / 256), dim3(256), 0, s, w_oihw, wr);
}

size_t adaf_stem_weight_floats() { return (size_t)2 * KS * 64; }

void adaf_launch_stem7x7(const float* x4, int n, int P, const float* wr, const float* scale, const float* bias, float* out,
                         int cus, hipStream_t s) {
    StemArgs a;
    a.x = x4; a.w = wr; a.scale = scale; a.bias = bias; a.out = out;
    a.n = n; a.P = P; a.OH = (P + 6 - 7) / 2 + 1; a.OW = a.OH; a.PH = a.PW = 0;
    a.tiles_y = (a.OH + TH - 1) / TH; a.tiles_x = (a.OW + TW - 1) / TW;
    a.ntiles = n * a.tiles_y * a.tiles_x;
    const int grid = a.ntiles < cus * 3 ? a.ntiles : cus * 3;
    hipLaunchKernelGGL(stem7x7_kernel, dim3(grid), dim3(256), 0, s, a);
}

// Measured (1024 patches): 96^2 -> 0.52 ms fused (6-row tiles) vs 0.40 + 0.16 ms for the two launches; 128^2, where only the
// 4-row tiles divide the map, 1.53 ms vs 0.70 + 0.28 ms.  The fused launch is used where the 6-row tiles are the cheaper cover.
bool adaf_stem7x7_pool_pays(int P) {
    const int oh = (P + 6 - 7) / 2 + 1, ph = (oh - 1) / 2 + 1;
    return ((ph + 5) / 6) * PoolCfg<6>::NWV < ((ph + 3) / 4) * PoolCfg<4>::NWV;
}

// conv 7x7/2 + BN + ReLU + max-pool 3x3/2/1 in one launch: out [n, PH, PW, 64] with PH = (OH - 1) / 2 + 1
void adaf_launch_stem7x7_pool(const float* x4, int n, int P, const float* wr, const float* scale, const float* bias, float* out,
                              int cus, hipStream_t s) {
    StemArgs a;
    a.x = x4; a.w = wr; a.scale = scale; a.bias = bias; a.out = out;
    a.n = n; a.P = P; a.OH = (P + 6 - 7) / 2 + 1; a.OW = a.OH; a.PH = (a.OH - 1) / 2 + 1; a.PW = a.PH;
    // rows of pooled pixels per tile: 4 (5 waves) or 6 (7 waves), whichever computes fewer conv pixels for this map
    const int cost4 = ((a.PH + 3) / 4) * PoolCfg<4>::NWV, cost6 = ((a.PH + 5) / 6) * PoolCfg<6>::NWV;
    if (cost6 < cost4) launch_pool<6>(a, cus, s);
    else launch_pool<4>(a, cus, s);
}
