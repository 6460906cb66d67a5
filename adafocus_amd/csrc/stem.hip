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
    const float* x;      // [n, P, P, 4] pixel-major patches -- or, with `act`, the planar frames [n, 3, H, W] the patches are cut from
    const float* act;    // nullptr, or [n / fpa, 2] fp32 (y, x) actions: the stem gathers its own input windows (get_patch folded in)
    int fpa, H, W;       // frames per action (1 = ActivityNet, T = Something-Something), frame size
    int nframes;         // frames behind `x` (patch i is cut from frame i % nframes)
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
    static constexpr int WPE = NWV >= 7 ? 3 : (2 * NWV + 3) / 4;      // waves per SIMD with two blocks resident on a CU (seven-wave blocks: 168 VGPRs
                                                                      // instead of 128 + 6 spilled -- the tile form now only runs where blocks are scarce)
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
        stem_mfma<C::RP, (C::WPE >= 4 ? 2 : 8)>(abase, bbase, h, acc0, acc1);     // (groups of 4 spilled 6 VGPRs under the 128-register cap of 4 waves per SIMD)
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

// ---- conv + BN + ReLU + max-pool over whole-width strips walked down the image (round 5) ------------------------------------------
// The tile form above recomputes a halo row and column per tile (13 x 17 conv outputs for 6 x 8 pooled ones: 1.17x the MFMA work) on
// seven waves per block (3.5 per SIMD with two blocks resident: 12.5 % of the matrix pipe's time is a missing wave), and its phases run
// in lockstep: MfmaUtil 0.61.  Here a block owns whole IMAGES: a strip is R conv rows of the full width (no column halo: column -1 is
// outside the map), strips are walked top to bottom and the one conv row a pooling window shares with the previous strip travels in
// REGISTERS: the thread that pooled the strip's last pooled row keeps the 3-column maxima of its bottom conv row and becomes the thread
// of the next strip's first pooled row (pooled-row index rotated by the strip number).  So every conv pixel is computed exactly once
// (R * OW = 32 * NW pixels: whole waves), the windows of strip s+1 are written to LDS while strip s pools (a separate buffer), the
// fetch of strip s+2 is in flight behind both, and a wave starts the next strip's MFMAs as soon as ITS pooling is done -- four barriers
// per strip, none between pooling and the next K walk.  Same k order, same BN / ReLU / max: bit-identical to the kernels above.
// With `act` the block gathers its windows straight from the planar frames at the crop origin (the reference's get_patch,
// ACT/models/utils.py:37-51, with crop.hip's exact coordinate arithmetic): no gather launch, no patch tensor.
template <int OW, int R>
struct RowsCfg {
    static constexpr int NPX = R * OW;                 // conv pixels per strip
    static constexpr int NW = NPX / 32;                // waves
    static constexpr int NT = 64 * NW;
    static constexpr int RW = 2 * OW + 5, RH = 2 * R + 5;   // staged input columns / rows
    static constexpr int RP = RW * 3;
    static constexpr int WPIX = RH * RW;
    static constexpr int WPT = (WPIX + NT - 1) / NT;
    static constexpr int SP = 36;                      // floats per pixel of the 32-channel staging image
    static constexpr int WIN = (RH * RP + 3) & ~3;
    static constexpr int SIMG = NPX * SP;
    static constexpr int LDSF = 2 * KS * 64 + WIN + SIMG;
    static constexpr int PW = OW / 2, PR = R / 2;      // pooled pixels per row, pooled rows per strip
    static constexpr int NS = OW / R;                  // strips per image (square maps)
    static constexpr int BPC = (LDSF * 4 + 1023) / 1024 * 1024 * 2 <= 160 * 1024 ? 2 : 1;
    static constexpr int WPE = (BPC * NW + 3) / 4;
    static_assert(NPX % 32 == 0 && OW % 2 == 0 && R % 2 == 0 && OW % R == 0 && PR * PW * 8 == NT, "strip geometry");
};

template <int OW, int R, int FR, int G = (RowsCfg<OW, R>::WPE >= 3 ? 4 : 8)>
__global__ __launch_bounds__((RowsCfg<OW, R>::NT)) __attribute__((amdgpu_waves_per_eu((RowsCfg<OW, R>::WPE), (RowsCfg<OW, R>::WPE))))
void stem7x7_pool_rows_kernel(const StemArgs a) {
    using C = RowsCfg<OW, R>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Ws = smem;
    float* reg = smem + 2 * KS * 64;
    float* S = reg + C::WIN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * KS * 64 / 4; i += C::NT) reinterpret_cast<f32x4*>(Ws)[i] = reinterpret_cast<const f32x4*>(a.w)[i];

    const int h = lane >> 5, li = lane & 31;
    const int q = 32 * wave + li;                                // this lane's conv pixel of the strip (as an A row)
    const float* abase = reg + ((2 * (q / OW)) * C::RW + 2 * (q % OW)) * 3;
    const float* bbase = Ws + h * KS * 64 + li;
    const float sc0 = a.scale[li], sc1 = a.scale[32 + li], bi0 = a.bias[li], bi1 = a.bias[32 + li];
    // pooling role: (pooled row j, pooled column px, channel quad c4)
    const int c4 = tid & 7, px = (tid >> 3) % C::PW, j = (tid >> 3) / C::PW;
    const int xl = (2 * px > 0 ? 2 * px - 1 : 0) * C::SP + 4 * c4, xm = 2 * px * C::SP + 4 * c4, xr = (2 * px + 1) * C::SP + 4 * c4;

    const int P = 2 * OW;
    const int nimg = (a.n - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;     // images of this block
    const int nsteps = nimg * C::NS;
    // gathering form: the window origins of this block's images, computed once (a dependent action load -> address -> frame load chain
    // in front of every strip's fetch would sit right before a barrier)
    constexpr int MAXO = 32;
    __shared__ int origin[2 * MAXO];
    if (FR != 0) {
        for (int i = tid; i < nimg && i < MAXO; i += C::NT) {
            const int img = blockIdx.x + i * gridDim.x;
            const int ai = img / a.fpa;
            // floor(action * (H - P)) for both axes (utils.py:40-42), clamped like crop_kernel
            const float span = (float)(a.H - P);
            const int y0 = (int)floorf(__fmul_rn(a.act[2 * ai], span)), x0 = (int)floorf(__fmul_rn(a.act[2 * ai + 1], span));
            origin[2 * i] = min(max(y0, 0), a.H - P);
            origin[2 * i + 1] = min(max(x0, 0), a.W - P);
        }
        __syncthreads();
    }
    f32x4 win[C::WPT];
    auto fetch = [&](int step) {
        const int img = blockIdx.x + (step / C::NS) * gridDim.x, st = step % C::NS;
        const int iy0 = 2 * R * st - 3;
        const float* src;
        if (FR) {
            // the window origin of this patch: floor(action * (H - P)) for both axes (utils.py:40-42), clamped like crop_kernel; patch `img`
            // is cut from frame img % nframes with action img / fpa (a second action set over the same frames: the reward baseline)
            const int ii = step / C::NS, fr = img % a.nframes;
            int y0, x0;
            if (ii < MAXO) { y0 = origin[2 * ii]; x0 = origin[2 * ii + 1]; }
            else {
                const int ai = img / a.fpa;
                const float span = (float)(a.H - P);
                y0 = (int)floorf(__fmul_rn(a.act[2 * ai], span)); x0 = (int)floorf(__fmul_rn(a.act[2 * ai + 1], span));
                y0 = min(max(y0, 0), a.H - P); x0 = min(max(x0, 0), a.W - P);
            }
            src = FR == 1 ? a.x + ((size_t)fr * 3 * a.H + y0) * a.W + x0 : a.x + (((size_t)fr * a.H + y0) * a.W + x0) * 4;
        } else {
            src = a.x + (size_t)img * P * P * 4;
        }
#pragma unroll
        for (int u = 0; u < C::WPT; ++u) {
            const int idx = tid + C::NT * u;
            const int r = idx / C::RW, c = idx - r * C::RW;
            const int iy = iy0 + r, ix = c - 3;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (idx < C::WPIX && (unsigned)iy < (unsigned)P && (unsigned)ix < (unsigned)P) {
                if (FR == 1) {
                    const float* p0 = src + (size_t)iy * a.W + ix;
                    const size_t plane = (size_t)a.H * a.W;
                    v.x = p0[0]; v.y = p0[plane]; v.z = p0[2 * plane];
                } else if (FR == 2) {
                    v = *reinterpret_cast<const f32x4*>(src + ((size_t)iy * a.W + ix) * 4);
                } else {
                    v = *reinterpret_cast<const f32x4*>(src + ((size_t)iy * P + ix) * 4);
                }
            }
            win[u] = v;
        }
    };
    auto put = [&]() {
#pragma unroll
        for (int u = 0; u < C::WPT; ++u) {
            const int idx = tid + C::NT * u;
            if (idx < C::WPIX) {
                float* d = reg + idx * 3;
                d[0] = win[u].x; d[1] = win[u].y; d[2] = win[u].z;
            }
        }
    };
    if (nsteps <= 0) return;
    fetch(0);
    put();
    if (nsteps > 1) fetch(1);
    __syncthreads();

    f32x4 carry[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int step = 0; step < nsteps; ++step) {
        const int img = blockIdx.x + (step / C::NS) * gridDim.x, st = step % C::NS;
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        stem_mfma<C::RP, G>(abase, bbase, h, acc0, acc1);
        __syncthreads();   // every wave is done with this strip's window; last strip's pooling reads of S are done

        const int je = (j + st) % C::PR;                            // this strip's pooled row of the thread (rotated: see above)
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            // BN + ReLU of 32 channels into S[pixel][channel]
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * h;
                S[p * C::SP + li] = half ? fmaxf(fmaf(acc1[r], sc1, bi1), 0.f) : fmaxf(fmaf(acc0[r], sc0, bi0), 0.f);
            }
            if (half == 0 && step + 1 < nsteps) {   // the next strip's window (fetched during the previous strip) -> LDS; the one after -> registers
                put();
                if (step + 2 < nsteps) fetch(step + 2);
            }
            __syncthreads();
            {
                auto hmax = [&](int row) {
                    const float* s0 = S + row * OW * C::SP;
                    const f32x4 l = *reinterpret_cast<const f32x4*>(s0 + xl), m = *reinterpret_cast<const f32x4*>(s0 + xm),
                                r_ = *reinterpret_cast<const f32x4*>(s0 + xr);
                    f32x4 o;
                    o.x = fmaxf(fmaxf(l.x, m.x), r_.x); o.y = fmaxf(fmaxf(l.y, m.y), r_.y);
                    o.z = fmaxf(fmaxf(l.z, m.z), r_.z); o.w = fmaxf(fmaxf(l.w, m.w), r_.w);
                    return o;
                };
                f32x4 top = carry[half];
                if (st == 0 && je == 0) top = f32x4{0.f, 0.f, 0.f, 0.f};     // first pooled row of an image: conv row -1 is outside
                if (je > 0) top = hmax(2 * je - 1);
                const f32x4 mid = hmax(2 * je), bot = hmax(2 * je + 1);
                f32x4 m;
                m.x = fmaxf(fmaxf(top.x, mid.x), bot.x); m.y = fmaxf(fmaxf(top.y, mid.y), bot.y);
                m.z = fmaxf(fmaxf(top.z, mid.z), bot.z); m.w = fmaxf(fmaxf(top.w, mid.w), bot.w);
                if (je == C::PR - 1) carry[half] = bot;
                const int py = st * C::PR + je;
                *reinterpret_cast<f32x4*>(a.out + (((size_t)img * C::PW + py) * C::PW + px) * 64 + 32 * half + 4 * c4) = m;
            }
            if (half == 0) __syncthreads();   // the second half overwrites the staging image
        }
    }
}

template <int OW, int R, int FR, int G = (RowsCfg<OW, R>::WPE >= 3 ? 4 : 8)>
void launch_rows(StemArgs a, int cus, hipStream_t s) {
    using C = RowsCfg<OW, R>;
    const int grid = a.n < cus * C::BPC ? a.n : cus * C::BPC;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem7x7_pool_rows_kernel<OW, R, FR, G>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              C::LDSF * 4);
    hipLaunchKernelGGL((stem7x7_pool_rows_kernel<OW, R, FR, G>), dim3(grid), dim3(C::NT), C::LDSF * 4, s, a);
}

// Strip height R and operand-group size G per patch size, from tools/stem_probe.py sweeps on 1024 patches (ms per launch, tile form first):
//   96^2: 0.519 -> R = 4 (two 6-wave blocks per CU) 0.509, R = 6 0.478, R = 8 (one 12-wave block per CU) 0.371 (G = 2 / 4 alike, G = 8 spills: 0.432)
//   128^2: 1.53 (0.98 as two launches) -> R = 2 (two 4-wave blocks) 0.69, R = 4 / 8 0.70-0.73;  144^2: 1.27 -> R = 4 (one 9-wave block) 1.07;
//   64^2: 0.39 -> 0.18.  Taller strips amortise the four barriers of a strip; a 16-wave block is the limit (R * OW <= 512).
template <int FR>
bool launch_rows_for(const StemArgs& a, int cus, hipStream_t s) {
    switch (a.P) {
        case 64: launch_rows<32, 4, FR, 8>(a, cus, s); return true;
        case 96: launch_rows<48, 8, FR, 2>(a, cus, s); return true;
        case 128: launch_rows<64, 2, FR, 8>(a, cus, s); return true;
        case 144: launch_rows<72, 4, FR, 2>(a, cus, s); return true;
        default: return false;
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
    a.x = x4; a.w = wr; a.scale = scale; a.bias = bias; a.out = out; a.act = nullptr; a.fpa = 1; a.H = a.W = 0; a.nframes = n;
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

// the strip-walking form: instantiated patch sizes, and a block owns whole images -- below one image per CU the tile form spreads better
bool adaf_stem7x7_rows_ok(int P, int n, int cus) {
    const int mode = adaf_options().stem_rows;      // 0 = off, 1 = default, 2 = at every image count (tests)
    if (!mode) return false;
    return (P == 64 || P == 96 || P == 128 || P == 144) && (n >= cus || mode == 2);
}

bool adaf_launch_stem7x7_pool_frames(const float* frames, bool pixel_major, int nframes, const float* act, int fpa, int H, int W, int n, int P,
                                     const float* wr, const float* scale, const float* bias, float* out, int cus, hipStream_t s) {
    if (!adaf_stem7x7_rows_ok(P, n, cus) || !act || fpa < 1 || nframes < 1 || H < P || W < P) return false;
    StemArgs a;
    a.x = frames; a.act = act; a.fpa = fpa; a.H = H; a.W = W; a.nframes = nframes; a.w = wr; a.scale = scale; a.bias = bias; a.out = out;
    a.n = n; a.P = P; a.OH = P / 2; a.OW = a.OH; a.PH = a.OH / 2; a.PW = a.PH; a.tiles_y = a.tiles_x = a.ntiles = 0;
    return pixel_major ? launch_rows_for<2>(a, cus, s) : launch_rows_for<1>(a, cus, s);
}

// conv 7x7/2 + BN + ReLU + max-pool 3x3/2/1 in one launch: out [n, PH, PW, 64] with PH = (OH - 1) / 2 + 1
void adaf_launch_stem7x7_pool(const float* x4, int n, int P, const float* wr, const float* scale, const float* bias, float* out,
                              int cus, hipStream_t s) {
    StemArgs a;
    a.x = x4; a.w = wr; a.scale = scale; a.bias = bias; a.out = out; a.act = nullptr; a.fpa = 1; a.H = a.W = 0; a.nframes = n;
    a.n = n; a.P = P; a.OH = (P + 6 - 7) / 2 + 1; a.OW = a.OH; a.PH = (a.OH - 1) / 2 + 1; a.PW = a.PH;
    if (adaf_stem7x7_rows_ok(P, n, cus) && launch_rows_for<0>(a, cus, s)) return;
    // rows of pooled pixels per tile: 4 (5 waves) or 6 (7 waves), whichever computes fewer conv pixels for this map
    const int cost4 = ((a.PH + 3) / 4) * PoolCfg<4>::NWV, cost6 = ((a.PH + 5) / 6) * PoolCfg<6>::NWV;
    if (cost6 < cost4) launch_pool<6>(a, cus, s);
    else launch_pool<4>(a, cus, s);
}
