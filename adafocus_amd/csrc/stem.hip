// ResNet stem: conv 7x7 / stride 2 / pad 3, 3 -> 64 channels, + BN affine + ReLU (ACT/models/resnet.py:138-141,
// 212-214), specialised because the generic engine wastes a third of its MFMAs on it (K = 49 taps x 4 padded
// channels = 196 -> 224 after slice padding, of which 147 are real).
//
// A block owns an 8 x 16 tile of output pixels (x 64 channels) and loops over tiles (persistent), keeping the
// whole filter bank in LDS.  Per tile the 22 x 37 x 3 input window is staged once in LDS (the zero 4th lane
// of the NHWC4 patch is dropped); MFMA A fragments are read straight from that window -- for kernel row kh
// the 7 x 3 values (kw, c) of a pixel are 21 CONTIGUOUS floats, so step j of the K walk is one ds_read_b32 at
// a compile-time offset from the lane's pixel base.  The two k-lanes of v_mfma_f32_32x32x2_f32 take kernel
// rows 0-3 (lanes 0-31) and 4-7 (lanes 32-63; row 7 is zero weights), i.e. K = 2 x 84 = 168, 87.5 % useful.
// Any k order is exact in fp32 as long as both operands use it.
#include "adaf_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TH = 8, TW = 16;            // output tile
constexpr int RH = 2 * TH + 6, RW = 2 * TW + 5;   // staged input rows (21 + 1 spare for the padded kernel row) / cols
constexpr int RP = RW * 3;                // floats per staged row
constexpr int KS = 84;                    // MFMA steps (k pairs)

// w_oihw [64][3][7][7] -> Wr[h][j][n]: h = 0 -> kernel rows 0..3, h = 1 -> rows 4..7 (7 = zero); j = krow*21 + kw*3 + c
__global__ void pack_stem_weight_kernel(const float* __restrict__ w, float* __restrict__ o) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 2 * KS * 64) return;
    const int n = idx & 63, j = (idx >> 6) % KS, h = idx / (KS * 64);
    const int kh = 4 * h + j / 21, r = j % 21, kw = r / 3, c = r % 3;
    o[idx] = kh < 7 ? w[((n * 3 + c) * 7 + kh) * 7 + kw] : 0.f;
}

struct StemArgs {
    const float* x;      // [n, P, P, 4]
    const float* w;      // Wr [2][84][64]
    const float* scale;  // [64]
    const float* bias;
    float* out;          // [n, OH, OW, 64]
    int n, P, OH, OW, tiles_y, tiles_x, ntiles;
};

__global__ __launch_bounds__(256) void stem7x7_kernel(const StemArgs a) {
    __shared__ __attribute__((aligned(16))) float Ws[2 * KS * 64];
    __shared__ __attribute__((aligned(16))) float reg[RH * RP + 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 2 * KS * 64 / 4; i += 256)
        reinterpret_cast<f32x4*>(Ws)[i] = reinterpret_cast<const f32x4*>(a.w)[i];

    const int h = lane >> 5, li = lane & 31;
    const int py = 2 * wave + (li >> 4), px = li & 15;          // this lane's output pixel inside the tile (as an A row)
    const float* abase = reg + ((2 * py + 4 * h) * RW + 2 * px) * 3;
    const float* bbase = Ws + h * KS * 64 + li;
    const float sc0 = a.scale[li], sc1 = a.scale[32 + li], bi0 = a.bias[li], bi1 = a.bias[32 + li];

    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        const int per_img = a.tiles_y * a.tiles_x;
        const int img = tile / per_img, rem = tile - img * per_img;
        const int ty = rem / a.tiles_x, tx = rem - ty * a.tiles_x;
        const int oy0 = ty * TH, ox0 = tx * TW;
        const int iy0 = 2 * oy0 - 3, ix0 = 2 * ox0 - 3;
        __syncthreads();   // previous tile's window fully consumed (and Ws written, first time round)
        for (int idx = tid; idx < RH * RW; idx += 256) {
            const int r = idx / RW, c = idx - r * RW;
            const int iy = iy0 + r, ix = ix0 + c;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)iy < (unsigned)a.P && (unsigned)ix < (unsigned)a.P)
                v = *reinterpret_cast<const f32x4*>(a.x + (((size_t)img * a.P + iy) * a.P + ix) * 4);
            float* d = reg + idx * 3;
            d[0] = v.x; d[1] = v.y; d[2] = v.z;
        }
        __syncthreads();

        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            const float av = abase[(j / 21) * RP + (j % 21)];
            const float b0 = bbase[j * 64], b1 = bbase[j * 64 + 32];
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc1, 0, 0, 0);
        }
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

}  // namespace

void adaf_launch_pack_stem_weight(const float* w_oihw, float* wr, hipStream_t s) {
    hipLaunchKernelGGL(pack_stem_weight_kernel, dim3((2 * KS * 64 + 255) / 256), dim3(256), 0, s, w_oihw, wr);
}

size_t adaf_stem_weight_floats() { return (size_t)2 * KS * 64; }

void adaf_launch_stem7x7(const float* x4, int n, int P, const float* wr, const float* scale, const float* bias, float* out,
                         int cus, hipStream_t s) {
    StemArgs a;
    a.x = x4; a.w = wr; a.scale = scale; a.bias = bias; a.out = out;
    a.n = n; a.P = P; a.OH = (P + 6 - 7) / 2 + 1; a.OW = a.OH;
    a.tiles_y = (a.OH + TH - 1) / TH; a.tiles_x = (a.OW + TW - 1) / TW;
    a.ntiles = n * a.tiles_y * a.tiles_x;
    const int grid = a.ntiles < cus * 3 ? a.ntiles : cus * 3;
    hipLaunchKernelGGL(stem7x7_kernel, dim3(grid), dim3(256), 0, s, a);
}
