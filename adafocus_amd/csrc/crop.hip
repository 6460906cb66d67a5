// Batched patch gather (the reference's get_patch, ACT/models/utils.py:37-51) as one launch.
//
// The reference turns every fp32 action into integer window origins on the host with four
// .item() syncs per sample and issues one tiny slice kernel per frame.  Here a block owns
// RB rows of one patch: it recomputes the window origin from the action with the reference's
// exact fp32 sequence (multiply by float(H-P), floorf, truncate), pulls the rows of all
// channel planes with 16-byte aligned loads (the window's x origin is only 4-byte aligned, so
// the aligned chunks that cover it are fetched and trimmed), transposes planar -> pixel-major
// through LDS and writes the contiguous output with 16-byte stores.
//
// HBM-bound: algorithmic bytes per patch = 2 * C * P * P * 4 (read window + write patch).
#include <cstdlib>

#include "adaf_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int RB = 8;  // patch rows per block

__device__ __forceinline__ void window_origin(const float* act, int ai, int H, int W, int P, int& y0, int& x0,
                                              int& ry, int& rx) {
    // floor(action * (image_size - patch_size)).int() with image_size = H for BOTH axes
    // (ACT/models/utils.py:40-42).  __fmul_rn: a lone IEEE multiply, never contracted.
    const float span = (float)(H - P);
    ry = (int)floorf(__fmul_rn(act[2 * ai], span));
    rx = (int)floorf(__fmul_rn(act[2 * ai + 1], span));
    y0 = min(max(ry, 0), H - P);
    x0 = min(max(rx, 0), W - P);
}

// MODE 0: out NCHW, one (frame, channel) plane per blockIdx.x
// MODE 1: out NHWC with CO = C channels; MODE 2: out NHWC4 (C = 3, CO = 4, lane 3 zero)
template <int MODE, bool VEC>
__global__ __launch_bounds__(256) void crop_kernel(const float* __restrict__ frames, int C, int H, int W,
                                                   const float* __restrict__ act, int fpa, int P,
                                                   float* __restrict__ out, int32_t* __restrict__ coords) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int tid = threadIdx.x;
    const int r0 = blockIdx.y * RB;
    const int rows = min(RB, P - r0);
    const int frame = (MODE == 0) ? blockIdx.x / C : blockIdx.x;
    const int plane0 = (MODE == 0) ? blockIdx.x : frame * C;  // first source plane
    const int nplanes = (MODE == 0) ? 1 : C;
    const int CO = (MODE == 0) ? 1 : (MODE == 2 ? 4 : C);
    const int ai = frame / fpa;

    int y0, x0, ry, rx;
    window_origin(act, ai, H, W, P, y0, x0, ry, rx);
    if (coords && blockIdx.y == 0 && tid == 0 && frame == ai * fpa && (MODE != 0 || blockIdx.x == frame * C)) {
        coords[2 * ai] = ry;
        coords[2 * ai + 1] = rx;
    }

    // ---- load phase: planar rows -> LDS tile[(row*P + x)*CO + c]
    if (VEC) {
        const int xa = x0 & ~3;
        const int nch = ((x0 & 3) + P + 3) >> 2;  // aligned 16-byte chunks covering [x0, x0+P)
        const int total = nplanes * rows * nch;
        for (int idx = tid; idx < total; idx += 256) {
            const int c = idx / (rows * nch);
            const int rem = idx - c * (rows * nch);
            const int row = rem / nch;
            const int ch = rem - row * nch;
            const float* src = frames + ((size_t)(plane0 + c) * H + (y0 + r0 + row)) * W + xa + 4 * ch;
            const f32x4 v = *reinterpret_cast<const f32x4*>(src);
            const int xb = xa + 4 * ch - x0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int x = xb + e;
                if ((unsigned)x < (unsigned)P) tile[(row * P + x) * CO + c] = v[e];
            }
        }
    } else {
        const int total = nplanes * rows * P;
        for (int idx = tid; idx < total; idx += 256) {
            const int c = idx / (rows * P);
            const int rem = idx - c * (rows * P);
            const int row = rem / P;
            const int x = rem - row * P;
            tile[(row * P + x) * CO + c] = frames[((size_t)(plane0 + c) * H + (y0 + r0 + row)) * W + x0 + x];
        }
    }
    if (MODE == 2) {
        for (int idx = tid; idx < rows * P; idx += 256) tile[idx * 4 + 3] = 0.f;
    }
    __syncthreads();

    // ---- store phase: the block's output is one contiguous run of rows*P*CO floats
    const size_t obase = (MODE == 0) ? ((size_t)blockIdx.x * P + r0) * P : ((size_t)frame * P + r0) * (size_t)P * CO;
    const int nout = rows * P * CO;
    if (VEC && (nout & 3) == 0 && (obase & 3) == 0) {
        for (int idx = tid; idx < (nout >> 2); idx += 256)
            *reinterpret_cast<f32x4*>(out + obase + 4 * (size_t)idx) = *reinterpret_cast<const f32x4*>(tile + 4 * idx);
    } else {
        for (int idx = tid; idx < nout; idx += 256) out[obase + idx] = tile[idx];
    }
}

template <int MODE>
hipError_t launch_mode(const float* frames, int nf, int C, int H, int W, const float* act, int fpa, int P, float* out,
                       int32_t* coords, hipStream_t s) {
    const int CO = (MODE == 0) ? 1 : (MODE == 2 ? 4 : C);
    const size_t lds = (size_t)RB * P * CO * sizeof(float);
    const dim3 grid(MODE == 0 ? nf * C : nf, (P + RB - 1) / RB);  // x = plane / frame (large), y = row block
    const bool vec = (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(frames) & 15) == 0) &&
                     ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
    if (vec)
        hipLaunchKernelGGL((crop_kernel<MODE, true>), grid, dim3(256), lds, s, frames, C, H, W, act, fpa, P, out, coords);
    else
        hipLaunchKernelGGL((crop_kernel<MODE, false>), grid, dim3(256), lds, s, frames, C, H, W, act, fpa, P, out, coords);
    return hipGetLastError();
}

// ---- row f1: uint8 ingest ---------------------------------------------------------------------
// Loader layout (ACT/ops/transforms.py:305-315 Stack + :318-336 ToTorchFormatTensor): one clip =
// (H, W, T*3) uint8, frame t's RGB at channels 3t..3t+2.  Output: (clip*T + t, H, W, 4) fp32 with
//   v = ((float(u8) / 255) - mean[c]) / std[c]        (img.float().div(255); t.sub_(m).div_(s), :64-77)
// computed with IEEE fp32 divide/subtract in that order (bit-exact with the reference), lane 3 = 0.
// A block owns a run of 256 source pixels (256 * 3T contiguous bytes): the bytes are brought into LDS with 16-byte
// coalesced loads (the first version had every lane walk its own 3T bytes with byte loads: 0.42 of HBM), the 3 x 256
// possible results are tabulated once per block WITH THE REFERENCE'S OPERATIONS (so the table lookup is bit-exact by
// construction and the two IEEE divisions per value leave the inner loop), then thread p emits pixel p of every frame:
// for a fixed t consecutive lanes write consecutive 16-byte pixels.
__global__ __launch_bounds__(256) void ingest_u8_kernel(const uint8_t* __restrict__ u8, long long pixels, int hw, int T, float m0, float m1,
                                                        float m2, float s0, float s1, float s2, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ing_sm[];
    float* lut = reinterpret_cast<float*>(ing_sm);                    // [3][256]
    unsigned char* bytes = ing_sm + 3 * 256 * sizeof(float);         // [256][3T]
    const int tid = threadIdx.x;
    const long long p0 = (long long)blockIdx.x * 256;
    const int npx = (int)(pixels - p0 < 256 ? pixels - p0 : 256);
    const int row = 3 * T;
    {
        const float v = (float)tid;
        lut[tid] = __fdiv_rn(__fsub_rn(__fdiv_rn(v, 255.f), m0), s0);
        lut[256 + tid] = __fdiv_rn(__fsub_rn(__fdiv_rn(v, 255.f), m1), s1);
        lut[512 + tid] = __fdiv_rn(__fsub_rn(__fdiv_rn(v, 255.f), m2), s2);
    }
    const uint8_t* src = u8 + p0 * row;
    const int nbytes = npx * row;
    if ((reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const int nv = nbytes >> 4;
        for (int i = tid; i < nv; i += 256) reinterpret_cast<u32x4*>(bytes)[i] = reinterpret_cast<const u32x4*>(src)[i];
        for (int i = (nv << 4) + tid; i < nbytes; i += 256) bytes[i] = src[i];
    } else {
        for (int i = tid; i < nbytes; i += 256) bytes[i] = src[i];
    }
    __syncthreads();
    if (tid >= npx) return;
    const long long idx = p0 + tid;
    const long long clip = idx / hw;
    const int p = (int)(idx - clip * hw);
    float* dst = out + ((size_t)clip * T * hw + p) * 4;
    const unsigned char* mine = bytes + tid * row;
    for (int t = 0; t < T; ++t) {
        const f32x4 v = {lut[mine[3 * t]], lut[256 + mine[3 * t + 1]], lut[512 + mine[3 * t + 2]], 0.f};
        *reinterpret_cast<f32x4*>(dst + (size_t)t * hw * 4) = v;
    }
}

// Patch gather from pixel-major frames (N, H, W, 4) -> (N, P, P, 4): every patch row is one aligned
// run of P*16 bytes, so this is a strided 2-D copy with 16-byte accesses and no transpose.
__global__ __launch_bounds__(256) void crop_nhwc4_kernel(const float* __restrict__ frames, int H, int W,
                                                         const float* __restrict__ act, int fpa, int P,
                                                         float* __restrict__ out, int32_t* __restrict__ coords) {
    const int frame = blockIdx.x;
    const int r0 = blockIdx.y * RB;
    const int rows = min(RB, P - r0);
    const int ai = frame / fpa;
    int y0, x0, ry, rx;
    window_origin(act, ai, H, W, P, y0, x0, ry, rx);
    if (coords && blockIdx.y == 0 && threadIdx.x == 0 && frame == ai * fpa) {
        coords[2 * ai] = ry;
        coords[2 * ai + 1] = rx;
    }
    const float* src = frames + (((size_t)frame * H + y0 + r0) * W + x0) * 4;
    float* dst = out + ((size_t)frame * P + r0) * (size_t)P * 4;
    for (int idx = threadIdx.x; idx < rows * P; idx += 256) {
        const int row = idx / P, x = idx - row * P;
        *reinterpret_cast<f32x4*>(dst + (size_t)idx * 4) = *reinterpret_cast<const f32x4*>(src + ((size_t)row * W + x) * 4);
    }
}


// ---- N1: crop-and-resize ------------------------------------------------------------------------
// (y, x, size) -> P x P patch: the window [y0, y0+S) x [x0, x0+S) with (y0, x0) = floor(action * (H - S)) -- get_patch's
// own expression with patch_size = S (ACT/models/utils.py:40-42) -- resampled to P x P with the bilinear rule of
// torchvision.transforms.Resize on tensors / F.interpolate(mode='bilinear', align_corners=False), the transform the
// reference constructs as `self.down` (ACT/models/gfv_net.py:58, STH/models/gfv_net.py:69) and never calls:
//   scale = S / P;  s = max(scale * (o + 0.5) - 0.5, 0);  i0 = int(s);  i1 = min(i0 + 1, S - 1);  l = s - i0
//   out = (1-ly) * ((1-lx) * v00 + lx * v01) + ly * ((1-lx) * v10 + lx * v11)
// With S == P every l is exactly 0 and the taps with zero weight are not touched, so the kernel REDUCES BIT-EXACTLY to
// the slice copy (the launcher then runs crop_kernel itself when the size is uniform).
// A block owns `rbr` output rows of one frame: the source rows they touch are staged once in LDS (16-byte loads;
// planar frames are transposed on the way exactly like crop_kernel), every thread then produces whole output pixels.
// IN4: frames are pixel-major (N, H, W, 4) (adaf_ingest_u8_f32's output) instead of planar NCHW.
template <int MODE, bool IN4, bool VEC>
__global__ __launch_bounds__(256) void crop_resize_kernel(const float* __restrict__ frames, int C, int H, int W,
                                                          const float* __restrict__ act, const int32_t* __restrict__ sizes,
                                                          int size_default, int fpa, int P, int rbr, float* __restrict__ out,
                                                          int32_t* __restrict__ coords) {
    extern __shared__ __attribute__((aligned(16))) float tile[];
    const int tid = threadIdx.x;
    const int frame = blockIdx.x;
    const int r0 = blockIdx.y * rbr;
    const int rows = min(rbr, P - r0);
    const int ai = frame / fpa;
    const int S = min(max(sizes ? sizes[ai] : size_default, 1), H);
    int y0, x0, ry, rx;
    window_origin(act, ai, H, W, S, y0, x0, ry, rx);
    if (coords && blockIdx.y == 0 && tid == 0 && frame == ai * fpa) {
        coords[2 * ai] = ry;
        coords[2 * ai + 1] = rx;
    }
    const float scale = __fdiv_rn((float)S, (float)P);
    const float s_lo = fmaxf(__fsub_rn(__fmul_rn(scale, (float)r0 + 0.5f), 0.5f), 0.f);
    const float s_hi = fmaxf(__fsub_rn(__fmul_rn(scale, (float)(r0 + rows - 1) + 0.5f), 0.5f), 0.f);
    const int ylo = min((int)s_lo, S - 1);
    const int yhi = min((int)s_hi + 1, S - 1);
    const int nrows = yhi - ylo + 1;
    const int CI = IN4 ? 4 : C;

    // ---- stage source rows [ylo, yhi] of the window: IN4 tile[(row*S + x)*4 + c], planar tile[(c*nrows + row)*S + x]
    if (IN4) {
        const float* src = frames + (((size_t)frame * H + y0 + ylo) * W + x0) * 4;
        for (int idx = tid; idx < nrows * S; idx += 256) {
            const int row = idx / S, x = idx - row * S;
            *reinterpret_cast<f32x4*>(tile + (size_t)idx * 4) = *reinterpret_cast<const f32x4*>(src + ((size_t)row * W + x) * 4);
        }
    } else if (VEC) {
        const int xa = x0 & ~3;
        const int nch = ((x0 & 3) + S + 3) >> 2;
        const int total = C * nrows * nch;
        for (int idx = tid; idx < total; idx += 256) {
            const int c = idx / (nrows * nch);
            const int rem = idx - c * (nrows * nch);
            const int row = rem / nch;
            const int ch = rem - row * nch;
            const f32x4 v = *reinterpret_cast<const f32x4*>(frames + ((size_t)(frame * C + c) * H + (y0 + ylo + row)) * W + xa + 4 * ch);
            const int xb = xa + 4 * ch - x0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int x = xb + e;
                if ((unsigned)x < (unsigned)S) tile[(c * nrows + row) * S + x] = v[e];
            }
        }
    } else {
        const int total = C * nrows * S;
        for (int idx = tid; idx < total; idx += 256) {
            const int c = idx / (nrows * S);
            const int rem = idx - c * (nrows * S);
            const int row = rem / S;
            const int x = rem - row * S;
            tile[idx] = frames[((size_t)(frame * C + c) * H + (y0 + ylo + row)) * W + x0 + x];
        }
    }
    __syncthreads();

    auto tap = [&](int c, int row, int x) -> float { return IN4 ? tile[(row * S + x) * 4 + c] : tile[(c * nrows + row) * S + x]; };
    for (int idx = tid; idx < rows * P; idx += 256) {
        const int orow = idx / P, ox = idx - orow * P;
        const float sy = fmaxf(__fsub_rn(__fmul_rn(scale, (float)(r0 + orow) + 0.5f), 0.5f), 0.f);
        const float sx = fmaxf(__fsub_rn(__fmul_rn(scale, (float)ox + 0.5f), 0.5f), 0.f);
        const int iy0 = min((int)sy, S - 1), ix0 = min((int)sx, S - 1);
        const int iy1 = min(iy0 + 1, S - 1), ix1 = min(ix0 + 1, S - 1);
        const float ly = fminf(fmaxf(__fsub_rn(sy, (float)iy0), 0.f), 1.f), lx = fminf(fmaxf(__fsub_rn(sx, (float)ix0), 0.f), 1.f);
        const float hy = __fsub_rn(1.f, ly), hx = __fsub_rn(1.f, lx);
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < (MODE == 2 ? 3 : C); ++c) {
            const float a00 = tap(c, iy0 - ylo, ix0);
            float top = a00, bot;
            if (lx != 0.f) top = __fadd_rn(__fmul_rn(hx, a00), __fmul_rn(lx, tap(c, iy0 - ylo, ix1)));
            float r = top;
            if (ly != 0.f) {
                const float b00 = tap(c, iy1 - ylo, ix0);
                bot = b00;
                if (lx != 0.f) bot = __fadd_rn(__fmul_rn(hx, b00), __fmul_rn(lx, tap(c, iy1 - ylo, ix1)));
                r = __fadd_rn(__fmul_rn(hy, top), __fmul_rn(ly, bot));
            }
            if (MODE == 0) out[((size_t)(frame * C + c) * P + r0 + orow) * P + ox] = r;
            else if (MODE == 1) out[(((size_t)frame * P + r0 + orow) * P + ox) * C + c] = r;
            else v[c] = r;
        }
        if (MODE == 2) *reinterpret_cast<f32x4*>(out + (((size_t)frame * P + r0 + orow) * P + ox) * 4) = f32x4{v[0], v[1], v[2], 0.f};
    }
    (void)CI;
}

// ---- nearest-neighbour resize: F.interpolate(images, (g, g)) with the default mode, the glancer's input when
// glance_size != input_size (ACT/main_dist.py:331-332, STH/evaluate.py:188).  ATen's rule (UpSample.h
// nearest_neighbor_compute_source_index): src = min(int(floorf(dst * scale)), in - 1), scale = float(in) / out.
// A copy, hence bit-exact.  One thread per output pixel (all channels).
template <int MODE, bool IN4>
__global__ void resize_nearest_kernel(const float* __restrict__ frames, int C, int H, int W, int OH, int OW, long long pixels,
                                      float* __restrict__ out) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= pixels) return;
    const int ox = (int)(idx % OW);
    const int oy = (int)((idx / OW) % OH);
    const int frame = (int)(idx / ((long long)OW * OH));
    const float sh = __fdiv_rn((float)H, (float)OH), sw = __fdiv_rn((float)W, (float)OW);
    const int iy = min((int)floorf(__fmul_rn((float)oy, sh)), H - 1);
    const int ix = min((int)floorf(__fmul_rn((float)ox, sw)), W - 1);
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (IN4) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(frames + (((size_t)frame * H + iy) * W + ix) * 4);
        v[0] = p.x; v[1] = p.y; v[2] = p.z;
    }
    for (int c = 0; c < (MODE == 2 ? 3 : C); ++c) {
        const float r = IN4 ? v[c] : frames[((size_t)(frame * C + c) * H + iy) * W + ix];
        if (MODE == 0) out[((size_t)(frame * C + c) * OH + oy) * OW + ox] = r;
        else if (MODE == 1) out[(((size_t)frame * OH + oy) * OW + ox) * C + c] = r;
        else v[c] = r;
    }
    if (MODE == 2) *reinterpret_cast<f32x4*>(out + (((size_t)frame * OH + oy) * OW + ox) * 4) = f32x4{v[0], v[1], v[2], 0.f};
}

}  // namespace

void adaf_launch_ingest_u8(const uint8_t* u8, int clips, int T, int H, int W, const float* mean, const float* stdv,
                           float* out, hipStream_t s) {
    const long long pixels = (long long)clips * H * W;
    const size_t lds = 3 * 256 * sizeof(float) + (size_t)256 * 3 * T;
    hipLaunchKernelGGL(ingest_u8_kernel, dim3((unsigned)((pixels + 255) / 256)), dim3(256), lds, s, u8, pixels, H * W, T, mean[0],
                       mean[1], mean[2], stdv[0], stdv[1], stdv[2], out);
}

void adaf_launch_crop_nhwc4(const float* frames, int nf, int H, int W, const float* act, int fpa, int P, float* out,
                            int32_t* coords, hipStream_t s) {
    hipLaunchKernelGGL(crop_nhwc4_kernel, dim3(nf, (P + RB - 1) / RB), dim3(256), 0, s, frames, H, W, act, fpa, P, out, coords);
}

hipError_t adaf_launch_crop(const float* frames, int nf, int C, int H, int W, const float* act, int fpa, int P,
                            float* out, int layout, int32_t* coords, hipStream_t s) {
    switch (layout) {
        case ADAF_LAYOUT_NCHW: return launch_mode<0>(frames, nf, C, H, W, act, fpa, P, out, coords, s);
        case ADAF_LAYOUT_NHWC: return launch_mode<1>(frames, nf, C, H, W, act, fpa, P, out, coords, s);
        case ADAF_LAYOUT_NHWC4: return launch_mode<2>(frames, nf, C, H, W, act, fpa, P, out, coords, s);
    }
    return hipErrorInvalidValue;
}

// rows of output per block such that the staged source rows fit `budget` bytes of LDS for the largest window
static int resize_rows_per_block(int channels_in, int size_max, int P, size_t budget, size_t* lds_bytes) {
    const double scale = (double)size_max / P;
    int rbr = 16;
    for (; rbr > 1; --rbr) {
        const int nrows = (int)(scale * rbr) + 3;
        if ((size_t)nrows * size_max * channels_in * sizeof(float) <= budget) break;
    }
    int nrows = (int)(scale * rbr) + 3;
    if (nrows > size_max) nrows = size_max;
    *lds_bytes = (size_t)nrows * size_max * channels_in * sizeof(float);
    return rbr;
}

template <int MODE, bool IN4>
static hipError_t launch_resize_mode(const float* frames, int nf, int C, int H, int W, const float* act, const int32_t* sizes,
                                     int size_default, int fpa, int P, float* out, int32_t* coords, hipStream_t s) {
    const int smax = sizes ? H : (size_default < H ? size_default : H);
    size_t lds = 0;
    const size_t budget = (size_t)20 * 1024;      // staged rows per block of the resampling gather
    // (20 KB of staged source rows per block: measured against 12 / 32 / 60 KB on 1024 frames -- 3.5 / 3.8 / 3.4 TB/s at S = 128 / 192 /
    // mixed vs 3.1 / 2.2 / 1.9 with 60 KB: seven resident blocks per CU hide each other's load -> interpolate -> store phases)
    const int rbr = resize_rows_per_block(IN4 ? 4 : C, smax, P, budget, &lds);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    const dim3 grid(nf, (P + rbr - 1) / rbr);
    const bool vec = (W % 4 == 0) && ((reinterpret_cast<uintptr_t>(frames) & 15) == 0);
    auto go = [&](auto kern) -> hipError_t {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, frames, C, H, W, act, sizes, size_default, fpa, P, rbr, out, coords);
        return hipGetLastError();
    };
    if (IN4) return go(crop_resize_kernel<MODE, true, true>);
    return vec ? go(crop_resize_kernel<MODE, false, true>) : go(crop_resize_kernel<MODE, false, false>);
}

hipError_t adaf_launch_crop_resize(const float* frames, int in4, int nf, int C, int H, int W, const float* act, const int32_t* sizes,
                                   int size_default, int fpa, int P, float* out, int layout, int32_t* coords, hipStream_t s) {
    if (in4) {
        switch (layout) {
            case ADAF_LAYOUT_NCHW: return launch_resize_mode<0, true>(frames, nf, C, H, W, act, sizes, size_default, fpa, P, out, coords, s);
            case ADAF_LAYOUT_NHWC: return launch_resize_mode<1, true>(frames, nf, C, H, W, act, sizes, size_default, fpa, P, out, coords, s);
            case ADAF_LAYOUT_NHWC4: return launch_resize_mode<2, true>(frames, nf, C, H, W, act, sizes, size_default, fpa, P, out, coords, s);
        }
        return hipErrorInvalidValue;
    }
    switch (layout) {
        case ADAF_LAYOUT_NCHW: return launch_resize_mode<0, false>(frames, nf, C, H, W, act, sizes, size_default, fpa, P, out, coords, s);
        case ADAF_LAYOUT_NHWC: return launch_resize_mode<1, false>(frames, nf, C, H, W, act, sizes, size_default, fpa, P, out, coords, s);
        case ADAF_LAYOUT_NHWC4: return launch_resize_mode<2, false>(frames, nf, C, H, W, act, sizes, size_default, fpa, P, out, coords, s);
    }
    return hipErrorInvalidValue;
}

hipError_t adaf_launch_resize_nearest(const float* frames, int in4, int nf, int C, int H, int W, int OH, int OW, float* out,
                                      int layout, hipStream_t s) {
    const long long pixels = (long long)nf * OH * OW;
    const dim3 grid((unsigned)((pixels + 255) / 256));
#define ADAF_NEAREST(MODE, IN4) hipLaunchKernelGGL((resize_nearest_kernel<MODE, IN4>), grid, dim3(256), 0, s, frames, C, H, W, OH, OW, pixels, out)
    if (in4) {
        if (layout == ADAF_LAYOUT_NCHW) ADAF_NEAREST(0, true);
        else if (layout == ADAF_LAYOUT_NHWC) ADAF_NEAREST(1, true);
        else ADAF_NEAREST(2, true);
    } else {
        if (layout == ADAF_LAYOUT_NCHW) ADAF_NEAREST(0, false);
        else if (layout == ADAF_LAYOUT_NHWC) ADAF_NEAREST(1, false);
        else ADAF_NEAREST(2, false);
    }
#undef ADAF_NEAREST
    return hipGetLastError();
}
