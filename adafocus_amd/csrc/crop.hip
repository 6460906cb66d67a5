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
#include "adaf_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

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
// One thread per source pixel: reads T*3 contiguous bytes, writes one 16-byte pixel into each of the
// T frames (for a fixed t consecutive lanes write consecutive pixels -> coalesced).
__global__ void ingest_u8_kernel(const uint8_t* __restrict__ u8, long long pixels, int hw, int T, float m0, float m1,
                                 float m2, float s0, float s1, float s2, float* __restrict__ out) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= pixels) return;
    const long long clip = idx / hw;
    const int p = (int)(idx - clip * hw);
    const uint8_t* src = u8 + idx * (3 * T);
    float* dst = out + ((size_t)clip * T * hw + p) * 4;
    for (int t = 0; t < T; ++t) {
        f32x4 v;
        v.x = __fdiv_rn(__fsub_rn(__fdiv_rn((float)src[3 * t + 0], 255.f), m0), s0);
        v.y = __fdiv_rn(__fsub_rn(__fdiv_rn((float)src[3 * t + 1], 255.f), m1), s1);
        v.z = __fdiv_rn(__fsub_rn(__fdiv_rn((float)src[3 * t + 2], 255.f), m2), s2);
        v.w = 0.f;
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

}  // namespace

void adaf_launch_ingest_u8(const uint8_t* u8, int clips, int T, int H, int W, const float* mean, const float* stdv,
                           float* out, hipStream_t s) {
    const long long pixels = (long long)clips * H * W;
    hipLaunchKernelGGL(ingest_u8_kernel, dim3((unsigned)((pixels + 255) / 256)), dim3(256), 0, s, u8, pixels, H * W, T, mean[0],
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
