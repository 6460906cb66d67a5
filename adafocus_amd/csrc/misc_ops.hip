// Bandwidth-bound companions of the conv engine: weight packing, BN folding, pooling,
// the stand-alone temporal shift, the GRU gate math and small glue copies.  All NHWC fp32,
// 16-byte accesses wherever the channel count allows.
#include "adaf_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

inline unsigned blocks_for(long long n, int per = 256) { return (unsigned)((n + per - 1) / per); }

// OIHW -> OHWI (+ zero pad of the I axis): w_ohwi[((o*KH+kh)*KW+kw)*cin_pad + i]
__global__ void pack_weight_kernel(const float* __restrict__ w, int cout, int cin, int kh, int kw, int cin_pad,
                                   float* __restrict__ o) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)cout * kh * kw * cin_pad;
    if (idx >= total) return;
    const int i = (int)(idx % cin_pad);
    long long t = idx / cin_pad;
    const int x = (int)(t % kw);
    t /= kw;
    const int y = (int)(t % kh);
    const int oc = (int)(t / kh);
    o[idx] = i < cin ? w[(((long long)oc * cin + i) * kh + y) * kw + x] : 0.f;
}

// the same with fp16 output (half-precision storage path, N2): round-to-nearest-even of the fp32 weight
__global__ void pack_weight_f16_kernel(const float* __restrict__ w, int cout, int cin, int kh, int kw, int cin_pad,
                                       _Float16* __restrict__ o) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)cout * kh * kw * cin_pad;
    if (idx >= total) return;
    const int i = (int)(idx % cin_pad);
    long long t = idx / cin_pad;
    const int x = (int)(t % kw);
    t /= kw;
    const int y = (int)(t % kh);
    const int oc = (int)(t / kh);
    o[idx] = (_Float16)(i < cin ? w[(((long long)oc * cin + i) * kh + y) * kw + x] : 0.f);
}

__global__ void cast_f32_f16_kernel(const float* __restrict__ x, long long count, _Float16* __restrict__ o) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < count) o[idx] = (_Float16)x[idx];
}

__global__ void cast_f16_f32_kernel(const _Float16* __restrict__ x, long long count, float* __restrict__ o) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < count) o[idx] = (float)x[idx];
}

// scale = gamma / sqrt(var + eps); bias = beta - mean * scale  (eval-mode BatchNorm)
__global__ void fold_bn_kernel(const float* g, const float* b, const float* m, const float* v, float eps, int c,
                               float* scale, float* bias) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c) return;
    const float s = g[i] / sqrtf(v[i] + eps);
    scale[i] = s;
    bias[i] = b[i] - m[i] * s;
}

// MaxPool2d(3, stride 2, pad 1), NHWC, 4 channels per thread; padding never wins (PyTorch pads
// with -inf), matching ACT/models/resnet.py:141.
__global__ void maxpool_kernel(const float* __restrict__ x, int n, int h, int w, int c4, int oh, int ow,
                               float* __restrict__ o) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)n * oh * ow * c4;
    if (idx >= total) return;
    const int cq = (int)(idx % c4);
    long long t = idx / c4;
    const int ox = (int)(t % ow);
    t /= ow;
    const int oy = (int)(t % oh);
    const int img = (int)(t / oh);
    const float ninf = -__builtin_inff();
    f32x4 best = {ninf, ninf, ninf, ninf};
    f32x4 v[9];   // unconditional loads from clamped addresses (a clamped tap duplicates a valid one: max is unaffected)
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = min(max(2 * oy - 1 + ky, 0), h - 1);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = min(max(2 * ox - 1 + kx, 0), w - 1);
            v[ky * 3 + kx] = *reinterpret_cast<const f32x4*>(x + (((size_t)img * h + iy) * w + ix) * (size_t)(4 * c4) + 4 * cq);
        }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        best.x = fmaxf(best.x, v[i].x);
        best.y = fmaxf(best.y, v[i].y);
        best.z = fmaxf(best.z, v[i].z);
        best.w = fmaxf(best.w, v[i].w);
    }
    *reinterpret_cast<f32x4*>(o + (size_t)idx * 4) = best;
}

// AdaptiveAvgPool2d(1): mean over hw pixels, NHWC, 4 channels per thread.
__global__ void avgpool_kernel(const float* __restrict__ x, int n, int hw, int c4, float* __restrict__ o, int ldo) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * c4) return;
    const int cq = idx % c4, img = idx / c4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const float* p = x + (size_t)img * hw * (4 * c4) + 4 * cq;
    for (int i = 0; i < hw; ++i) s += *reinterpret_cast<const f32x4*>(p + (size_t)i * (4 * c4));
    const float inv = (float)hw;
    f32x4 r = {s.x / inv, s.y / inv, s.z / inv, s.w / inv};
    *reinterpret_cast<f32x4*>(o + (size_t)img * ldo + 4 * cq) = r;
}

// TemporalShift.shift -- STH/ops/temporal_shift.py:28-46.  One thread per element.
__global__ void tshift_kernel(const float* __restrict__ x, long long total, int c, int hw, int T, int fold, int nhwc,
                              float* __restrict__ o) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const long long frame_elems = (long long)c * hw;
    const long long f = idx / frame_elems;
    const int within = (int)(idx - f * frame_elems);
    const int ch = nhwc ? within % c : within / hw;
    const int t = (int)(f % T);
    float v;
    if (ch < fold) v = (t < T - 1) ? x[idx + frame_elems] : 0.f;          // from t+1
    else if (ch < 2 * fold) v = (t > 0) ? x[idx - frame_elems] : 0.f;     // from t-1
    else v = x[idx];
    o[idx] = v;
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.f / (1.f + expf(-v)); }

// One GRU step's gate math (PyTorch order r,z,n):  gi = W_ih x_t + b_ih (precomputed for all t),
// gh = W_hh h_{t-1} (without bias; null at t = 0 where h = 0), bhh = b_hh.
//   r = s(gi_r + gh_r + bhh_r); z = s(gi_z + gh_z + bhh_z); n = tanh(gi_n + r*(gh_n + bhh_n)); h' = (1-z)n + z h
__global__ void gru_gates_kernel(const float* __restrict__ gi, int ldgi, const float* __restrict__ gh,
                                 const float* __restrict__ bhh, const float* __restrict__ hprev, int ldh,
                                 float* __restrict__ hout, int ldo, int B, int Hd) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * Hd) return;
    const int b = idx / Hd, j = idx - b * Hd;
    const float* gir = gi + (size_t)b * ldgi;
    float hr = bhh[j], hz = bhh[Hd + j], hn = bhh[2 * Hd + j];
    float hp = 0.f;
    if (gh) {
        const float* ghr = gh + (size_t)b * 3 * Hd;
        hr += ghr[j];
        hz += ghr[Hd + j];
        hn += ghr[2 * Hd + j];
        hp = hprev[(size_t)b * ldh + j];
    }
    const float r = sigmoidf_(gir[j] + hr);
    const float z = sigmoidf_(gir[Hd + j] + hz);
    const float nn = tanhf(gir[2 * Hd + j] + r * hn);
    hout[(size_t)b * ldo + j] = (1.f - z) * nn + z * hp;
}

// out[b,c] = mean_t logit[b,t,c] (+ mean_t glog[b,t,c])  -- ConsensusModule('avg'), STH/ops/basic_ops.py:17-26
__global__ void segment_mean_kernel(const float* __restrict__ logit, int B, int T, int C,
                                    const float* __restrict__ glog, int Tg, float* __restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * C) return;
    const int b = idx / C, c = idx - b * C;
    float s = 0.f;
    for (int t = 0; t < T; ++t) s += logit[((size_t)b * T + t) * C + c];
    float r = s / (float)T;
    if (glog) {
        float g = 0.f;
        for (int t = 0; t < Tg; ++t) g += glog[((size_t)b * Tg + t) * C + c];
        r = g / (float)Tg + r;
    }
    out[idx] = r;
}

// [C,1,3,3] (PyTorch depthwise) -> [3][3][C]
__global__ void pack_dw_weight_kernel(const float* __restrict__ w, int c, float* __restrict__ o) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 9 * c) return;
    const int ch = idx % c, tap = idx / c;
    o[idx] = w[ch * 9 + tap];
}

// Depthwise 3x3 (pad 1, stride 1|2) + BN affine + ReLU6|ReLU|none, NHWC, 4 channels x PX output pixels per thread.
// HBM/L2-bound VALU work (no contraction across channels, so the matrix cores have nothing to do): taps are
// aligned 16-byte loads that neighbouring lanes (adjacent channel groups) coalesce; a thread that owns PX
// horizontally adjacent outputs loads each input column once (stride 1: PX+2 columns instead of 3*PX).
// T = float | _Float16: element type of the activations in HBM (N2: half-precision storage; the taps, the BN affine and
// the accumulation stay fp32 either way).
template <typename T> struct Vec4;
template <> struct Vec4<float> {
    static __device__ __forceinline__ f32x4 ld(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
    static __device__ __forceinline__ void st(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
};
template <> struct Vec4<_Float16> {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ f32x4 ld(const _Float16* p) {
        const h4 v = *reinterpret_cast<const h4*>(p);
        return f32x4{(float)v.x, (float)v.y, (float)v.z, (float)v.w};
    }
    static __device__ __forceinline__ void st(_Float16* p, f32x4 v) {
        *reinterpret_cast<h4*>(p) = h4{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    }
};

template <int PX, int S, int RY, typename T = float>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 3))) void dwconv3x3_kernel(const T* __restrict__ x, int n, int h, int w, int c4, int oh, int ow,
                                 const float* __restrict__ wt, const float* __restrict__ scale,
                                 const float* __restrict__ bias, float lo, float hi, T* __restrict__ o) {
    // a thread owns PX horizontally adjacent outputs of RY consecutive output rows: the S*(RY-1)+3 input rows and
    // S*(PX-1)+3 input columns they touch are loaded once
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int owg = (ow + PX - 1) / PX, ohg = (oh + RY - 1) / RY;
    const long long total = (long long)n * ohg * owg * c4;
    if (idx >= total) return;
    const int cq = (int)(idx % c4);
    long long t = idx / c4;
    const int oxg = (int)(t % owg);
    t /= owg;
    const int oy0 = (int)(t % ohg) * RY;
    const int img = (int)(t / ohg);
    const int c = 4 * c4;
    const int ox0 = oxg * PX;
    constexpr int NC = S * (PX - 1) + 3;   // input columns the PX outputs touch
    constexpr int NR = S * (RY - 1) + 3;   // input rows the RY output rows touch
    // every tap is loaded unconditionally from a clamped (always valid) address and zeroed by a select, so
    // the NR x NC loads of a thread are independent and issue back to back (no branch, no wait in between)
    f32x4 v[NR][NC];
#pragma unroll
    for (int ky = 0; ky < NR; ++ky) {
        const int iy = oy0 * S - 1 + ky;
        const bool rv = (unsigned)iy < (unsigned)h;
        const T* row = x + (((size_t)img * h + min(max(iy, 0), h - 1)) * w) * (size_t)c + 4 * cq;
#pragma unroll
        for (int ci = 0; ci < NC; ++ci) {
            const int ix = ox0 * S - 1 + ci;
            const bool ok = rv && (unsigned)ix < (unsigned)w;
            const f32x4 ld = Vec4<T>::ld(row + (size_t)min(max(ix, 0), w - 1) * c);
            v[ky][ci] = ok ? ld : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    f32x4 acc[RY][PX];
#pragma unroll
    for (int r = 0; r < RY; ++r)
#pragma unroll
        for (int p = 0; p < PX; ++p) acc[r][p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const f32x4 k = *reinterpret_cast<const f32x4*>(wt + (ky * 3 + kx) * c + 4 * cq);
#pragma unroll
            for (int r = 0; r < RY; ++r)
#pragma unroll
                for (int p = 0; p < PX; ++p) {
                    const f32x4 a = v[r * S + ky][p * S + kx];
                    acc[r][p].x = fmaf(a.x, k.x, acc[r][p].x);
                    acc[r][p].y = fmaf(a.y, k.y, acc[r][p].y);
                    acc[r][p].z = fmaf(a.z, k.z, acc[r][p].z);
                    acc[r][p].w = fmaf(a.w, k.w, acc[r][p].w);
                }
        }
    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + 4 * cq);
    const f32x4 bi = *reinterpret_cast<const f32x4*>(bias + 4 * cq);
#pragma unroll
    for (int rr = 0; rr < RY; ++rr) {
        const int oy = oy0 + rr;
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            const int ox = ox0 + p;
            if (ox < ow && oy < oh) {
                f32x4 r;
                r.x = fminf(fmaxf(fmaf(acc[rr][p].x, sc.x, bi.x), lo), hi);
                r.y = fminf(fmaxf(fmaf(acc[rr][p].y, sc.y, bi.y), lo), hi);
                r.z = fminf(fmaxf(fmaf(acc[rr][p].z, sc.z, bi.z), lo), hi);
                r.w = fminf(fmaxf(fmaf(acc[rr][p].w, sc.w, bi.w), lo), hi);
                Vec4<T>::st(o + ((((size_t)img * oh + oy) * ow + ox) * (size_t)c + 4 * cq), r);
            }
        }
    }
}

// Discrete policy head: idx = argmax_a logits[row, a] (first maximum, like Tensor.max(1)[1] --
// softmax is monotone so the reference's argmax over probabilities is the argmax over logits),
// action = table[idx]  (ACT/models/ppo.py:94, gfv_net.py:345-347).  One thread per row.
__global__ void grid_actions_kernel(const float* __restrict__ logits, int rows, int a, const float* __restrict__ table,
                                    long long* __restrict__ idx_out, float* __restrict__ act_out) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= rows) return;
    const float* p = logits + (size_t)r * a;
    int best = 0;
    float bv = p[0];
    for (int i = 1; i < a; ++i)
        if (p[i] > bv) { bv = p[i]; best = i; }
    if (idx_out) idx_out[r] = best;
    if (act_out) {
        act_out[2 * r] = table[2 * best];
        act_out[2 * r + 1] = table[2 * best + 1];
    }
}

__global__ void copy2d_kernel(const float* __restrict__ src, int lds, float* __restrict__ dst, int ldd, int rows,
                              int cols) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)rows * cols) return;
    const int r = (int)(idx / cols), c = (int)(idx - (long long)r * cols);
    dst[(size_t)r * ldd + c] = src[(size_t)r * lds + c];
}

}  // namespace

// fp32 -> three bf16 planes (h, m, l) with h + m + l == w exactly (round-to-nearest split, the same arithmetic as
// split3 in conv_gemm.hip so pre-split and on-the-fly operands are identical)
__global__ void split_weight_kernel(const float* __restrict__ w, size_t count, unsigned short* __restrict__ planes) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float x = w[i];
    const __bf16 hb = (__bf16)x;
    const float r1 = x - (float)hb;
    const __bf16 mb = (__bf16)r1;
    const float r2 = r1 - (float)mb;
    planes[i] = __builtin_bit_cast(unsigned short, hb);
    planes[count + i] = __builtin_bit_cast(unsigned short, mb);
    planes[2 * count + i] = (unsigned short)(__float_as_uint(r2) >> 16);
}

void adaf_launch_split_weight(const float* w, size_t count, unsigned short* planes, hipStream_t s) {
    hipLaunchKernelGGL(split_weight_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, w, count, planes);
}

void adaf_launch_pack_weight(const float* w, int cout, int cin, int kh, int kw, int cin_pad, float* o, hipStream_t s) {
    const long long total = (long long)cout * kh * kw * cin_pad;
    hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks_for(total)), dim3(256), 0, s, w, cout, cin, kh, kw, cin_pad, o);
}

void adaf_launch_fold_bn(const float* g, const float* b, const float* m, const float* v, float eps, int c,
                         float* scale, float* bias, hipStream_t s) {
    hipLaunchKernelGGL(fold_bn_kernel, dim3(blocks_for(c)), dim3(256), 0, s, g, b, m, v, eps, c, scale, bias);
}

void adaf_launch_maxpool(const float* x, int n, int h, int w, int c, float* o, hipStream_t s) {
    const int oh = (h + 2 - 3) / 2 + 1, ow = (w + 2 - 3) / 2 + 1;
    const long long total = (long long)n * oh * ow * (c / 4);
    hipLaunchKernelGGL(maxpool_kernel, dim3(blocks_for(total)), dim3(256), 0, s, x, n, h, w, c / 4, oh, ow, o);
}

void adaf_launch_avgpool(const float* x, int n, int hw, int c, float* o, int ldo, hipStream_t s) {
    hipLaunchKernelGGL(avgpool_kernel, dim3(blocks_for((long long)n * (c / 4))), dim3(256), 0, s, x, n, hw, c / 4, o, ldo);
}

void adaf_launch_tshift(const float* x, int nt, int c, int hw, int T, int div, int layout, float* o, hipStream_t s) {
    const long long total = (long long)nt * c * hw;
    hipLaunchKernelGGL(tshift_kernel, dim3(blocks_for(total)), dim3(256), 0, s, x, total, c, hw, T, c / div,
                       layout == ADAF_LAYOUT_NHWC ? 1 : 0, o);
}

void adaf_launch_gru_gates(const float* gi, int ldgi, const float* gh, const float* bhh, const float* hprev, int ldh,
                           float* hout, int ldo, int B, int Hd, hipStream_t s) {
    hipLaunchKernelGGL(gru_gates_kernel, dim3(blocks_for((long long)B * Hd)), dim3(256), 0, s, gi, ldgi, gh, bhh, hprev,
                       ldh, hout, ldo, B, Hd);
}

void adaf_launch_segment_mean(const float* logit, int B, int T, int C, const float* glog, int Tg, float* out,
                              hipStream_t s) {
    hipLaunchKernelGGL(segment_mean_kernel, dim3(blocks_for((long long)B * C)), dim3(256), 0, s, logit, B, T, C, glog, Tg,
                       out);
}

void adaf_launch_pack_dw_weight(const float* w, int c, float* o, hipStream_t s) {
    hipLaunchKernelGGL(pack_dw_weight_kernel, dim3(blocks_for(9LL * c)), dim3(256), 0, s, w, c, o);
}

void adaf_launch_dwconv3x3(const float* x, int n, int h, int w, int c, int stride, const float* wt, const float* scale,
                           const float* bias, int act, float* o, hipStream_t s) {
    const int oh = (h + 2 - 3) / stride + 1, ow = (w + 2 - 3) / stride + 1;
    const float lo = act == ADAF_ACT_NONE ? -__builtin_inff() : 0.f;
    const float hi = act == ADAF_ACT_RELU6 ? 6.f : __builtin_inff();
    // thread tile at stride 1: 4 x 4 outputs share 6 x 6 input chunks (2.25 loads per output; measured on the glancer's maps at
    // 512 frames: 4x2 (3 loads per output, round 2's shape) 3.7-4.4 TB/s, 2x2 3.2-4.2, 4x1 3.0-3.6, 7x2 3.8-4.5,
    // 4x4 4.3-4.9 -- fewer, fatter threads with 36 independent loads in flight each win although 14 = 3.5 x 4 wastes an eighth of them)
    if (stride == 1) {
        const long long total = (long long)n * ((oh + 3) / 4) * ((ow + 3) / 4) * (c / 4);
        hipLaunchKernelGGL((dwconv3x3_kernel<4, 1, 4>), dim3(blocks_for(total)), dim3(256), 0, s, x, n, h, w, c / 4, oh, ow, wt, scale, bias, lo, hi, o);
    } else {             // 2 outputs share 5 input columns (more would spill the tap registers)
        const long long total = (long long)n * oh * ((ow + 1) / 2) * (c / 4);
        hipLaunchKernelGGL((dwconv3x3_kernel<2, 2, 1>), dim3(blocks_for(total)), dim3(256), 0, s, x, n, h, w, c / 4, oh, ow, wt,
                           scale, bias, lo, hi, o);
    }
}

void adaf_launch_grid_actions(const float* logits, int rows, int a, const float* table, long long* idx, float* act,
                              hipStream_t s) {
    hipLaunchKernelGGL(grid_actions_kernel, dim3(blocks_for(rows)), dim3(256), 0, s, logits, rows, a, table, idx, act);
}

void adaf_launch_copy2d(const float* src, int lds, float* dst, int ldd, int rows, int cols, hipStream_t s) {
    hipLaunchKernelGGL(copy2d_kernel, dim3(blocks_for((long long)rows * cols)), dim3(256), 0, s, src, lds, dst, ldd, rows,
                       cols);
}

// ---- half-precision storage (N2) ------------------------------------------------------------------------------
void adaf_launch_pack_weight_f16(const float* w, int cout, int cin, int kh, int kw, int cin_pad, void* o, hipStream_t s) {
    const long long total = (long long)cout * kh * kw * cin_pad;
    hipLaunchKernelGGL(pack_weight_f16_kernel, dim3(blocks_for(total)), dim3(256), 0, s, w, cout, cin, kh, kw, cin_pad,
                       static_cast<_Float16*>(o));
}

void adaf_launch_cast(const void* x, long long count, void* o, int to_f16, hipStream_t s) {
    if (to_f16)
        hipLaunchKernelGGL(cast_f32_f16_kernel, dim3(blocks_for(count)), dim3(256), 0, s, static_cast<const float*>(x), count,
                           static_cast<_Float16*>(o));
    else
        hipLaunchKernelGGL(cast_f16_f32_kernel, dim3(blocks_for(count)), dim3(256), 0, s, static_cast<const _Float16*>(x), count,
                           static_cast<float*>(o));
}

void adaf_launch_dwconv3x3_f16(const void* x, int n, int h, int w, int c, int stride, const float* wt, const float* scale,
                               const float* bias, int act, void* o, hipStream_t s) {
    const int oh = (h + 2 - 3) / stride + 1, ow = (w + 2 - 3) / stride + 1;
    const float lo = act == ADAF_ACT_NONE ? -__builtin_inff() : 0.f;
    const float hi = act == ADAF_ACT_RELU6 ? 6.f : __builtin_inff();
    const _Float16* xi = static_cast<const _Float16*>(x);
    _Float16* oo = static_cast<_Float16*>(o);
    if (stride == 1) {
        const long long total = (long long)n * ((oh + 3) / 4) * ((ow + 3) / 4) * (c / 4);
        hipLaunchKernelGGL((dwconv3x3_kernel<4, 1, 4, _Float16>), dim3(blocks_for(total)), dim3(256), 0, s, xi, n, h, w, c / 4, oh, ow, wt,
                           scale, bias, lo, hi, oo);
    } else {
        const long long total = (long long)n * oh * ((ow + 1) / 2) * (c / 4);
        hipLaunchKernelGGL((dwconv3x3_kernel<2, 2, 1, _Float16>), dim3(blocks_for(total)), dim3(256), 0, s, xi, n, h, w, c / 4, oh, ow, wt,
                           scale, bias, lo, hi, oo);
    }
}
