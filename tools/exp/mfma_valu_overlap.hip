// Does a matrix-pipe wave run beside a vector-pipe wave on the same SIMD of gfx950?  (round 6)
// DESIGN 3.4 concluded from an additive ablation of a fused kernel that "fp32 MFMA does not run beside fp32 VALU work of another wave";
// MI355X_MICROARCH.md says the pipes are separate.  This is the direct measurement: one workgroup per CU (150 KB LDS request), waves
// 0-3 (one per SIMD) run an MFMA loop, waves 4-7 (the second wave of each SIMD) a packed-FMA loop; every wave stamps its own s_memtime.
//   alone:  only one role is launched (the other role's waves exit at once)        -> cycles per instruction of the role by itself
//   paired: both roles, iteration counts chosen so both would take the same time alone -> the overlap factor
//   same-wave: one wave issues 1 MFMA + k independent packed FMAs per iteration    -> what a single wave can hide under its own MFMA
// hipcc --offload-arch=gfx950 -O3 tools/exp/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MK: 0 = v_mfma_f32_32x32x2_f32 (16 passes), 1 = v_mfma_f32_32x32x16_bf16 (8 passes)
template <int MK>
__device__ __forceinline__ void mfma_loop(int iters, float seed, float* sink) {
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) { c0[r] = seed; c1[r] = seed + 1; c2[r] = seed + 2; c3[r] = seed + 3; }
    const float a = seed * 1e-3f, b = seed * 2e-3f;
    f32x4 av = {a, a, a, a}, bv = {b, b, b, b};
    for (int it = 0; it < iters; ++it) {
        if (MK == 0) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
        } else {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), c3, 0, 0, 0);
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    if (s == 12345.f) *sink = s;
}

// VK: 0 = v_pk_fma_f32, 1 = v_fma_f32, 2 = ds_read_b128 stream (LDS pipe instead of the VALU)
template <int VK>
__device__ __forceinline__ void valu_loop(int iters, float seed, float* sink, const float* lds) {
    f32x2 b[16];
    for (int i = 0; i < 16; ++i) b[i] = f32x2{seed + i, seed - i};
    const f32x2 w = {seed * 0.999f, seed * 1.0001f};
    const f32x2 z = {0.5f, 0.25f};
    for (int it = 0; it < iters; ++it) {
        if (VK == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) b[i] = __builtin_elementwise_fma(b[i], w, z);
        } else if (VK == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) b[i].x = __builtin_fmaf(b[i].x, w.x, 0.5f);
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const f32x4 v = *reinterpret_cast<const volatile f32x4*>(lds + 4 * ((threadIdx.x + 64 * i + it) & 1023));
                b[i].x += v.x;
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += b[i].x + b[i].y;
    if (s == 12345.f) *sink = s;
}

template <int MK, int VK>
__global__ __launch_bounds__(512) void roles(float* out, long long* cyc, int mfma_iters, int valu_iters, float seed) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) dsm[i] = seed * i;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    const long long t0 = __builtin_readcyclecounter();
    if (wave < 4) { if (mfma_iters > 0) mfma_loop<MK>(mfma_iters, seed, out); }
    else if (valu_iters > 0) valu_loop<VK>(valu_iters, seed, out, dsm);
    const long long t1 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

// one wave: per iteration 2 MFMAs (two accumulators) + NV independent packed FMAs
template <int MK, int NV>
__global__ __launch_bounds__(256) void same_wave(float* out, long long* cyc, int iters, float seed) {
    extern __shared__ __attribute__((aligned(16))) float dsm[];
    f32x16 c0, c1;
    for (int r = 0; r < 16; ++r) { c0[r] = seed; c1[r] = seed + 1; }
    f32x2 b[16];
    for (int i = 0; i < 16; ++i) b[i] = f32x2{seed + i, seed - i};
    const f32x2 w = {seed * 0.999f, seed * 1.0001f};
    const f32x2 z = {0.5f, 0.25f};
    const float a = seed * 1e-3f, bb = seed * 2e-3f;
    f32x4 av = {a, a, a, a}, bv = {bb, bb, bb, bb};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (MK == 0) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, c1, 0, 0, 0);
        } else {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bv), c1, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) b[i % 16] = __builtin_elementwise_fma(b[i % 16], w, z);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
    for (int i = 0; i < 16; ++i) s += b[i].x + b[i].y;
    if (s == 12345.f) out[0] = s + dsm[0];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

static double ms_of(hipEvent_t e0, hipEvent_t e1) { float ms; hipEventElapsedTime(&ms, e0, e1); return ms; }

template <int MK, int VK>
void run_roles(const char* mname, const char* vname, double clk_per_mfma, double clk_per_valu, float* d, long long* dc) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&roles<MK, VK>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int MI = 20000;                                    // x4 MFMAs per iteration
    const int VI = (int)(MI * 4 * clk_per_mfma / (16 * clk_per_valu));   // x16 vector instructions per iteration: same time alone
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<long long> h(256 * 8);
    double res[3][3];
    const int cfg[3][2] = {{MI, 0}, {0, VI}, {MI, VI}};
    for (int c = 0; c < 3; ++c) {
        hipLaunchKernelGGL((roles<MK, VK>), dim3(256), dim3(512), 150 * 1024, 0, d, dc, cfg[c][0], cfg[c][1], 1.0001f);
        hipEventRecord(e0);
        hipLaunchKernelGGL((roles<MK, VK>), dim3(256), dim3(512), 150 * 1024, 0, d, dc, cfg[c][0], cfg[c][1], 1.0001f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipMemcpy(h.data(), dc, h.size() * 8, hipMemcpyDeviceToHost);
        double m = 0, v = 0;
        for (int b = 0; b < 256; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : v) += (double)h[b * 8 + w] / (256 * 4);
        res[c][0] = ms_of(e0, e1); res[c][1] = m; res[c][2] = v;
    }
    // s_memtime ticks at a constant 100 MHz on gfx9: report times from the wall clock and ratios from the ticks
    printf("%-22s + %-14s  alone: mfma %.3f ms (%.1f clk/MFMA @2.4GHz), vector %.3f ms (%.2f clk/instr);  paired: wall %.3f ms, mfma waves x%.2f, "
           "vector waves x%.2f of their time alone  -> overlap %s\n",
           mname, vname, res[0][0], res[0][0] * 1e-3 * 2.4e9 / (MI * 4.0), res[1][0], res[1][0] * 1e-3 * 2.4e9 / (VI * 16.0), res[2][0],
           res[2][1] / res[0][1], res[2][2] / res[1][2], res[2][0] < 0.6 * (res[0][0] + res[1][0]) ? "YES" : res[2][0] < 0.85 * (res[0][0] + res[1][0]) ? "partial" : "NO (additive)");
}

template <int MK, int NV>
void run_same(const char* mname, float* d, long long* dc) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 40000;
    hipLaunchKernelGGL((same_wave<MK, NV>), dim3(256), dim3(256), 150 * 1024, 0, d, dc, iters, 1.0001f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((same_wave<MK, NV>), dim3(256), dim3(256), 150 * 1024, 0, d, dc, iters, 1.0001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    printf("  same wave, %-22s 2 MFMA + %2d v_pk_fma_f32 per iteration: %.1f clk per iteration @2.4GHz\n", mname, NV, ms_of(e0, e1) * 1e-3 * 2.4e9 / iters);
}

int main() {
    float* d; long long* dc;
    hipMalloc(&d, 64); hipMalloc(&dc, 256 * 8 * 8);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&same_wave<0, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    run_roles<0, 0>("v_mfma_f32_32x32x2_f32", "v_pk_fma_f32", 64, 4, d, dc);
    run_roles<0, 1>("v_mfma_f32_32x32x2_f32", "v_fma_f32", 64, 4, d, dc);
    run_roles<0, 2>("v_mfma_f32_32x32x2_f32", "ds_read_b128", 64, 16, d, dc);
    run_roles<1, 0>("v_mfma_f32_32x32x16_bf16", "v_pk_fma_f32", 32, 4, d, dc);
    run_roles<1, 1>("v_mfma_f32_32x32x16_bf16", "v_fma_f32", 32, 4, d, dc);
    run_roles<1, 2>("v_mfma_f32_32x32x16_bf16", "ds_read_b128", 32, 16, d, dc);
#define SAME(MK, NV, NAME) hipFuncSetAttribute(reinterpret_cast<const void*>(&same_wave<MK, NV>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); run_same<MK, NV>(NAME, d, dc);
    SAME(0, 0, "v_mfma_f32_32x32x2_f32") SAME(0, 8, "v_mfma_f32_32x32x2_f32") SAME(0, 16, "v_mfma_f32_32x32x2_f32") SAME(0, 24, "v_mfma_f32_32x32x2_f32")
    SAME(0, 32, "v_mfma_f32_32x32x2_f32") SAME(0, 48, "v_mfma_f32_32x32x2_f32")
    SAME(1, 0, "v_mfma_f32_32x32x16_bf16") SAME(1, 4, "v_mfma_f32_32x32x16_bf16") SAME(1, 8, "v_mfma_f32_32x32x16_bf16") SAME(1, 12, "v_mfma_f32_32x32x16_bf16")
    SAME(1, 16, "v_mfma_f32_32x32x16_bf16") SAME(1, 24, "v_mfma_f32_32x32x16_bf16")
    return 0;
}
