// VALU issue rates on gfx950: cycles per wave-instruction of v_fma_f32, v_pk_fma_f32, v_pk_mul_f32, v_exp_f32, v_rcp_f32, v_cvt at
// 1 / 2 / 4 waves per SIMD (workgroups of 256 / 512 / 1024 threads, one per CU through a 150 KB LDS request).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int OP>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float seed) {
    extern __shared__ char dsm[];
    float a[16]; f32x2 b[8];
    for (int i = 0; i < 16; ++i) a[i] = seed + i + threadIdx.x * 1e-3f;
    for (int i = 0; i < 8; ++i) b[i] = f32x2{a[2 * i], a[2 * i + 1]};
    const float w = seed * 0.999f; const f32x2 w2 = {w, w * 1.0001f};
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (OP == 0) { _Pragma("unroll") for (int i = 0; i < 16; ++i) a[i] = __builtin_fmaf(a[i], w, 0.5f); }
            if (OP == 1) { _Pragma("unroll") for (int i = 0; i < 8; ++i) b[i] = __builtin_elementwise_fma(b[i], w2, f32x2{0.5f, 0.25f}); }
            if (OP == 2) { _Pragma("unroll") for (int i = 0; i < 8; ++i) b[i] = b[i] * w2; }
            if (OP == 3) { _Pragma("unroll") for (int i = 0; i < 16; ++i) a[i] = __builtin_amdgcn_exp2f(a[i]); }
            if (OP == 4) { _Pragma("unroll") for (int i = 0; i < 16; ++i) a[i] = __builtin_amdgcn_rcpf(a[i]); }
            if (OP == 5) { _Pragma("unroll") for (int i = 0; i < 16; ++i) a[i] = (float)(_Float16)a[i] + 0.f * w; }
            if (OP == 6) { _Pragma("unroll") for (int i = 0; i < 16; ++i) a[i] = a[i] * w; }
        }
    }
    long long t1 = clock64();
    float s = 0; for (int i = 0; i < 16; ++i) s += a[i]; for (int i = 0; i < 8; ++i) s += b[i].x + b[i].y;
    if (s == 12345.f) out[0] = s + (float)(size_t)dsm;
    if (threadIdx.x == 0 && blockIdx.x == 0) out[1] = (float)(t1 - t0);
}
template <int OP> void run(const char* name, float* d, int per_iter) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<OP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int threads : {256, 512, 1024}) {
        const int iters = 2000;
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 150 * 1024, 0, d, iters, 1.0001f);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<OP>, dim3(256), dim3(threads), 150 * 1024, 0, d, iters, 1.0001f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        const double insts = (double)iters * per_iter;      // wave-instructions per wave
        const int wps = threads / 256;
        printf("%-14s waves/SIMD %d: %.2f clk per wave-instr (clock64), %.2f clk per instr per SIMD (wall @2.4GHz: %.2f)\n", name, wps, h[1] / insts,
               h[1] / insts / wps, ms * 1e-3 * 2.4e9 / insts / wps);
    }
}
int main() {
    float* d; hipMalloc(&d, 64);
    run<0>("v_fma_f32", d, 64); run<1>("v_pk_fma_f32", d, 32); run<2>("v_pk_mul_f32", d, 32); run<3>("v_exp_f32", d, 64);
    run<4>("v_rcp_f32", d, 64); run<5>("cvt f16 rt", d, 128); run<6>("v_mul_f32", d, 64);
    return 0;
}
