// What does ONE burst of filter loads cost when every workgroup of a launch asks the L2 for the same few hundred KB at the same moment
// (the B-fragment stream of mbconv_whole.hip: 8 waves per CU, one workgroup per CU, each wave NL x 1 KB)?
// Prints the mean s_memtime cycles from the first load's issue to s_waitcnt vmcnt(0), per configuration:
//   mode 0  every wave of the chip reads its own private region (no sharing; HBM / MALL once, then L2)
//   mode 1  wave w of every workgroup reads region (w + rot(block)) % NREG of a shared buffer (the kernel's pattern)
//   mode 2  like 1 without the rotation (all workgroups, wave w -> region w)
// build: hipcc -O3 --offload-arch=gfx950 tools/exp/l2_burst_bench.hip -o /tmp/l2_burst_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NL>
__global__ __launch_bounds__(512) void burst_kernel(const u32x4* __restrict__ w, int mode, int nreg, int reg_stride, int rounds,
                                                    unsigned long long* out, unsigned* sink, int store_first, unsigned long long* junk) {
    extern __shared__ char dsm[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long tot = 0;
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (int r = 0; r < rounds; ++r) {
        int reg;
        if (mode == 0) reg = (blockIdx.x * 8 + wave) * rounds + r;
        else if (mode == 1) reg = (wave + r * 8 + (int)((blockIdx.x * 5u) % (unsigned)nreg)) % nreg;
        else reg = (wave + r * 8) % nreg;
        const u32x4* p = w + (size_t)reg * reg_stride + lane;
        __syncthreads();
        const unsigned long long t0 = __builtin_readcyclecounter();
        // store_first: a global store in front of the burst (vmcnt counts it too: does the burst then wait for a write acknowledgement?)
        if (store_first == 1 && lane == 0) junk[(blockIdx.x * 8 + wave) * 16 + r] = t0;
        if (store_first == 2) junk[((size_t)(blockIdx.x * 8 + wave) * 16 + r) * 64 + lane] = t0;
        u32x4 v[NL];
#pragma unroll
        for (int u = 0; u < NL; ++u) v[u] = p[u * 64];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t1 = __builtin_readcyclecounter();
#pragma unroll
        for (int u = 0; u < NL; ++u) acc ^= v[u];
        tot += t1 - t0;
    }
    if (lane == 0) out[blockIdx.x * 8 + wave] = tot;
    if (acc.x == 0x12345678u && acc.y == 1u) sink[0] = acc.z + (unsigned)(size_t)dsm;
}

template <int NL>
void run(const u32x4* w, int mode, int nreg, int reg_stride, int rounds, int blocks, size_t lds, unsigned long long* out, unsigned* sink, int store_first = 0,
         unsigned long long* junk = nullptr) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&burst_kernel<NL>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    std::vector<unsigned long long> h(blocks * 8);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(burst_kernel<NL>, dim3(blocks), dim3(512), lds, 0, w, mode, nreg, reg_stride, rounds, out, sink, store_first, junk);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), out, h.size() * 8, hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h) s += (double)v;
    printf("store %d NL %2d mode %d nreg %3d blocks %4d lds %3zu KB: %8.0f cycles per burst (%5.1f B/clk per CU)\n", store_first, NL, mode, nreg, blocks, lds >> 10,
           s / h.size() / rounds, 8.0 * NL * 1024 / (s / h.size() / rounds));
}

int main() {
    const size_t maxb = (size_t)1 << 30;
    u32x4* w; unsigned* sink; unsigned long long* out;
    hipMalloc(&w, maxb); hipMemset(w, 1, maxb); hipMalloc(&sink, 64); hipMalloc(&out, 8 * 4096 * 8);
    const int rounds = 4;
    unsigned long long* junk; hipMalloc(&junk, (size_t)8 * 1024 * 16 * 64 * 8);
    for (int sf : {0, 1, 2}) {
        run<6>(w, 1, 13, 18 * 64, rounds, 1024, (size_t)150 << 10, out, sink, sf, junk);
        run<12>(w, 1, 13, 18 * 64, rounds, 1024, (size_t)150 << 10, out, sink, sf, junk);
    }
    return 0;
    for (size_t lds : {(size_t)150 << 10, (size_t)60 << 10})
        for (int blocks : {256, 1024})
            for (int mode : {0, 1, 2}) {
                const int stride = 18 * 64;       // 18 KB between regions (in 16-byte units), like a channel pair's fragments
                run<3>(w, mode, 13, stride, rounds, blocks, lds, out, sink);
                run<6>(w, mode, 13, stride, rounds, blocks, lds, out, sink);
                run<12>(w, mode, 13, stride, rounds, blocks, lds, out, sink);
            }
    return 0;
}
