// How fast can every workgroup of a launch stream the SAME few-MB buffer out of L2 (the weight stream of mbconv_whole.hip)?
// usage: l2_stream_bench  -> table over (threads per block, LDS per block (=> blocks per CU), loads in flight per thread, region size)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int U>
__global__ __launch_bounds__(1024) void stream_kernel(const u32x4* __restrict__ w, int chunks, int rot, unsigned* sink) {
    extern __shared__ char dsm[];
    const int nthr = blockDim.x;
    u32x4 acc = {0u, 0u, 0u, 0u};
    const int per_iter = nthr * U;
    int start = rot ? (int)((blockIdx.x * 977u) % (unsigned)(chunks / per_iter)) * per_iter : 0;
    for (int it = 0; it < chunks / per_iter; ++it) {
        int base = start + it * per_iter; if (base >= chunks) base -= chunks;
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = w[base + u * nthr + threadIdx.x];
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    if (acc.x == 0x12345678u && acc.y == 1u) sink[0] = acc.z + (unsigned)(size_t)dsm;
}
template <int U>
float run(const u32x4* w, int chunks, int rot, unsigned* sink, int blocks, int threads, size_t lds) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kernel<U>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(stream_kernel<U>, dim3(blocks), dim3(threads), lds, 0, w, chunks, rot, sink);
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(stream_kernel<U>, dim3(blocks), dim3(threads), lds, 0, w, chunks, rot, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}
int main() {
    const size_t maxb = 8u << 20;
    u32x4* w; unsigned* sink;
    hipMalloc(&w, maxb); hipMemset(w, 1, maxb); hipMalloc(&sink, 64);
    const int blocks = 1024;
    printf("region_KB threads lds_KB U rot  ms  TB/s_aggregate GB/s_per_CU\n");
    for (size_t region : {512u << 10, 2048u << 10, 6144u << 10})
        for (int threads : {512, 1024})
            for (size_t lds : {(size_t)150 << 10, (size_t)70 << 10, (size_t)30 << 10})
                for (int rot : {0, 1})
                    for (int U : {4, 16}) {
                        const int chunks = (int)(region / 16);
                        float ms = U == 4 ? run<4>(w, chunks, rot, sink, blocks, threads, lds) : run<16>(w, chunks, rot, sink, blocks, threads, lds);
                        double bytes = (double)region * blocks;
                        printf("%6zu %5d %4zu %3d %d  %.3f  %.2f  %.1f\n", region >> 10, threads, lds >> 10, U, rot, ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
                    }
    return 0;
}
