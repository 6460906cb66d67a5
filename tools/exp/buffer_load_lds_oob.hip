// Does a range-checked buffer load TO LDS (buffer_load_dwordx4 ... lds) write zeros for the lanes whose offset is out of range, or leave
// their LDS slots alone?  (A lean temporally shifted conv1 would need zeros for the rows at clip ends without a per-lane select.)
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/exp/buffer_load_lds_oob.hip -o /tmp/bl_oob && /tmp/bl_oob
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k(const float* src, int nbytes, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 4];
    const int lane = threadIdx.x;
    for (int i = 0; i < 4; ++i) lds[lane * 4 + i] = -7.f;          // sentinel
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, nbytes, 0x00020000);
    // lanes 0..31 in range, lanes 32..63 far out of range
    const unsigned voff = lane < 32 ? lane * 16u : 0x7ffffff0u;
    typedef __attribute__((address_space(3))) void* lptr_t;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)lds, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = lds[lane * 4 + i];
}

int main() {
    float *src, *out;
    hipMalloc(&src, 64 * 16);
    hipMalloc(&out, 64 * 16);
    std::vector<float> h(256);
    for (int i = 0; i < 256; ++i) h[i] = 1.f + i;
    hipMemcpy(src, h.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, 1024, out);
    hipMemcpy(h.data(), out, 1024, hipMemcpyDeviceToHost);
    printf("lane 0: %g %g %g %g   lane 31: %g %g   lane 32 (out of range): %g %g %g %g   lane 63: %g %g\n", h[0], h[1], h[2], h[3], h[124], h[125], h[128],
           h[129], h[130], h[131], h[252], h[253]);
    printf("out-of-range lanes -> %s\n", h[128] == 0.f ? "ZEROS written to LDS" : h[128] == -7.f ? "LDS left untouched" : "something else");
    return 0;
}
