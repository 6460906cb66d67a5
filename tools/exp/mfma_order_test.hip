// Is a v_mfma_f32_16x16x4_f32 chain bit-identical to the engine's v_mfma_f32_32x32x2_f32 chain when it visits k in the same order?
// Engine order inside an 8-k group (a lane's fragment = floats 4*half .. 4*half+3; products x, y, z, w):
//   MFMA x: k = {g+0 (lanes 0-31), g+4 (lanes 32-63)}, y: {g+1, g+5}, z: {g+2, g+6}, w: {g+3, g+7}
// i.e. 0,4,1,5,2,6,3,7 if an instruction adds its k-lanes in ascending lane-group order.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k32(const float* A, const float* B, int K, float* C) {   // A [32][K], B [32][K] (row n, k), C [32][32]
    const int lane = threadIdx.x, nl = lane & 31, half = lane >> 5;
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    for (int g = 0; g < K; g += 8) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(A + nl * K + g + 4 * half);
        const f32x4 b = *reinterpret_cast<const f32x4*>(B + nl * K + g + 4 * half);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * half) * 32 + nl] = acc[i];
}

// order[4]: the k offsets (within the 8-k group) the four lane groups of the FIRST 16x16x4 take; the second instruction takes +2
__global__ void k16(const float* A, const float* B, int K, float* C, int o0, int o1, int o2, int o3, int p0, int p1, int p2, int p3) {
    const int lane = threadIdx.x, r = lane & 15, kg = lane >> 4;
    const int ofs1 = kg == 0 ? o0 : kg == 1 ? o1 : kg == 2 ? o2 : o3;
    const int ofs2 = kg == 0 ? p0 : kg == 1 ? p1 : kg == 2 ? p2 : p3;
    for (int ti = 0; ti < 2; ++ti)
        for (int tj = 0; tj < 2; ++tj) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            for (int g = 0; g < K; g += 8) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(ti * 16 + r) * K + g + ofs1], B[(tj * 16 + r) * K + g + ofs1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(ti * 16 + r) * K + g + ofs2], B[(tj * 16 + r) * K + g + ofs2], acc, 0, 0, 0);
            }
            for (int i = 0; i < 4; ++i) C[(ti * 16 + 4 * kg + i) * 32 + tj * 16 + r] = acc[i];
        }
}

int main() {
    const int K = 512;
    std::vector<float> A(32 * K), B(32 * K);
    srand(7);
    for (auto& v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto& v : B) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 37.f;
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096); hipMalloc(&dD, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dA, dB, K, dC);
    std::vector<float> C(1024), D(1024), R(1024);
    hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
    // host fmaf chains in candidate orders
    const int cand[4][8] = {{0, 4, 1, 5, 2, 6, 3, 7}, {4, 0, 5, 1, 6, 2, 7, 3}, {0, 1, 2, 3, 4, 5, 6, 7}, {0, 4, 1, 5, 2, 6, 3, 7}};
    for (int c = 0; c < 3; ++c) {
        int diff = 0;
        for (int i = 0; i < 32; ++i)
            for (int j = 0; j < 32; ++j) {
                float s = 0.f;
                for (int g = 0; g < K; g += 8)
                    for (int q = 0; q < 8; ++q) s = fmaf(A[i * K + g + cand[c][q]], B[j * K + g + cand[c][q]], s);
                if (memcmp(&s, &C[i * 32 + j], 4)) ++diff;
            }
        printf("32x32x2 chain vs host fmaf order %d: %d of 1024 differ\n", c, diff);
    }
    const int orders[4][8] = {{0, 4, 1, 5, 2, 6, 3, 7}, {0, 1, 2, 3, 4, 5, 6, 7}, {0, 1, 4, 5, 2, 3, 6, 7}, {4, 0, 5, 1, 6, 2, 7, 3}};
    for (int o = 0; o < 4; ++o) {
        hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dB, K, dD, orders[o][0], orders[o][1], orders[o][2], orders[o][3], orders[o][4], orders[o][5],
                           orders[o][6], orders[o][7]);
        hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
        int diff = 0;
        double md = 0;
        for (int i = 0; i < 1024; ++i) { if (memcmp(&C[i], &D[i], 4)) ++diff; md = fmax(md, fabs((double)C[i] - D[i])); }
        printf("16x16x4 lane-group order {%d %d %d %d | %d %d %d %d} vs 32x32x2: %d of 1024 differ (max abs %.3g)\n", orders[o][0], orders[o][1], orders[o][2],
               orders[o][3], orders[o][4], orders[o][5], orders[o][6], orders[o][7], diff, md);
    }
    return 0;
}
