// The T-sequential part of a GRU (ACT/models/gfv_net.py:427-435 classifier, ACT/models/ppo.py:67-96 policy) as ONE
// persistent kernel instead of 2 launches per step.  The input projections gi = W_ih x + b_ih of all
// steps are computed beforehand by one GEMM; what is sequential is  h_t = GRU(gi_t, W_hh h_{t-1}).
//
//   grid  = H / 8 blocks; block j owns hidden units [8j, 8j+8) = 24 rows of W_hh (gates r, z, n).
//   W_hh  : those 24 rows stay in REGISTERS for the whole scan (wave w holds the k range [w*H/4, (w+1)*H/4) as MFMA
//           B fragments: 32 x f32x4 per lane), so the 12.6 MB of W_hh are read from HBM once, not T times.
//   step  : every wave multiplies h_{t-1}[B x H/4] (A fragments straight from global/L2) with its slice on the fp32
//           matrix pipe, the four partial [B x 24] products meet in LDS, 256 threads apply the gate math and write h_t.
//   sync  : one grid-wide barrier per step (agent-scope release/acquire around a global counter; h_t crosses XCDs, whose
//           L2s are not coherent without it).  All H/8 = 128 blocks must be co-resident: a block needs 34 KB of LDS and
//           ~200 VGPRs, so two fit per CU and four scans can be in flight on different streams without blocking each
//           other out; the spin is bounded, a block that times out poisons its outputs with NaN instead of hanging.
#include "adaf_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float sigm(float v) { return 1.f / (1.f + expf(-v)); }

template <int H, int JB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void gru_scan_kernel(const float* __restrict__ gi, const float* __restrict__ whh,
                                                       const float* __restrict__ bhh, float* hs, unsigned* bar, int B, int T) {
    constexpr int KQ = H / 4, NKK = KQ / 8;
    __shared__ float red[4][2][32][33];
    __shared__ int timed_out;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, nl = lane & 31;
    const int j0 = blockIdx.x * JB;
    if (tid == 0) timed_out = 0;

    f32x4 wreg[NKK];
    {
        const int g = nl / JB, jj = nl - g * JB;
        const bool valid = nl < 3 * JB;
        const float* wrow = whh + ((size_t)(valid ? g * H + j0 + jj : 0)) * H + wave * KQ + 4 * half;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(wrow + 8 * kk);
            wreg[kk] = valid ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const int mt = (B + 31) >> 5;
    __syncthreads();

    for (int t = 0; t < T; ++t) {
        if (t > 0) {
            for (int m = 0; m < mt; ++m) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                const int brow = 32 * m + nl;
                const float* arow = hs + ((size_t)(brow < B ? brow : B - 1) * T + (t - 1)) * H + wave * KQ + 4 * half;
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) {
                    const f32x4 af = *reinterpret_cast<const f32x4*>(arow + 8 * kk);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, wreg[kk].x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, wreg[kk].y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, wreg[kk].z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, wreg[kk].w, acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) red[wave][m][(r & 3) + 8 * (r >> 2) + 4 * half][nl] = acc[r];
            }
            __syncthreads();
        }
        for (int idx = tid; idx < B * JB; idx += 256) {
            const int b = idx / JB, jj = idx - b * JB, j = j0 + jj;
            float hr = bhh[j], hz = bhh[H + j], hn = bhh[2 * H + j], hp = 0.f;
            if (t > 0) {
                const int m = b >> 5, row = b & 31;
                hr += (red[0][m][row][jj] + red[1][m][row][jj]) + (red[2][m][row][jj] + red[3][m][row][jj]);
                hz += (red[0][m][row][JB + jj] + red[1][m][row][JB + jj]) + (red[2][m][row][JB + jj] + red[3][m][row][JB + jj]);
                hn += (red[0][m][row][2 * JB + jj] + red[1][m][row][2 * JB + jj]) + (red[2][m][row][2 * JB + jj] + red[3][m][row][2 * JB + jj]);
                hp = hs[((size_t)b * T + (t - 1)) * H + j];
            }
            const float* gir = gi + ((size_t)b * T + t) * 3 * H;
            const float r = sigm(gir[j] + hr);
            const float z = sigm(gir[H + j] + hz);
            const float nn = tanhf(gir[2 * H + j] + r * hn);
            hs[((size_t)b * T + t) * H + j] = (1.f - z) * nn + z * hp;
        }
        if (t + 1 < T) {
            __syncthreads();
            if (tid == 0) {
                __threadfence();                      // release: this block's h_t is visible device-wide
                atomicAdd(bar + t, 1u);
                unsigned spins = 0;
                while (__hip_atomic_load(bar + t, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) {
                    __builtin_amdgcn_s_sleep(4);
                    if (++spins > (1u << 23)) { timed_out = 1; break; }
                }
                __threadfence();                      // acquire: the other blocks' h_t
            }
            __syncthreads();
            if (timed_out) {   // never observed; refuses to hang the device if the grid cannot become co-resident
                for (int idx = tid; idx < B * JB; idx += 256)
                    for (int tt = t + 1; tt < T; ++tt)
                        hs[((size_t)(idx / JB) * T + tt) * H + j0 + idx % JB] = __builtin_nanf("");
                return;
            }
        }
    }
}

}  // namespace

bool adaf_gru_scan_persistent_ok(int batch, int hidden, int cus) { return hidden == 1024 && batch >= 1 && batch <= 64 && cus * 2 >= hidden / 8; }

void adaf_launch_gru_scan_persistent(const float* gi, const float* whh, const float* bhh, float* hs, unsigned* bar, int batch,
                                     int steps, hipStream_t s) {
    (void)hipMemsetAsync(bar, 0, sizeof(unsigned) * steps, s);
    hipLaunchKernelGGL((gru_scan_kernel<1024, 8>), dim3(1024 / 8), dim3(256), 0, s, gi, whh, bhh, hs, bar, batch, steps);
}
