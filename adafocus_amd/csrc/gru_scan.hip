// The T-sequential part of a GRU (ACT/models/gfv_net.py:427-435 classifier, ACT/models/ppo.py:67-96 policy) as ONE
// persistent kernel instead of 2 launches per step, with the classifier's per-step nn.Linear
// (gfv_net.py:433) riding in the same matrix products.  The input projections gi = W_ih x + b_ih of all steps are
// computed beforehand by one GEMM; what is sequential is  h_t = GRU(gi_t, W_hh h_{t-1}).
//
//   grid  = H / 8 blocks; block j owns hidden units [8j, 8j+8) = 24 rows of W_hh (gates r, z, n) -- 24 of the 32
//           columns of its MFMA B operand.  The 8 spare columns carry rows of the classifier weight: block j also owns
//           classes [cpb*j, cpb*j + cpb) (cpb = ceil(C / gridDim) <= 8), so the product  h_{t-1} x [W_hh rows | fc rows]^T
//           of step t yields the gates of step t AND the logits of step t-1; one extra pass after the last step emits
//           logits_{T-1} (= `last`).  No separate FC GEMM, no copy kernel.
//   W_hh  : the block's 24 (+cpb) rows stay in REGISTERS for the whole scan (wave w holds the k range [w*H/4, (w+1)*H/4)
//           as MFMA B fragments: 32 x f32x4 per lane), so the 12.6 MB of W_hh are read from HBM once, not T times.
//   step  : every wave multiplies h_{t-1}[64 x H/4] (A fragments straight from global/L2) with its slice on the fp32
//           matrix pipe, the four partial [64 x 32] products meet in LDS, 256 threads apply the gate math and write h_t;
//           batches above 64 clips are walked in chunks of 64 inside the step.
//   sync  : one grid-wide barrier per step (agent-scope release/acquire around a global counter; h_t crosses XCDs, whose
//           L2s are not coherent without it).  All H/8 = 128 blocks must be co-resident.  The launcher does not assume
//           that: it asks the runtime (hipOccupancyMaxActiveBlocksPerMultiprocessor) how many blocks fit per CU, bounds
//           the number of scans in flight by it, and can launch with hipLaunchCooperativeKernel (mode 2), which makes
//           the runtime itself refuse a grid that cannot be co-resident.  The spin is bounded anyway; a block that
//           times out poisons its outputs with NaN instead of hanging the device.
#include "adaf_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float sigm(float v) { return 1.f / (1.f + expf(-v)); }

struct GruScanArgs {
    const float* gi;     // [B, T, 3H] input projections (+ b_ih)
    const float* whh;    // [3H, H]
    const float* bhh;    // [3H]
    const float* h0;     // [B, H] initial state or nullptr (= 0)
    float* hs;           // [B, T, H]
    unsigned* bar;       // >= T + 1 zeroed counters
    int B, T;
    const float* fcw;    // [C, H] or nullptr
    const float* fcb;    // [C]
    float* logits;       // [B*T, C]
    float* last;         // [B, C] or nullptr
    int C, cpb;          // classes, classes per block
    unsigned* timeouts;  // device counter bumped by a block whose barrier wait ran out (or nullptr)
    int groups, bg;      // the batch in `groups` independent slices of `bg` clips, each with its own H / JB blocks and barrier words
    int bpad;            // 0: one counter per step (flat barrier); > 0: XCD-hierarchical barrier, 17 words per step at a pitch of `bpad` words
};

template <int H, int JB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void gru_scan_kernel(const GruScanArgs a) {
    constexpr int KQ = H / 4, NKK = KQ / 8, G3 = 3 * JB;
    __shared__ float red[4][2][32][33];
    __shared__ int timed_out;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5, nl = lane & 31;
    // Clips are independent: with more than one m-tile of clips and CUs to spare the batch is cut into `groups` slices, each scanned
    // by its own set of H / JB blocks with its own barrier words (B = 64: two slices of 32 clips on 256 CUs -- the products of a step
    // take half the time; every clip's arithmetic is unchanged).
    constexpr int NBLK = H / JB;
    const int grp = blockIdx.x / NBLK, jb = blockIdx.x - grp * NBLK;
    const int j0 = jb * JB;
    const int c0 = jb * a.cpb;                         // first class of this block
    const int T = a.T;
    const int bbeg = grp * a.bg;
    const int B = min(a.B, bbeg + a.bg);               // this slice's clips: [bbeg, B)
    unsigned* bar = a.bar + (size_t)grp * (T + 1) * (a.bpad > 0 ? 17 * a.bpad : 1);
    if (tid == 0) timed_out = 0;

    f32x4 wreg[NKK];
    {
        const float* wrow = nullptr;
        if (nl < G3) {
            const int g = nl / JB, jj = nl - g * JB;
            wrow = a.whh + (size_t)(g * H + j0 + jj) * H;
        } else if (a.fcw && nl - G3 < a.cpb && c0 + nl - G3 < a.C) {
            wrow = a.fcw + (size_t)(c0 + nl - G3) * H;
        }
        const bool valid = wrow != nullptr;
        const float* src = (valid ? wrow : a.whh) + wave * KQ + 4 * half;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + 8 * kk);
            wreg[kk] = valid ? v : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();

    const bool fc = a.fcw != nullptr;
    const int steps_total = T + (fc ? 1 : 0);
    const int nchunk = (B - bbeg + 63) >> 6;
    // What the gate math of a step needs besides the products -- the step's input projections, the previous state of the thread's own
    // (clip, unit) -- does not depend on the products: it is REQUESTED BEFORE them (three HBM / L2 round trips that used to start behind the
    // step's block barrier), and the hidden biases are per-thread constants of the whole scan (256 % JB == 0: a thread keeps its unit).
    constexpr int GIT = (64 * JB + 255) / 256;              // gate elements per thread and chunk of 64 clips
    static_assert(256 % JB == 0, "a thread keeps its hidden unit across its gate elements");
    const int gj = j0 + tid % JB;
    const float bh_r = a.bhh[gj], bh_z = a.bhh[H + gj], bh_n = a.bhh[2 * H + gj];
    for (int t = 0; t < steps_total; ++t) {
        const bool have_prev = t > 0 || a.h0 != nullptr;
        for (int ch = 0; ch < nchunk; ++ch) {
            const int b0 = bbeg + (ch << 6);
            const int rows = B - b0 < 64 ? B - b0 : 64;
            float g_r[GIT], g_z[GIT], g_n[GIT], g_hp[GIT];
            if (t < T) {
#pragma unroll
                for (int it = 0; it < GIT; ++it) {
                    const int idx = tid + 256 * it;
                    const int b = b0 + (idx < rows * JB ? idx / JB : 0);
                    const float* gir = a.gi + ((size_t)b * T + t) * 3 * H + gj;
                    g_r[it] = gir[0]; g_z[it] = gir[H]; g_n[it] = gir[2 * H];
                    g_hp[it] = !have_prev ? 0.f : t > 0 ? a.hs[((size_t)b * T + (t - 1)) * H + gj] : a.h0[(size_t)b * H + gj];
                }
            }
            if (have_prev) {
                const int mt = (rows + 31) >> 5;
                for (int m = 0; m < mt; ++m) {
                    f32x16 acc;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                    int brow = b0 + 32 * m + nl;
                    if (brow >= B) brow = B - 1;
                    const float* arow = (t > 0 ? a.hs + ((size_t)brow * T + (t - 1)) * H : a.h0 + (size_t)brow * H) + wave * KQ + 4 * half;
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
            if (t < T) {
#pragma unroll
                for (int it = 0; it < GIT; ++it) {
                    const int idx = tid + 256 * it;
                    if (idx >= rows * JB) break;
                    const int bl = idx / JB, jj = idx - bl * JB, j = j0 + jj, b = b0 + bl;
                    float hr = bh_r, hz = bh_z, hn = bh_n;
                    const float hp = g_hp[it];
                    if (have_prev) {
                        const int m = bl >> 5, row = bl & 31;
                        hr += (red[0][m][row][jj] + red[1][m][row][jj]) + (red[2][m][row][jj] + red[3][m][row][jj]);
                        hz += (red[0][m][row][JB + jj] + red[1][m][row][JB + jj]) + (red[2][m][row][JB + jj] + red[3][m][row][JB + jj]);
                        hn += (red[0][m][row][2 * JB + jj] + red[1][m][row][2 * JB + jj]) + (red[2][m][row][2 * JB + jj] + red[3][m][row][2 * JB + jj]);
                    }
                    const float r = sigm(g_r[it] + hr);
                    const float z = sigm(g_z[it] + hz);
                    const float nn = tanhf(g_n[it] + r * hn);
                    a.hs[((size_t)b * T + t) * H + j] = (1.f - z) * nn + z * hp;
                }
            }
            if (fc && t > 0) {   // logits of step t-1 came out of the same product (columns 3*JB ..)
                for (int idx = tid; idx < rows * a.cpb; idx += 256) {
                    const int bl = idx / a.cpb, ci = idx - bl * a.cpb, cls = c0 + ci, b = b0 + bl;
                    if (cls < a.C) {
                        const int m = bl >> 5, row = bl & 31, col = G3 + ci;
                        const float v = a.fcb[cls] + ((red[0][m][row][col] + red[1][m][row][col]) + (red[2][m][row][col] + red[3][m][row][col]));
                        a.logits[((size_t)b * T + (t - 1)) * a.C + cls] = v;
                        if (t == T && a.last) a.last[(size_t)b * a.C + cls] = v;
                    }
                }
            }
            if (ch + 1 < nchunk) __syncthreads();   // `red` is rewritten by the next chunk
        }
        if (t + 1 < steps_total) {
            __syncthreads();
            if (tid == 0) {
                // release (write back this XCD's L2: the block's h_t becomes visible device-wide) -> arrive -> ONE relaxed poll loop ->
                // acquire (invalidate this CU's L1).  Polling with acquire loads and two full __threadfence()s, as this barrier did
                // until round 4, is 13 us per step on this device; this form 7 (MI355X_MICROARCH.md, barrier-counter row).  The asm wait
                // keeps the arrive behind the write-back (the compiler may drop the fence's own wait when its scoreboard looks empty).
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                unsigned spins = 0;
                if (a.bpad > 0) {
                    // XCD-hierarchical form (round 5; MI355X_MICROARCH.md barrier-xcd: 4.1 us against 7.4 for one counter at 256 workgroups):
                    // 128 arrivals on one word serialise (~12 ns each) and 128 pollers hammer its line.  Blocks are dispatched round-robin
                    // over the 8 XCDs, so block jb arrives on the counter of group jb & 7 (16 arrivals); the group's LAST arriver is its
                    // leader for this step: release fence -> top counter (8 arrivals) -> poll it -> acquire fence -> the group's flag; the
                    // other 15 poll that flag.  Correct for any block -> XCD mapping (the groups are by block index); the mapping only
                    // decides whether a group's words stay within one XCD's reach.  17 words per step: [top | 8 counters | 8 flags].
                    unsigned* rec = bar + (size_t)t * 17 * a.bpad;
                    const int g8 = jb & 7;
                    const unsigned members = (unsigned)((NBLK + 7 - g8) >> 3);
                    const unsigned old = __hip_atomic_fetch_add(rec + (1 + g8) * a.bpad, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (old == members - 1u) {
                        // acquire + release: the other members' writes (ordered before their relaxed arrival by their own release fences) must
                        // happen-before this leader's arrival on the top counter, which is what the other groups' leaders synchronise with
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
                        __hip_atomic_fetch_add(rec, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        while (__hip_atomic_load(rec, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(NBLK < 8 ? NBLK : 8)) {
                            __builtin_amdgcn_s_sleep(1);
                            if (++spins > (1u << 24)) { timed_out = 1; break; }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                        __hip_atomic_store(rec + (9 + g8) * a.bpad, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        while (__hip_atomic_load(rec + (9 + g8) * a.bpad, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                            __builtin_amdgcn_s_sleep(1);
                            if (++spins > (1u << 24)) { timed_out = 1; break; }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    }
                } else {
                    __hip_atomic_fetch_add(bar + t, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    while (__hip_atomic_load(bar + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)NBLK) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > (1u << 24)) { timed_out = 1; break; }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
            }
            __syncthreads();
            if (timed_out) {   // never observed; refuses to hang the device if the grid cannot become co-resident
                if (tid == 0 && a.timeouts) atomicAdd(a.timeouts, 1u);     // ... and tells the host (adaf_gru_scan_timeouts)
                const float nan = __builtin_nanf("");
                for (int idx = tid; idx < (B - bbeg) * JB; idx += 256)
                    for (int tt = t + 1; tt < T; ++tt) a.hs[((size_t)(bbeg + idx / JB) * T + tt) * H + j0 + idx % JB] = nan;
                if (fc)
                    for (int idx = tid; idx < (B - bbeg) * a.cpb; idx += 256) {
                        const int b = bbeg + idx / a.cpb, cls = c0 + idx % a.cpb;
                        if (cls < a.C) {
                            for (int tt = t; tt < T; ++tt) a.logits[((size_t)b * T + tt) * a.C + cls] = nan;
                            if (a.last) a.last[(size_t)b * a.C + cls] = nan;
                        }
                    }
                return;
            }
        }
    }
}

// the barrier words are cleared by a kernel, not by hipMemsetAsync: captured into a HIP graph (GFV.capture_hot_path) the memset
// node did not reliably re-zero them on replay (measured: logits of later steps differed from the eager launch)
// ... and with agent-scope atomic stores: the scan's blocks count on these words with device-scope atomics from every XCD, and
// a plain store is only guaranteed to reach them through the end-of-kernel write-back, which a graph replay need not do between
// two of its nodes.
__global__ void zero_words_kernel(unsigned* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) __hip_atomic_store(p + i, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

constexpr int kH = 1024, kJB = 8, kGrid = kH / kJB;

}  // namespace

// How many scan blocks the runtime says fit on one CU (0 if the query fails): asked, not assumed.
int adaf_gru_scan_blocks_per_cu() {
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, gru_scan_kernel<kH, kJB>, 256, 0) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return nb;
}

bool adaf_gru_scan_persistent_ok(int batch, int hidden, int classes, int resident_blocks) {
    return hidden == kH && batch >= 1 && batch <= 256 && classes <= 8 * kGrid && resident_blocks >= kGrid;
}

// slices a scan of `batch` clips is cut into (each takes H / 8 = 128 co-resident blocks): 2 when there is more than one m-tile of clips
// and the device can hold both sets of blocks
int adaf_gru_scan_groups(int batch, int resident_blocks) {
    return (adaf_options().gru_scan_slices >= 2 && batch > 32 && resident_blocks >= 2 * kGrid) ? 2 : 1;
}

// The one place that decides how a scan uses its barrier buffer of `bar_words` words: how many slices (each with its own grid of kGrid
// blocks and its own barrier records), which record layout, and how many words to clear.  gru_scan() in api.hip reserves its scan slots from
// the same plan, so the slot accounting and the launch cannot disagree.  groups == 0: the buffer cannot hold even the single counters.
AdafGruScanPlan adaf_gru_scan_plan(int batch, int steps, size_t bar_words, int resident_blocks) {
    AdafGruScanPlan p{0, 0, 0};
    if ((size_t)(steps + 1) > bar_words) return p;
    p.groups = adaf_gru_scan_groups(batch, resident_blocks);
    if (p.groups > 1 && (size_t)p.groups * (steps + 1) > bar_words) p.groups = 1;
    // XCD-hierarchical barrier records (17 words per step) at a pitch of 16 words (64 bytes) when the buffer has the room, packed
    // otherwise, the single counter per step when even that does not fit
    const size_t recs = (size_t)p.groups * (steps + 1) * 17;
    p.bpad = recs * 16 <= bar_words ? 16 : recs <= bar_words ? 1 : 0;
    p.nzero = p.bpad ? recs * p.bpad : (size_t)p.groups * (steps + 1);
    return p;
}

hipError_t adaf_launch_gru_scan_persistent(const float* gi, const float* whh, const float* bhh, const float* h0, float* hs,
                                           unsigned* bar, const AdafGruScanPlan& plan, int batch, int steps, const float* fcw, const float* fcb,
                                           float* logits, float* last, int classes, bool cooperative, unsigned* timeouts, hipStream_t s) {
    if (plan.groups < 1) return hipErrorInvalidValue;
    const int groups = plan.groups, bpad = plan.bpad;
    const size_t nzero = plan.nzero;
    GruScanArgs a;
    a.bpad = bpad;
    a.timeouts = timeouts;
    a.gi = gi; a.whh = whh; a.bhh = bhh; a.h0 = h0; a.hs = hs; a.bar = bar; a.B = batch; a.T = steps;
    a.fcw = fcw; a.fcb = fcb; a.logits = logits; a.last = last; a.C = fcw ? classes : 0;
    a.cpb = fcw ? (classes + kGrid - 1) / kGrid : 0;
    a.groups = groups;
    a.bg = a.groups == 1 ? batch : ((batch + a.groups - 1) / a.groups + 31) / 32 * 32;      // whole m-tiles per slice
    const int zthreads = nzero > 256 ? 256 : 64;
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)((nzero + zthreads - 1) / zthreads)), dim3(zthreads), 0, s, bar, (int)nzero);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    if (cooperative) {
        void* params[] = {&a};
        return hipLaunchCooperativeKernel(reinterpret_cast<const void*>(gru_scan_kernel<kH, kJB>), dim3(kGrid * a.groups), dim3(256), params, 0, s);
    }
    hipLaunchKernelGGL((gru_scan_kernel<kH, kJB>), dim3(kGrid * a.groups), dim3(256), 0, s, a);
    return hipGetLastError();
}
