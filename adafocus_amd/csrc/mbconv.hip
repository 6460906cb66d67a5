// Fused expand (1x1 conv + BN + ReLU6) -> depthwise 3x3 (+ BN + ReLU6) of a MobileNetV2 inverted-residual
// block (ACT/models/mobilenet.py:42-68; STH/models/mobilenetv2.py): the 6x-expanded feature map never goes
// to HBM.  For the glancer's high-resolution blocks (b2..b7, 112^2..28^2 maps) that tensor is the largest
// transfer of the whole network (4.8 MB per 224^2 frame for b2 alone, written once and read once).
//
// One block = one spatial tile of one frame:
//   stride 1: 8x8 outputs <- 10x10 input halo;  stride 2: 3x8 outputs <- 7x17 input halo  (<= 128 halo pixels)
//   1. the halo pixels' Cin input channels go to LDS once;
//   2. per chunk of 32 hidden channels:  E[halo][32] = X[halo][Cin] * We[32][Cin]^T on the fp32 matrix pipe
//      (one 32x32 tile per wave, same k order as the conv engine -> bit-identical to the unfused expand),
//      BN + ReLU6, halo pixels outside the image forced to 0 (the depthwise conv pads the EXPANDED map),
//      E to LDS;  then the 3x3 depthwise taps on the VALU straight from LDS (same tap order as
//      dwconv3x3_kernel), BN + ReLU6, 16-byte stores.
// The halo is recomputed by neighbouring tiles (1.56x / 1.24x of the expand FLOPs, which are ~1/10 of the
// block's traffic-equivalent cost at these channel counts).
#include "adaf_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

template <int S>
__global__ __launch_bounds__(256) void mb_expand_dw_kernel(const MbFuseArgs a) {
    constexpr int TH = S == 1 ? 8 : 3, TW = 8;
    constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3;
    constexpr int HP = IH * IW;     // 100 / 119 halo pixels
    constexpr int MP = 128;         // padded to four 32-row bands, one per wave
    constexpr int EP = 36;          // E row pitch in floats
    static_assert(HP <= MP, "halo must fit four bands");
    extern __shared__ __attribute__((aligned(16))) float smem[];   // X[MP][cin+4] | We chunk [32][cin+4] | E[MP][EP]
    float* Xs = smem;
    float* Ws = smem + MP * (a.cin + 4);
    float* Es = Ws + 32 * (a.cin + 4);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bid = blockIdx.x;
    const int tx = bid % a.tiles_x;
    bid /= a.tiles_x;
    const int ty = bid % a.tiles_y;
    const int img = bid / a.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW;
    const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
    const int cq = a.cin >> 2;      // 16-byte chunks per pixel
    const int xp = a.cin + 4;       // LDS pitch of X and We rows: conflict-free b128 fragment reads for cin = 16, 24, 32
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    for (int idx = tid; idx < MP * cq; idx += 256) {
        const int p = idx / cq, c = idx - p * cq;
        const int iy = iy0 + p / IW, ix = ix0 + p % IW;
        const bool ok = p < HP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        const f32x4 v = *reinterpret_cast<const f32x4*>(ok ? a.x + (((size_t)img * a.H + iy) * a.W + ix) * a.cin + 4 * c : a.zeros);
        *reinterpret_cast<f32x4*>(&Xs[p * xp + 4 * c]) = v;
    }
    // which of this lane's 16 accumulator rows are halo pixels inside the image
    unsigned emask = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int p = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int iy = iy0 + p / IW, ix = ix0 + p % IW;
        if (p < HP && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) emask |= 1u << r;
    }
    const int c4 = tid & 7, pg = tid >> 3;
    const int fr = (32 * wave + (lane & 31)) * xp + 4 * (lane >> 5);
    const int fw = (lane & 31) * xp + 4 * (lane >> 5);

    // expand weights of a chunk: 32 rows x cq chunks <= 256 -> at most one 16-byte piece per thread
    const int wr = tid / cq, wc = tid - wr * cq;
    const bool wmine = tid < 32 * cq;
    auto wload = [&](int ch0) -> f32x4 {
        return (wmine && ch0 + wr < a.hid) ? *reinterpret_cast<const f32x4*>(a.we + (size_t)(ch0 + wr) * a.cin + 4 * wc) : zero4;
    };
    if (wmine) *reinterpret_cast<f32x4*>(&Ws[wr * xp + 4 * wc]) = wload(0);

    for (int ch0 = 0; ch0 < a.hid; ch0 += 32) {
        // everything this chunk and the next need from global memory is requested before the first barrier, so its
        // latency runs under the expand GEMM instead of in front of the depthwise phase / the next chunk
        const f32x4 wnext = wload(ch0 + 32);
        const int ch = ch0 + 4 * c4;
        const int chs = ch < a.hid ? ch : 0;
        f32x4 k[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) k[t] = *reinterpret_cast<const f32x4*>(a.wd + (size_t)t * a.hid + chs);
        const f32x4 dsc = *reinterpret_cast<const f32x4*>(a.sd + chs);
        const f32x4 dbi = *reinterpret_cast<const f32x4*>(a.bd + chs);
        const int nch = ch0 + (lane & 31);
        const bool nv = nch < a.hid;
        const float esc = nv ? a.se[nch] : 0.f, ebi = nv ? a.be[nch] : 0.f;
        __syncthreads();
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int kk = 0; kk < a.cin / 8; ++kk) {
            const f32x4 af = *reinterpret_cast<const f32x4*>(&Xs[fr + 8 * kk]);
            const f32x4 bf = *reinterpret_cast<const f32x4*>(&Ws[fw + 8 * kk]);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc, 0, 0, 0);
        }
        {
            const float sc = esc, bi = ebi;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                const float e = fminf(fmaxf(fmaf(acc[r], sc, bi) + 0.f, 0.f), 6.f);
                Es[p * EP + (lane & 31)] = ((emask >> r) & 1u) ? e : 0.f;
            }
        }
        __syncthreads();
        if (wmine) *reinterpret_cast<f32x4*>(&Ws[wr * xp + 4 * wc]) = wnext;   // every wave is past its reads of this chunk's weights
        if (ch < a.hid) {
            const f32x4 sc = dsc, bi = dbi;
#pragma unroll
            for (int o = pg; o < TH * TW; o += 32) {
                const int oy = o / TW, ox = o - oy * TW;
                const int gy = oy0 + oy, gx = ox0 + ox;
                f32x4 s = zero4;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const f32x4 v = *reinterpret_cast<const f32x4*>(&Es[((oy * S + ky) * IW + ox * S + kx) * EP + 4 * c4]);
                        const f32x4 w = k[ky * 3 + kx];
                        s.x = fmaf(v.x, w.x, s.x);
                        s.y = fmaf(v.y, w.y, s.y);
                        s.z = fmaf(v.z, w.z, s.z);
                        s.w = fmaf(v.w, w.w, s.w);
                    }
                if (gy < a.OH && gx < a.OW) {
                    f32x4 r;
                    r.x = fminf(fmaxf(fmaf(s.x, sc.x, bi.x), 0.f), 6.f);
                    r.y = fminf(fmaxf(fmaf(s.y, sc.y, bi.y), 0.f), 6.f);
                    r.z = fminf(fmaxf(fmaf(s.z, sc.z, bi.z), 0.f), 6.f);
                    r.w = fminf(fmaxf(fmaf(s.w, sc.w, bi.w), 0.f), 6.f);
                    *reinterpret_cast<f32x4*>(a.out + (((size_t)img * a.OH + gy) * a.OW + gx) * a.hid + ch) = r;
                }
            }
        }
        // no barrier here: the next chunk's barrier (after its weight load) orders these reads of E before the next writes
    }
}

// ---------------------------------------------------------------------------------------------------------
// Stem (3x3 stride-2 conv 3->32 + BN + ReLU6) -> block 1 (depthwise 3x3 + BN + ReLU6 -> project 1x1 32->16 + BN)
// in one persistent kernel: reads the 224^2 pixel-major frame, writes the 112^2 x 16 map; the two 112^2 x 32
// intermediates (1.6 MB per frame each, the 2nd and 3rd largest tensors of the network) stay in LDS.
//   tile = 8x8 outputs <- 10x10 stem outputs (halo) <- 21x21 input pixels
//   stem GEMM: the k index of the packed filter is (tap, channel-of-4), so a lane's 16-byte A fragment IS one input
//   pixel of the tile in LDS -- no im2col; K = 36 -> five groups of 8 (the 10th "tap" reads a zero vector), the
//   same group order as the conv engine's two k slices, so the result is bit-identical to the unfused stem.
// (Requesting the next tile's pixels during the current tile's compute was tried and was slower: 14.6 vs 13.9 ms
//  per 1024 frames for the whole glancer; three co-resident blocks already cover the load.)
template <int DUMMY>
__global__ __launch_bounds__(256) void mb_stem_b1_kernel(const MbStemArgs a) {
    constexpr int TW = 8, IW = 10, HP = 100, MP = 128, EP = 36;
    constexpr int XW = 21, XPIX = 21 * 21;
    __shared__ __attribute__((aligned(16))) float Xs[(XPIX + 1) * 4];   // + one zero pixel
    __shared__ __attribute__((aligned(16))) float Wst[32 * 44];
    __shared__ __attribute__((aligned(16))) float Es[MP * EP];
    __shared__ __attribute__((aligned(16))) float Ds[64 * EP];
    __shared__ __attribute__((aligned(16))) float Wps[32 * EP];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, half = lane >> 5;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    for (int idx = tid; idx < 32 * 10; idx += 256) {
        const int nr = idx / 10, c = idx - nr * 10;
        *reinterpret_cast<f32x4*>(&Wst[nr * 44 + 4 * c]) = c < 9 ? *reinterpret_cast<const f32x4*>(a.ws + nr * 36 + 4 * c) : zero4;
    }
    for (int idx = tid; idx < 32 * 8; idx += 256) {
        const int nr = idx >> 3, c = idx & 7;
        *reinterpret_cast<f32x4*>(&Wps[nr * EP + 4 * c]) = nr < 16 ? *reinterpret_cast<const f32x4*>(a.wp + nr * 32 + 4 * c) : zero4;
    }
    if (tid < 4) Xs[XPIX * 4 + tid] = 0.f;
    const int c4 = tid & 7, pg = tid >> 3;
    f32x4 k[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) k[t] = *reinterpret_cast<const f32x4*>(a.wd + t * 32 + 4 * c4);
    const f32x4 dsc = *reinterpret_cast<const f32x4*>(a.sd + 4 * c4);
    const f32x4 dbi = *reinterpret_cast<const f32x4*>(a.bd + 4 * c4);
    const int nl = lane & 31;
    const float ssc = a.ss[nl], sbi = a.bs[nl];
    const float psc = nl < 16 ? a.sp[nl] : 0.f, pbi = nl < 16 ? a.bp[nl] : 0.f;
    // A-fragment offsets of this lane's halo pixel: tap t = 2*kk + half -> input pixel (2*hy + kh, 2*hx + kw) of the tile
    const int p = 32 * wave + nl;
    const int pl = p < HP ? p : 0;
    const int base = ((2 * (pl / IW)) * XW + 2 * (pl % IW)) * 4;
    int aoff[5];
#pragma unroll
    for (int kk = 0; kk < 5; ++kk) {
        const int t = 2 * kk + half;
        aoff[kk] = t < 9 ? base + ((t / 3) * XW + t % 3) * 4 : XPIX * 4;
    }
    const int woff = nl * 44 + 4 * half;

    for (int tile = blockIdx.x; tile < a.total_tiles; tile += gridDim.x) {
        int q = tile;
        const int tx = q % a.tiles_x;
        q /= a.tiles_x;
        const int ty = q % a.tiles_y;
        const int img = q / a.tiles_y;
        const int oy0 = ty * 8, ox0 = tx * 8;
        const int hy0 = oy0 - 1, hx0 = ox0 - 1;
        const int iy0 = 2 * hy0 - 1, ix0 = 2 * hx0 - 1;
        __syncthreads();   // the previous tile's readers are done with Xs / Es / Ds
        for (int idx = tid; idx < XPIX; idx += 256) {
            const int r = idx / XW, c = idx - r * XW;
            const int iy = iy0 + r, ix = ix0 + c;
            const bool ok = (unsigned)iy < (unsigned)a.S && (unsigned)ix < (unsigned)a.S;
            *reinterpret_cast<f32x4*>(&Xs[idx * 4]) =
                *reinterpret_cast<const f32x4*>(ok ? a.x + (((size_t)img * a.S + iy) * a.S + ix) * 4 : a.zeros);
        }
        unsigned emask = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pp = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * half;
            const int hy = hy0 + pp / IW, hx = hx0 + pp % IW;
            if (pp < HP && (unsigned)hy < (unsigned)a.H1 && (unsigned)hx < (unsigned)a.H1) emask |= 1u << r;
        }
        __syncthreads();
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int kk = 0; kk < 5; ++kk) {
            const f32x4 af = *reinterpret_cast<const f32x4*>(&Xs[aoff[kk]]);
            const f32x4 bf = *reinterpret_cast<const f32x4*>(&Wst[woff + 8 * kk]);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int pp = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * half;
            const float e = fminf(fmaxf(fmaf(acc[r], ssc, sbi) + 0.f, 0.f), 6.f);
            Es[pp * EP + nl] = ((emask >> r) & 1u) ? e : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int o = pg; o < 64; o += 32) {
            const int oy = o / TW, ox = o - oy * TW;
            f32x4 s = zero4;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(&Es[((oy + ky) * IW + ox + kx) * EP + 4 * c4]);
                    const f32x4 w = k[ky * 3 + kx];
                    s.x = fmaf(v.x, w.x, s.x);
                    s.y = fmaf(v.y, w.y, s.y);
                    s.z = fmaf(v.z, w.z, s.z);
                    s.w = fmaf(v.w, w.w, s.w);
                }
            f32x4 r;
            r.x = fminf(fmaxf(fmaf(s.x, dsc.x, dbi.x), 0.f), 6.f);
            r.y = fminf(fmaxf(fmaf(s.y, dsc.y, dbi.y), 0.f), 6.f);
            r.z = fminf(fmaxf(fmaf(s.z, dsc.z, dbi.z), 0.f), 6.f);
            r.w = fminf(fmaxf(fmaf(s.w, dsc.w, dbi.w), 0.f), 6.f);
            *reinterpret_cast<f32x4*>(&Ds[o * EP + 4 * c4]) = r;
        }
        __syncthreads();
        if (wave < 2) {   // project: 64 pixels x 16 channels, K = 32
            f32x16 pa;
#pragma unroll
            for (int r = 0; r < 16; ++r) pa[r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const f32x4 af = *reinterpret_cast<const f32x4*>(&Ds[(32 * wave + nl) * EP + 8 * kk + 4 * half]);
                const f32x4 bf = *reinterpret_cast<const f32x4*>(&Wps[nl * EP + 8 * kk + 4 * half]);
                pa = __builtin_amdgcn_mfma_f32_32x32x2f32(af.x, bf.x, pa, 0, 0, 0);
                pa = __builtin_amdgcn_mfma_f32_32x32x2f32(af.y, bf.y, pa, 0, 0, 0);
                pa = __builtin_amdgcn_mfma_f32_32x32x2f32(af.z, bf.z, pa, 0, 0, 0);
                pa = __builtin_amdgcn_mfma_f32_32x32x2f32(af.w, bf.w, pa, 0, 0, 0);
            }
            if (nl < 16) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int o = 32 * wave + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const int gy = oy0 + o / TW, gx = ox0 + o % TW;
                    if (gy < a.H1 && gx < a.H1)
                        a.out[(((size_t)img * a.H1 + gy) * a.H1 + gx) * 16 + nl] = fmaf(pa[r], psc, pbi) + 0.f;
                }
            }
        }
    }
}

}  // namespace

bool adaf_mb_expand_dw_ok(int cin, int hid, int hw) { return cin % 8 == 0 && cin <= 32 && hid % 4 == 0 && hw >= 28; }

void adaf_launch_mb_expand_dw(MbFuseArgs a, int stride, hipStream_t s) {
    const int th = stride == 1 ? 8 : 3, tw = 8;
    a.tiles_x = (a.OW + tw - 1) / tw;
    a.tiles_y = (a.OH + th - 1) / th;
    const unsigned blocks = (unsigned)a.n * a.tiles_x * a.tiles_y;
    const size_t smem = sizeof(float) * ((size_t)160 * (a.cin + 4) + 128 * 36);
    if (stride == 1) hipLaunchKernelGGL((mb_expand_dw_kernel<1>), dim3(blocks), dim3(256), smem, s, a);
    else hipLaunchKernelGGL((mb_expand_dw_kernel<2>), dim3(blocks), dim3(256), smem, s, a);
}

void adaf_launch_mb_stem_b1(MbStemArgs a, int cus, hipStream_t s) {
    a.tiles_x = (a.H1 + 7) / 8;
    a.tiles_y = a.tiles_x;
    a.total_tiles = a.n * a.tiles_x * a.tiles_y;
    const int blocks = a.total_tiles < cus * 3 ? a.total_tiles : cus * 3;
    hipLaunchKernelGGL((mb_stem_b1_kernel<0>), dim3(blocks), dim3(256), 0, s, a);
}
