// Internal declarations shared by the kernel translation units and the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <string>

#include "../../include/adafocus.h"
#include <type_traits>

struct adaf_handle {
    int device = 0;
    int cus = 256;
    float* zeros = nullptr;  // device, 256 bytes of zeros
    unsigned* scan_timeouts = nullptr;   // device counter: blocks of persistent GRU scans whose grid barrier timed out (they NaN-poison
                                         // their outputs; adaf_gru_scan_timeouts() makes that visible to the host)
    int conv_pos_major = 1;  // k x k convs may use position-major tiles with padding-tap skipping (conv_gemm.hip PM kernels)
    int gru_persistent = 1;  // GRU scans as one persistent kernel where the shape allows (gru_scan.hip): 0 off, 1 on, 2 on + cooperative launch
    // At most scan_slots persistent scans may execute at once (their grid barriers need every block resident; how many
    // fit is ASKED of the runtime at adaf_create: scan_resident = blocks per CU x CUs): launch i waits for the
    // completion event of launch i - scan_slots.
    hipEvent_t scan_done[4] = {nullptr, nullptr, nullptr, nullptr};
    bool scan_used[4] = {false, false, false, false};
    int scan_next = 0;
    int scan_resident = 0;   // scan blocks the device can hold at once (occupancy query)
    int scan_slots = 1;      // min(4, scan_resident / blocks per scan)
    std::string err;
};

// Helper streams for networks that run two frame chunks side by side (adaf_mobilenetv2, adaf_effnet): one (stream, fork event, join event) per
// CALLER stream, so forwards issued from different streams do not serialise their second chunks on one shared helper.  At most kMax entries:
// a further caller stream takes over the least recently used entry (round 6; earlier the 17th stream got no helper and stopped pairing -- and
// entries of destroyed streams were never reused).  An entry is nothing but a helper stream and two events that are re-recorded on every call,
// so handing it to another caller stream (or to a recycled handle value) is harmless.  Guarded by a mutex: host threads may run the same
// network on different streams.  prepare() creates a few entries ahead (finalize): the first paired call may then happen under stream capture.
#include <mutex>
#include <vector>
struct AdafAuxPool {
    struct Aux { hipStream_t key = nullptr; hipStream_t stream = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr; unsigned long long used = 0; };
    static constexpr size_t kMax = 16;
    std::vector<Aux> entries;
    std::mutex mu;
    unsigned long long tick = 0;
    AdafAuxPool() { entries.reserve(kMax); }      // (entries never move: callers hold pointers into the vector)
    bool create(Aux* a) {
        return hipStreamCreateWithFlags(&a->stream, hipStreamNonBlocking) == hipSuccess &&
               hipEventCreateWithFlags(&a->ev_fork, hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&a->ev_join, hipEventDisableTiming) == hipSuccess;
    }
    void prepare(size_t count) {
        std::lock_guard<std::mutex> lk(mu);
        while (entries.size() < count && entries.size() < kMax) {
            Aux a;
            if (!create(&a)) return;
            entries.push_back(a);       // key == nullptr: free
        }
    }
    // the helper of caller stream `st` (nullptr if none could be created)
    Aux* get(hipStream_t st) {
        std::lock_guard<std::mutex> lk(mu);
        Aux* lru = nullptr;
        for (Aux& a : entries) {
            if (a.key == st && a.used) { a.used = ++tick; return &a; }
            if (!lru || a.used < lru->used) lru = &a;
        }
        if (entries.size() < kMax && !(lru && lru->used == 0)) {       // no free entry yet: make one
            Aux a;
            if (create(&a)) { entries.push_back(a); lru = &entries.back(); }
        }
        if (!lru) return nullptr;
        lru->key = st;
        lru->used = ++tick;
        return lru;
    }
    void destroy() {
        for (Aux& a : entries) {
            if (a.stream) (void)hipStreamDestroy(a.stream);
            if (a.ev_fork) (void)hipEventDestroy(a.ev_fork);
            if (a.ev_join) (void)hipEventDestroy(a.ev_join);
        }
        entries.clear();
    }
};

// Process-wide tuning / A-B switches of the library (adaf_set_global_option / adaf_get_global_option in include/adafocus.h; defaults = the plan
// every number in DESIGN.md is measured with).  They replace the environment variables of earlier rounds: tests flip them in-process.
struct AdafOptions {
    int conv_pool = 1;            // "conv_pool": global average pool in the last conv3's epilogue
    int mb_strip = 1;             // "mb_strip": strip-walking forms of the glancer's front kernels (mbstrip.hip, round 6); 0 = the wave-private tiles
    int mbv2_chunk = 512;         // "mbv2_chunk": frames per chunk of the MobileNetV2 forward
    int latency_rows = 1536;      // "latency_rows": GEMM rows up to which a new trunk sends convs to the small-batch form
    int latency_linear_rows = 128;  // "latency_linear_rows": the same for adaf_linear / GRU projections
    unsigned effnet_plan = 511u;  // "effnet_plan": ADAF_EF_PLAN_* bits
    unsigned effnet_fused_blocks = 0xffffffffu;   // "effnet_fused_blocks": MBConv blocks (bit = block index) the fused expand + depthwise launch may take
    int effnet_chunk = 1024;      // "effnet_chunk": frames per chunk of the EfficientNet forward
    int stem_rows = 1;            // "stem_rows": stem + max-pool over whole-width strips walked down the image (stem.hip, round 5); 0 = the tile form, 2 = also below one image per CU (tests)
    int split_stage1_f32 = 1;     // "split_stage1_f32": the split-bf16 trunk takes the fp32 pipe's fused stage-1 launches (api.hip run_trunk)
    int split_lean = 1;           // "split_lean": the split tiles' K loop with scalar-base DMA (conv_gemm.hip launch_glds; 0 = the pointer-per-lane form, A/B)
    int tsm_lean = 1;             // "tsm_lean": a conv1 with the fused temporal shift on the lean K loop (range-checked buffer DMA; conv_gemm.hip launch_glds; 0 = the per-lane select form, A/B)
    int gru_graph_persistent = 0; // "gru_graph_persistent": 1 = a stream capture keeps the persistent GRU scan (the caller guarantees exclusive use of the device
                                  // while the graph replays); 0 = captures take the launch-per-step form, which has no grid barrier to starve
    int gru_scan_slices = 2;      // "gru_scan_slices": clip slices a persistent GRU scan may be cut into (1 | 2; gru_scan.hip)
};
AdafOptions& adaf_options();

// fp32 -> fp16 STORAGE: what is converted is the ROUNDED fp32 result.  Left to itself the compiler folds `(_Float16)(a * b)` (and a * b + c,
// a + c) into v_fma_mixlo_f16 -- ONE rounding of the exact result -- for SOME elements of an unrolled epilogue and not for others (seen: 15 of
// the 16 accumulators of ef_expand_kernel; the last went v_mul_f32 + v_cvt_f16_f32), and the two disagree in the last bit about once per
// 2^12 values: a frame's features then depended on which MFMA row its pixels landed on, i.e. on its position in the batch, and the
// "bit-identical" kernel forms were identical only as long as the compiler made the same choice in each.  The empty asm makes the fp32
// value opaque to that fold (no instruction).
__device__ __forceinline__ _Float16 adaf_f16_of(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("" : "+v"(v));
#endif
    return (_Float16)v;
}

// Flattened description of one implicit-GEMM convolution launch.
struct ConvArgs {
    const float* x;
    const float* w;      // [N][K] with K ordered (kh, kw, cin)
    const unsigned short* wsp;  // optional: the same weights as three bf16 planes [3][N][K] (h, m, l parts; split tiles 6x)
    const float* scale;  // may be null (= 1)
    const float* bias;   // may be null (= 0)
    const float* res;    // may be null
    float* out;
    int M, N, K;         // GEMM extents: pixels out, channels out, kh*kw*cin
    int cin, H, W, OH, OW, KH, KW, stride, pad;
    int ldx, ldo, ldr;
    int act;
    int tsm_T, tsm_fold, tsm_hw;
    const float* zeros;  // >= 64 bytes of zeros (handle-owned): target of predicated-off loads
    int vec_epi;         // 1: 16-byte epilogue is legal (aligned out/res/scale/bias, strides % 4 == 0)
    int in16, out16, res16;   // half-precision STORAGE (N2): x and w / out / res hold fp16 (strides stay in elements)
    int pm_allow;        // k x k convs: position-major tiles with padding-tap skipping may be used (bit-identical; default 1)
    // two convs over the same input as ONE GEMM (filter banks concatenated along N): column tiles at n0 >= split_n write
    // out_b[m * ldo_b + n] (out_b already displaced by -split_n) with activation act_b; 0 = off
    int split_n;
    float* out_b;
    int ldo_b, act_b;
    int pm_images, pm_groups;   // set by the launcher: images in the batch, image groups of BM per pixel position
    int tiles_n;         // ceil(N / BN) for the chosen tile
    int nblocks;
    // global average pool in the epilogue (adaf_launch_conv_pool: the trunk's last conv3): a tile's rows are whole images of pool_hw
    // pixels (pool_rows = floor(128 / pool_hw) * pool_hw of them), the activated outputs are averaged per image in pixel order and
    // pool_out[image * pool_ld + n] is written instead of `out`
    int pool_hw, pool_rows, pool_ld;
    float* pool_out;
};

// mbconv.hip: fused expand 1x1 -> depthwise 3x3 of an inverted-residual block
struct MbFuseArgs {
    const float* x;      // [n][H][W][cin]
    int n, H, W, cin;
    const float* we;     // expand weights [hid][cin]
    const float* se;     // expand BN scale / bias [hid]
    const float* be;
    const float* wd;     // depthwise weights [3][3][hid]
    const float* sd;
    const float* bd;
    float* out;          // [n][OH][OW][hid]
    int hid, OH, OW;
    int tiles_x, tiles_y;
    const float* zeros;
    // whole-block kernel (expand -> depthwise -> project [+ identity]): project filter [cout][hid] + BN, identity rows, output
    const float* wp;
    const float* sp;
    const float* bp;
    const float* res;    // [n][OH][OW][cout] or nullptr
    float* out2;         // [n][OH][OW][cout]
    int cout;
    // mb_expand_dw_s_kernel (mbstrip.hip: expand -> depthwise of the stride-1 blocks with 64 / 96 input channels): the temporal shift of the
    // block's INPUT inside the pixel loads -- clips of tsm_T frames, the first tsm_fold channels from the next frame, the next tsm_fold from
    // the previous one, zeros at clip ends (the buffer descriptor spans the clip: a neighbour frame outside it is out of range) -- 0 = no shift
    int tsm_T, tsm_fold;
};
struct MbStemArgs {       // fused stem -> block 1 (t = 1: depthwise + project) of MobileNetV2
    const float* x;       // [n][S][S][4] pixel-major frames
    int n, S, H1;         // H1 = stem output size
    const float* ws;      // stem filter [32][3][3][4]
    const float* ss;      // BN scale / bias of the stem [32]
    const float* bs;
    const float* wd;      // depthwise [3][3][32]
    const float* sd;
    const float* bd;
    const float* wp;      // project [16][32]
    const float* sp;
    const float* bp;
    float* out;           // [n][H1][H1][16]
    int tiles_x, tiles_y, total_tiles;
    const float* zeros;
};
void adaf_launch_mb_stem_b1(MbStemArgs a, int cus, hipStream_t s);
bool adaf_mb_stem_b1_strip_ok(int S, int H1);                      // mbstrip.hip
void adaf_launch_mb_stem_b1_strip(MbStemArgs a, hipStream_t s);
bool adaf_mb_block_strip_ok(int cin, int hid, int cout, int stride, int h, int w);
void adaf_launch_mb_block_strip(MbFuseArgs a, hipStream_t s);
bool adaf_mb_expand_dw_strip_ok(int cin, int hid, int stride, int h, int w);     // expand -> depthwise on 14-column strips (64 / 96 input channels)
void adaf_launch_mb_expand_dw_strip(MbFuseArgs a, hipStream_t s);
bool adaf_mb_expand_dw_ok(int cin, int hid, int hw);
void adaf_launch_mb_expand_dw(MbFuseArgs a, int stride, hipStream_t s);
bool adaf_mb_block_ok(int cin, int hid, int cout, int stride, int hw);
void adaf_launch_mb_block(MbFuseArgs a, hipStream_t s);

#ifdef __HIPCC__
// Sum over the 64 lanes of a wave; every lane ends up with the same bits.  A butterfly (1, 2, 4, 8, 16, 32) whose first four steps stay on
// the VALU: quad permutes for 1 and 2, then the half-row and row MIRRORS for 4 and 8 -- after the step before them the lanes of a group hold
// identical bits, so taking the mirrored lane IS the xor exchange --, ds_swizzle for 16, v_permlane32_swap for 32.  (__shfl_xor compiles to
// ds_bpermute_b32: six dependent LDS round trips per sum -- the squeeze FC of the EfficientNet kernels spent more time in them than in its
// filter rows.)  All 64 lanes must be active.  The SE kernels of effnet.hip and mbconv_whole.hip share it: their gates agree bit for bit.
__device__ __forceinline__ float adaf_wave_sum(float v) {
    auto dpp = [](float x, auto ctrl_tag) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl_tag)::value, 0xf, 0xf, false));
    };
    v += dpp(v, std::integral_constant<int, 0xB1>());      // quad_perm [1, 0, 3, 2]
    v += dpp(v, std::integral_constant<int, 0x4E>());      // quad_perm [2, 3, 0, 1]
    v += dpp(v, std::integral_constant<int, 0x141>());     // row_half_mirror
    v += dpp(v, std::integral_constant<int, 0x140>());     // row_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));      // lane ^ 16
    const auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    return v + __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);                                // lane ^ 32
}
#endif

// mbconv_whole.hip: whole-image MBConv blocks (EfficientNet, fp16 storage)
size_t adaf_mbw_bfrag_halfs(int n, int k, bool even_tiles);
void adaf_launch_pack_bfrag_f16(const float* w, int n, int k, bool even_tiles, void* o, hipStream_t s);
int adaf_mbw_tap_row(int k);       // floats per channel in the depthwise operand rows: k*k taps + BN scale + BN bias, padded to 16 bytes
void adaf_launch_pack_dw_rows(const float* wd, const float* sd, const float* bd, int hid, int k, float* o, hipStream_t s);
bool adaf_mbw_eligible(int hw, int k, int stride, int cin, int hid, int cout, int sq);
int adaf_mbw_pad_before(int hw, int k, int stride);
bool adaf_launch_mbconv_whole(const void* x, int n, int hw, int stride, int cin, int hid, int cout, int sq, int k, const void* wef, const float* se,
                              const float* be, const float* wdl, const float* se_wr, const float* se_br,
                              const float* se_wet, const float* se_be, const void* wpf, const float* sp, const float* bp, bool skip, void* out,
                              hipStream_t s);

// gru_scan.hip
int adaf_gru_scan_blocks_per_cu();
bool adaf_gru_scan_persistent_ok(int batch, int hidden, int classes, int resident_blocks);
int adaf_gru_scan_groups(int batch, int resident_blocks);
struct AdafGruScanPlan {
    int groups;      // slices the scan is cut into (0: the barrier buffer is too small for a persistent scan)
    int bpad;        // pitch of the XCD-hierarchical barrier records in words (16 or 1), 0 = one counter per step
    size_t nzero;    // words of the barrier buffer the launch clears
};
AdafGruScanPlan adaf_gru_scan_plan(int batch, int steps, size_t bar_words, int resident_blocks);
hipError_t adaf_launch_gru_scan_persistent(const float* gi, const float* whh, const float* bhh, const float* h0, float* hs,
                                           unsigned* bar, const AdafGruScanPlan& plan, int batch, int steps, const float* fcw, const float* fcb,
                                           float* logits, float* last, int classes, bool cooperative, unsigned* timeouts, hipStream_t s);

// conv_gemm.hip
int adaf_launch_conv_gemm(const ConvArgs& a, int tile, int cus, hipStream_t s);  // returns chosen tile (>0) or <0
int adaf_pick_conv_tile(int M, int N, int K, int cus);
int adaf_launch_conv_lat(const ConvArgs& a, hipStream_t s);   // conv_lat.hip: small-batch form (tile id 95), 1 = launched, 0 = not eligible
int adaf_launch_conv_pool(ConvArgs a, int hw, float* pool_out, int pool_ld, hipStream_t s);   // 1 = launched (tile id 96), 0 = not eligible
int adaf_launch_conv_pool16(ConvArgs a, int hw, float* pool_out, int pool_ld, hipStream_t s); // fp16 operands (EfficientNet's head); same return
bool adaf_conv_tile_exists(int tile);   // is `tile` an id adaf_launch_conv_gemm has a kernel for
void adaf_launch_conv_naive(const ConvArgs& a, hipStream_t s);
// conv2 3x3 (64 -> 64) -> conv3 1x1 (+ identity, ReLU) [-> the next block's conv1 1x1] in one launch (stage 1 of the trunk);
// returns 0 or < 0 when the shape is not eligible
bool adaf_fused_tail_shift_ok(const ConvArgs& c2, int n3, int ldr, int tsm_T1, int tsm_fold1);
int adaf_launch_fused_tail(const ConvArgs& c2, const float* w3, const float* s3, const float* b3, const float* res, int ldr,
                           float* out, int n3, const float* w1n, const float* s1n, const float* b1n, float* out1, int n1,
                           hipStream_t s, int tsm_T1 = 0, int tsm_fold1 = 0);

// crop.hip
hipError_t adaf_launch_crop(const float* frames, int nf, int C, int H, int W, const float* act, int fpa, int P,
                            float* out, int layout, int32_t* coords, hipStream_t s);

hipError_t adaf_launch_crop_resize(const float* frames, int in4, int nf, int C, int H, int W, const float* act, const int32_t* sizes,
                                   int size_default, int fpa, int P, float* out, int layout, int32_t* coords, hipStream_t s);
hipError_t adaf_launch_resize_nearest(const float* frames, int in4, int nf, int C, int H, int W, int OH, int OW, float* out,
                                      int layout, hipStream_t s);

// misc_ops.hip
void adaf_launch_split_weight(const float* w, size_t count, unsigned short* planes, hipStream_t s);
void adaf_launch_pack_weight(const float* w, int cout, int cin, int kh, int kw, int cin_pad, float* o, hipStream_t s);
void adaf_launch_fold_bn(const float* g, const float* b, const float* m, const float* v, float eps, int c,
                         float* scale, float* bias, hipStream_t s);
void adaf_launch_maxpool(const float* x, int n, int h, int w, int c, float* o, hipStream_t s);
void adaf_launch_avgpool(const float* x, int n, int hw, int c, float* o, int ldo, hipStream_t s);
void adaf_launch_tshift(const float* x, int nt, int c, int hw, int T, int div, int layout, float* o, hipStream_t s);
void adaf_launch_gru_gates(const float* gi, int ldgi_t, const float* gh, const float* bhh, const float* hprev,
                           int ldh, float* hout, int ldo, int B, int Hd, hipStream_t s);
void adaf_launch_segment_mean(const float* logit, int B, int T, int C, const float* glog, int Tg, float* out,
                              hipStream_t s);
void adaf_launch_copy2d(const float* src, int lds, float* dst, int ldd, int rows, int cols, hipStream_t s);
void adaf_launch_pack_dw_weight(const float* w, int c, float* o, hipStream_t s);
void adaf_launch_dwconv3x3(const float* x, int n, int h, int w, int c, int stride, const float* wt, const float* scale,
                           const float* bias, int act, float* o, hipStream_t s);
void adaf_launch_grid_actions(const float* logits, int rows, int a, const float* table, long long* idx, float* act,
                              hipStream_t s);
void adaf_launch_ingest_u8(const uint8_t* u8, int clips, int T, int H, int W, const float* mean, const float* stdv,
                           float* out, hipStream_t s);
void adaf_launch_crop_nhwc4(const float* frames, int nf, int H, int W, const float* act, int fpa, int P, float* out,
                            int32_t* coords, hipStream_t s);
// half-precision storage (N2)
void adaf_launch_pack_weight_f16(const float* w, int cout, int cin, int kh, int kw, int cin_pad, void* o, hipStream_t s);
void adaf_launch_cast(const void* x, long long count, void* o, int to_f16, hipStream_t s);
void adaf_launch_dwconv3x3_f16(const void* x, int n, int h, int w, int c, int stride, const float* wt, const float* scale,
                               const float* bias, int act, void* o, hipStream_t s);
// stem.hip
void adaf_launch_pack_stem_weight(const float* w_oihw, float* wr, hipStream_t s);
size_t adaf_stem_weight_floats();
void adaf_launch_stem7x7(const float* x4, int n, int P, const float* wr, const float* scale, const float* bias, float* out,
                         int cus, hipStream_t s);
bool adaf_stem7x7_rows_ok(int P, int n, int cus);   // the strip-walking stem + pool kernel exists for this patch size and there are enough images to fill the device
// the same launch gathering its own windows from planar frames [n, 3, H, W] at floor(action * (H - P)) (get_patch folded in); false = not available
bool adaf_launch_stem7x7_pool_frames(const float* frames, bool pixel_major, int nframes, const float* act, int fpa, int H, int W, int n, int P,
                                     const float* wr, const float* scale, const float* bias, float* out, int cus, hipStream_t s);
bool adaf_stem7x7_pool_pays(int P);     // does the fused stem + max-pool launch beat the two separate ones at this patch size
void adaf_launch_stem7x7_pool(const float* x4, int n, int P, const float* wr, const float* scale, const float* bias, float* out,
                              int cus, hipStream_t s);
