// C-ABI layer (include/adafocus.h): argument validation, launch planning, the ResNet-50 trunk
// object and the GRU classifier driver.  No PyTorch types, no allocation in forward calls.
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "adaf_internal.h"

namespace {

int fail(adaf_handle* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    return code;
}

int hip_fail(adaf_handle* h, hipError_t e, const char* what) {
    return fail(h, ADAF_E_LAUNCH, "%s: %s", what, hipGetErrorString(e));
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline int conv_out(int in, int k, int stride, int pad) { return (in + 2 * pad - k) / stride + 1; }

// Validates a conv description and flattens it; returns ADAF_OK or an error code.
int make_conv_args(adaf_handle* h, const adaf_conv_params* p, const float* x, const float* w, const float* scale,
                   const float* bias, const float* res, float* out, ConvArgs* a) {
    if (!p || !x || !w || !out) return fail(h, ADAF_E_BADARG, "conv: null pointer");
    if (p->n <= 0 || p->h <= 0 || p->w <= 0 || p->cin <= 0 || p->cout <= 0 || p->kh <= 0 || p->kw <= 0 ||
        p->stride <= 0 || p->pad < 0)
        return fail(h, ADAF_E_BADARG, "conv: non-positive extent");
    if (p->cin % 4) return fail(h, ADAF_E_LAYOUT, "conv: cin=%d must be a multiple of 4 (pad the channel axis)", p->cin);
    const int ldx = p->ldx ? p->ldx : p->cin, ldo = p->ldo ? p->ldo : p->cout, ldr = p->ldr ? p->ldr : p->cout;
    if (ldx < p->cin || ldo < p->cout || ldr < p->cout) return fail(h, ADAF_E_BADARG, "conv: pixel stride smaller than channels");
    if (ldx % 4 || !aligned16(x) || !aligned16(w)) return fail(h, ADAF_E_LAYOUT, "conv: x/w must be 16-byte aligned, ldx % 4 == 0");
    if (p->act < ADAF_ACT_NONE || p->act > ADAF_ACT_SWISH) return fail(h, ADAF_E_BADARG, "conv: unknown activation %d", p->act);
    const int oh = conv_out(p->h, p->kh, p->stride, p->pad), ow = conv_out(p->w, p->kw, p->stride, p->pad);
    if (oh <= 0 || ow <= 0) return fail(h, ADAF_E_BADARG, "conv: empty output");
    const long long M = (long long)p->n * oh * ow;
    if (M > 0x7fffffffLL || (long long)p->n * p->h * p->w > 0x7fffffffLL) return fail(h, ADAF_E_BADARG, "conv: too many pixels");
    int fold = 0;
    if (p->tsm_segments > 0) {
        if (p->kh != 1 || p->kw != 1 || p->stride != 1 || p->pad != 0)
            return fail(h, ADAF_E_BADARG, "conv: fused temporal shift needs a 1x1 stride-1 conv");
        if (p->tsm_div <= 0 || p->n % p->tsm_segments) return fail(h, ADAF_E_BADARG, "conv: n %% tsm_segments != 0");
        fold = p->cin / p->tsm_div;
        if (fold % 4) return fail(h, ADAF_E_LAYOUT, "conv: temporal-shift fold=%d must be a multiple of 4", fold);
    }
    a->wsp = nullptr;
    a->in16 = a->out16 = a->res16 = 0;
    a->pm_allow = h->conv_pos_major; a->pm_images = a->pm_groups = 0;
    a->split_n = 0; a->out_b = nullptr; a->ldo_b = 0; a->act_b = 0;
    a->x = x; a->w = w; a->scale = scale; a->bias = bias; a->res = res; a->out = out;
    a->M = (int)M; a->N = p->cout; a->K = p->kh * p->kw * p->cin;
    a->cin = p->cin; a->H = p->h; a->W = p->w; a->OH = oh; a->OW = ow; a->KH = p->kh; a->KW = p->kw;
    a->stride = p->stride; a->pad = p->pad; a->ldx = ldx; a->ldo = ldo; a->ldr = ldr; a->act = p->act;
    a->tsm_T = p->tsm_segments > 0 ? p->tsm_segments : 0; a->tsm_fold = fold; a->tsm_hw = p->h * p->w;
    a->tiles_n = 0; a->nblocks = 0;
    a->zeros = h->zeros;
    a->vec_epi = (p->cout % 4 == 0 && ldo % 4 == 0 && ldr % 4 == 0 && aligned16(out) && (!res || aligned16(res)) &&
                  (!scale || aligned16(scale)) && (!bias || aligned16(bias))) ? 1 : 0;
    return ADAF_OK;
}

}  // namespace

AdafOptions& adaf_options() {
    static AdafOptions o;
    return o;
}

namespace {
struct OptKey { const char* name; double lo, hi; };
// (round 6: the switches that measured as no-gain and had no user are gone -- conv_lean, pm_fill, resize_lds_kb, mb_wave, dw3_variant, gru_barrier)
const OptKey kOptKeys[] = {{"conv_pool", 0, 1}, {"mb_strip", 0, 1}, {"mbv2_chunk", 1, 1 << 20}, {"latency_rows", 0, 1 << 30}, {"latency_linear_rows", 0, 1 << 30},
                           {"effnet_plan", 0, 511}, {"effnet_chunk", 1, 1 << 20}, {"gru_scan_slices", 1, 2}, {"effnet_fused_blocks", 0, 4294967295.0},
                           {"stem_rows", 0, 2}, {"split_stage1_f32", 0, 1}, {"gru_graph_persistent", 0, 1}, {"split_lean", 0, 1}, {"tsm_lean", 0, 1}};
int find_opt(const char* key) {
    if (!key) return -1;
    for (size_t i = 0; i < sizeof(kOptKeys) / sizeof(kOptKeys[0]); ++i)
        if (!strcmp(kOptKeys[i].name, key)) return (int)i;
    return -1;
}
}  // namespace

// ======================================================================================
extern "C" {

int adaf_version(void) { return ADAF_VERSION; }

int adaf_create(int device, adaf_handle** out) {
    if (!out) return ADAF_E_BADARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return ADAF_E_ARCH;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return ADAF_E_ARCH;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return ADAF_E_ARCH;  // the kernels are gfx950 code objects
    adaf_handle* h = new adaf_handle();
    h->device = device;
    h->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    int cur = 0;
    (void)hipGetDevice(&cur);
    (void)hipSetDevice(device);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&h->zeros), 256);
    if (e == hipSuccess) e = hipMemset(h->zeros, 0, 256);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&h->scan_timeouts), 256);
    if (e == hipSuccess) e = hipMemset(h->scan_timeouts, 0, 256);
    for (int i = 0; i < 4 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&h->scan_done[i], hipEventDisableTiming);
    if (e == hipSuccess) {
        // The occupancy API can report one block per CU too many for kernels in this SGPR range (the scan uses 90;
        // MI355X_MICROARCH.md "Correctness boundaries"), and a grid barrier must never count on a slot that is not there:
        // budget with one block per CU less than reported (the scan's 128 blocks then still fit twice over).
        const int per_cu = adaf_gru_scan_blocks_per_cu();
        h->scan_resident = (per_cu > 1 ? per_cu - 1 : per_cu) * h->cus;
        const int slots = h->scan_resident / 128;
        h->scan_slots = slots < 1 ? 1 : (slots > 4 ? 4 : slots);
    }
    (void)hipSetDevice(cur);
    if (e != hipSuccess) { delete h; return ADAF_E_NOMEM; }
    *out = h;
    return ADAF_OK;
}

int adaf_destroy(adaf_handle* h) {
    if (h && h->zeros) (void)hipFree(h->zeros);
    if (h && h->scan_timeouts) (void)hipFree(h->scan_timeouts);
    if (h)
        for (int i = 0; i < 4; ++i)
            if (h->scan_done[i]) (void)hipEventDestroy(h->scan_done[i]);
    delete h;
    return ADAF_OK;
}

const char* adaf_last_error(const adaf_handle* h) { return h ? h->err.c_str() : "null handle"; }
int adaf_device_cus(const adaf_handle* h) { return h ? h->cus : 0; }
int adaf_set_conv_pos_major(adaf_handle* h, int on) {
    if (!h) return ADAF_E_BADARG;
    h->conv_pos_major = on < 0 || on > 2 ? 1 : on;    // 2: position-major rows WITHOUT tap skipping (experiments)
    return ADAF_OK;
}
int adaf_set_global_option(const char* key, double value) {
    const int k = find_opt(key);
    if (k < 0 || !(value >= kOptKeys[k].lo && value <= kOptKeys[k].hi)) return ADAF_E_BADARG;
    AdafOptions& o = adaf_options();
    switch (k) {
        case 0: o.conv_pool = (int)value; break;
        case 1: o.mb_strip = (int)value; break;
        case 2: o.mbv2_chunk = (int)value; break;
        case 3: o.latency_rows = (int)value; break;
        case 4: o.latency_linear_rows = (int)value; break;
        case 5: o.effnet_plan = (unsigned)value; break;
        case 6: o.effnet_chunk = (int)value; break;
        case 7: o.gru_scan_slices = (int)value; break;
        case 8: o.effnet_fused_blocks = (unsigned)value; break;
        case 9: o.stem_rows = (int)value; break;
        case 10: o.split_stage1_f32 = (int)value; break;
        case 11: o.gru_graph_persistent = (int)value; break;
        case 12: o.split_lean = (int)value; break;
        default: o.tsm_lean = (int)value; break;
    }
    return ADAF_OK;
}

double adaf_get_global_option(const char* key) {
    const AdafOptions& o = adaf_options();
    switch (find_opt(key)) {
        case 0: return o.conv_pool;
        case 1: return o.mb_strip;
        case 2: return o.mbv2_chunk;
        case 3: return o.latency_rows;
        case 4: return o.latency_linear_rows;
        case 5: return o.effnet_plan;
        case 6: return o.effnet_chunk;
        case 7: return o.gru_scan_slices;
        case 8: return o.effnet_fused_blocks;
        case 9: return o.stem_rows;
        case 10: return o.split_stage1_f32;
        case 11: return o.gru_graph_persistent;
        case 12: return o.split_lean;
        case 13: return o.tsm_lean;
        default: return __builtin_nan("");
    }
}

int adaf_set_gru_persistent(adaf_handle* h, int on) {
    if (!h) return ADAF_E_BADARG;
    if (on < 0 || on > 2) return fail(h, ADAF_E_BADARG, "set_gru_persistent: mode %d (0 off, 1 on, 2 on + cooperative launch)", on);
    h->gru_persistent = on;
    return ADAF_OK;
}

int adaf_gru_scan_timeouts(adaf_handle* h, unsigned* count_out) {
    if (!h || !count_out) return ADAF_E_BADARG;
    int cur = 0;
    (void)hipGetDevice(&cur);
    (void)hipSetDevice(h->device);
    hipError_t e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(count_out, h->scan_timeouts, sizeof(unsigned), hipMemcpyDeviceToHost);
    (void)hipSetDevice(cur);
    return e == hipSuccess ? ADAF_OK : hip_fail(h, e, "gru_scan_timeouts");
}

// ---- crop ------------------------------------------------------------------------------
int adaf_crop_gather_f32(adaf_handle* h, const float* frames, int n_frames, int channels, int height, int width,
                         const float* action_yx, int n_actions, int frames_per_action, int patch, float* out,
                         int out_layout, int32_t* coords_out, void* stream) {
    if (!h) return ADAF_E_BADARG;
    if (n_frames == 0) return ADAF_OK;  // empty batch: nothing to gather
    if (!frames || !action_yx || !out) return fail(h, ADAF_E_BADARG, "crop: null pointer");
    if (n_frames < 0 || channels <= 0 || height <= 0 || width <= 0 || patch <= 0 || frames_per_action <= 0)
        return fail(h, ADAF_E_BADARG, "crop: non-positive extent");
    if (patch > height) return fail(h, ADAF_E_BADARG, "crop: patch %d larger than frame height %d", patch, height);
    if (width < height) return fail(h, ADAF_E_BADARG, "crop: width < height (the reference scales both axes by H-P)");
    if ((long long)n_actions * frames_per_action != n_frames)
        return fail(h, ADAF_E_BADARG, "crop: n_actions*frames_per_action=%lld != n_frames=%d",
                    (long long)n_actions * frames_per_action, n_frames);
    if (out_layout == ADAF_LAYOUT_NHWC4 && channels != 3) return fail(h, ADAF_E_LAYOUT, "crop: NHWC4 needs 3 channels");
    if (out_layout == ADAF_LAYOUT_NHWC && channels > 16) return fail(h, ADAF_E_LAYOUT, "crop: NHWC output supports <= 16 channels");
    if (out_layout < ADAF_LAYOUT_NCHW || out_layout > ADAF_LAYOUT_NHWC4) return fail(h, ADAF_E_LAYOUT, "crop: unknown layout");
    const int co = out_layout == ADAF_LAYOUT_NCHW ? 1 : (out_layout == ADAF_LAYOUT_NHWC4 ? 4 : channels);
    if ((size_t)8 * patch * co * sizeof(float) > 160 * 1024) return fail(h, ADAF_E_BADARG, "crop: patch too wide for the LDS tile");
    hipError_t e = adaf_launch_crop(frames, n_frames, channels, height, width, action_yx, frames_per_action, patch, out,
                                    out_layout, coords_out, (hipStream_t)stream);
    return e == hipSuccess ? ADAF_OK : hip_fail(h, e, "crop launch");
}

int adaf_crop_gather_nhwc4_f32(adaf_handle* h, const float* frames_nhwc4, int n_frames, int height, int width,
                               const float* action_yx, int n_actions, int frames_per_action, int patch, float* out_nhwc4,
                               int32_t* coords_out, void* stream) {
    if (!h) return ADAF_E_BADARG;
    if (n_frames == 0) return ADAF_OK;
    if (!frames_nhwc4 || !action_yx || !out_nhwc4) return fail(h, ADAF_E_BADARG, "crop_nhwc4: null pointer");
    if (n_frames < 0 || height <= 0 || width <= 0 || patch <= 0 || frames_per_action <= 0)
        return fail(h, ADAF_E_BADARG, "crop_nhwc4: non-positive extent");
    if (patch > height || width < height) return fail(h, ADAF_E_BADARG, "crop_nhwc4: patch > height or width < height");
    if ((long long)n_actions * frames_per_action != n_frames) return fail(h, ADAF_E_BADARG, "crop_nhwc4: n_actions*frames_per_action != n_frames");
    if (!aligned16(frames_nhwc4) || !aligned16(out_nhwc4)) return fail(h, ADAF_E_LAYOUT, "crop_nhwc4: 16-byte alignment required");
    adaf_launch_crop_nhwc4(frames_nhwc4, n_frames, height, width, action_yx, frames_per_action, patch, out_nhwc4, coords_out,
                           (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ADAF_OK : hip_fail(h, e, "crop_nhwc4 launch");
}

int adaf_crop_resize_f32(adaf_handle* h, const float* frames, int in_layout, int n_frames, int channels, int height, int width,
                         const float* action_yx, int n_actions, int frames_per_action, const int32_t* size_px,
                         int size_default, int patch, float* out, int out_layout, int32_t* coords_out, void* stream) {
    if (!h) return ADAF_E_BADARG;
    if (n_frames == 0) return ADAF_OK;
    if (!frames || !action_yx || !out) return fail(h, ADAF_E_BADARG, "crop_resize: null pointer");
    if (n_frames < 0 || channels <= 0 || height <= 0 || width <= 0 || patch <= 0 || frames_per_action <= 0)
        return fail(h, ADAF_E_BADARG, "crop_resize: non-positive extent");
    if (width < height) return fail(h, ADAF_E_BADARG, "crop_resize: width < height (the reference scales both axes by H-S)");
    if ((long long)n_actions * frames_per_action != n_frames)
        return fail(h, ADAF_E_BADARG, "crop_resize: n_actions*frames_per_action != n_frames");
    if (!size_px && (size_default < 1 || size_default > height))
        return fail(h, ADAF_E_BADARG, "crop_resize: window size %d outside [1, height=%d]", size_default, height);
    if (in_layout != ADAF_LAYOUT_NCHW && in_layout != ADAF_LAYOUT_NHWC4) return fail(h, ADAF_E_LAYOUT, "crop_resize: frames must be NCHW or NHWC4");
    if (out_layout < ADAF_LAYOUT_NCHW || out_layout > ADAF_LAYOUT_NHWC4) return fail(h, ADAF_E_LAYOUT, "crop_resize: unknown output layout");
    if ((in_layout == ADAF_LAYOUT_NHWC4 || out_layout == ADAF_LAYOUT_NHWC4) && channels != 3) return fail(h, ADAF_E_LAYOUT, "crop_resize: NHWC4 needs 3 channels");
    if (out_layout == ADAF_LAYOUT_NHWC && channels > 16) return fail(h, ADAF_E_LAYOUT, "crop_resize: NHWC output supports <= 16 channels");
    if ((in_layout == ADAF_LAYOUT_NHWC4 && !aligned16(frames)) || (out_layout == ADAF_LAYOUT_NHWC4 && !aligned16(out)))
        return fail(h, ADAF_E_LAYOUT, "crop_resize: 16-byte alignment required for pixel-major buffers");
    hipStream_t st = (hipStream_t)stream;
    if (!size_px && size_default == patch) {
        // scale 1: the resample IS the slice copy -- run the gather itself (bit-exact by construction)
        if (in_layout == ADAF_LAYOUT_NCHW)
            return adaf_crop_gather_f32(h, frames, n_frames, channels, height, width, action_yx, n_actions, frames_per_action, patch, out,
                                        out_layout, coords_out, stream);
        if (out_layout == ADAF_LAYOUT_NHWC4)
            return adaf_crop_gather_nhwc4_f32(h, frames, n_frames, height, width, action_yx, n_actions, frames_per_action, patch, out,
                                              coords_out, stream);
    }
    hipError_t e = adaf_launch_crop_resize(frames, in_layout == ADAF_LAYOUT_NHWC4, n_frames, channels, height, width, action_yx, size_px,
                                           size_default, frames_per_action, patch, out, out_layout, coords_out, st);
    return e == hipSuccess ? ADAF_OK : hip_fail(h, e, "crop_resize launch");
}

int adaf_resize_nearest_f32(adaf_handle* h, const float* frames, int in_layout, int n_frames, int channels, int height, int width,
                            int out_h, int out_w, float* out, int out_layout, void* stream) {
    if (!h) return ADAF_E_BADARG;
    if (n_frames == 0) return ADAF_OK;
    if (!frames || !out) return fail(h, ADAF_E_BADARG, "resize_nearest: null pointer");
    if (n_frames < 0 || channels <= 0 || height <= 0 || width <= 0 || out_h <= 0 || out_w <= 0)
        return fail(h, ADAF_E_BADARG, "resize_nearest: non-positive extent");
    if (in_layout != ADAF_LAYOUT_NCHW && in_layout != ADAF_LAYOUT_NHWC4) return fail(h, ADAF_E_LAYOUT, "resize_nearest: frames must be NCHW or NHWC4");
    if (out_layout < ADAF_LAYOUT_NCHW || out_layout > ADAF_LAYOUT_NHWC4) return fail(h, ADAF_E_LAYOUT, "resize_nearest: unknown output layout");
    if ((in_layout == ADAF_LAYOUT_NHWC4 || out_layout == ADAF_LAYOUT_NHWC4) && channels != 3) return fail(h, ADAF_E_LAYOUT, "resize_nearest: NHWC4 needs 3 channels");
    if ((in_layout == ADAF_LAYOUT_NHWC4 && !aligned16(frames)) || (out_layout == ADAF_LAYOUT_NHWC4 && !aligned16(out)))
        return fail(h, ADAF_E_LAYOUT, "resize_nearest: 16-byte alignment required for pixel-major buffers");
    hipError_t e = adaf_launch_resize_nearest(frames, in_layout == ADAF_LAYOUT_NHWC4, n_frames, channels, height, width, out_h, out_w, out,
                                              out_layout, (hipStream_t)stream);
    return e == hipSuccess ? ADAF_OK : hip_fail(h, e, "resize_nearest launch");
}

int adaf_ingest_u8_f32(adaf_handle* h, const uint8_t* clips_hwc, int n_clips, int frames, int height, int width,
                       const float* mean3, const float* std3, float* out_nhwc4, void* stream) {
    if (!h) return ADAF_E_BADARG;
    if (n_clips == 0) return ADAF_OK;
    if (!clips_hwc || !mean3 || !std3 || !out_nhwc4) return fail(h, ADAF_E_BADARG, "ingest: null pointer");
    if (n_clips < 0 || frames <= 0 || frames > 64 || height <= 0 || width <= 0) return fail(h, ADAF_E_BADARG, "ingest: non-positive extent (frames <= 64)");
    if (!aligned16(out_nhwc4)) return fail(h, ADAF_E_LAYOUT, "ingest: output must be 16-byte aligned");
    for (int c = 0; c < 3; ++c)
        if (!(std3[c] > 0.f)) return fail(h, ADAF_E_BADARG, "ingest: std must be positive");
    adaf_launch_ingest_u8(clips_hwc, n_clips, frames, height, width, mean3, std3, out_nhwc4, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ADAF_OK : hip_fail(h, e, "ingest launch");
}

// ---- conv ------------------------------------------------------------------------------
int adaf_conv2d_bn_act_f32(adaf_handle* h, const adaf_conv_params* p, const float* x, const float* w_ohwi,
                           const float* scale, const float* bias, const float* residual, float* out, void* stream) {
    if (!h) return ADAF_E_BADARG;
    ConvArgs a;
    int rc = make_conv_args(h, p, x, w_ohwi, scale, bias, residual, out, &a);
    if (rc) return rc;
    if (p->tile < 0 || (p->tile > 80 && p->tile != 95) || (p->tile && !adaf_conv_tile_exists(p->tile)))
        return fail(h, ADAF_E_BADARG, "conv: no kernel variant with tile id %d", p->tile);
    if (adaf_launch_conv_gemm(a, p->tile, h->cus, (hipStream_t)stream) < 0) return fail(h, ADAF_E_LAUNCH, "conv: no tile");
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ADAF_OK : hip_fail(h, e, "conv launch");
}

int adaf_conv2d_naive_f32(adaf_handle* h, const adaf_conv_params* p, const float* x, const float* w_ohwi,
                          const float* scale, const float* bias, const float* residual, float* out, void* stream) {
    if (!h) return ADAF_E_BADARG;
    ConvArgs a;
    int rc = make_conv_args(h, p, x, w_ohwi, scale, bias, residual, out, &a);
    if (rc) return rc;
    adaf_launch_conv_naive(a, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ADAF_OK : hip_fail(h, e, "naive conv launch");
}

// ---- half-precision storage (N2) -----------------------------------------------------------------------------
int adaf_conv2d_bn_act_f16(adaf_handle* h, const adaf_conv_params* p, const void* x, int x_dtype, const void* w_ohwi,
                           const float* scale, const float* bias, const void* residual_f16, void* out, int out_dtype,
                           void* stream) {
    if (!h) return ADAF_E_BADARG;
    if ((x_dtype != ADAF_DTYPE_F32 && x_dtype != ADAF_DTYPE_F16) || (out_dtype != ADAF_DTYPE_F32 && out_dtype != ADAF_DTYPE_F16))
        return fail(h, ADAF_E_BADARG, "conv_f16: unknown dtype");
    if (x_dtype == ADAF_DTYPE_F32 && out_dtype == ADAF_DTYPE_F32) return fail(h, ADAF_E_BADARG, "conv_f16: nothing is fp16; use adaf_conv2d_bn_act_f32");
    ConvArgs a;
    int rc = make_conv_args(h, p, static_cast<const float*>(x), static_cast<const float*>(w_ohwi), scale, bias,
                            static_cast<const float*>(residual_f16), static_cast<float*>(out), &a);
    if (rc) return rc;
    a.in16 = x_dtype == ADAF_DTYPE_F16;
    a.out16 = out_dtype == ADAF_DTYPE_F16;
    a.res16 = residual_f16 != nullptr;
    if (a.in16 && (p->cin % 8 || a.ldx % 8)) return fail(h, ADAF_E_LAYOUT, "conv_f16: fp16 operands need cin %% 8 == 0 (16-byte chunks)");
    if (!a.in16 && residual_f16) return fail(h, ADAF_E_BADARG, "conv_f16: a residual needs fp16 operands");
    if (p->tile && (p->tile < 81 || p->tile > 88)) return fail(h, ADAF_E_BADARG, "conv_f16: tile ids are 81..84, 88");
    if (adaf_launch_conv_gemm(a, p->tile, h->cus, (hipStream_t)stream) < 0)
        return fail(h, ADAF_E_LAYOUT, "conv_f16: shape not eligible (1x1: cin %% 8 == 0; k x k: cin %% 64 == 0)");
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ADAF_OK : hip_fail(h, e, "conv_f16 launch");
}

int adaf_pack_conv_weight_f16(adaf_handle* h, const float* w_oihw, int cout, int cin, int kh, int kw, int cin_pad,
                              void* w_ohwi_f16, void* stream) {
    if (!h) return ADAF_E_BADARG;
    if (!w_oihw || !w_ohwi_f16 || cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0 || cin_pad < cin || cin_pad % 8)
        return fail(h, ADAF_E_BADARG, "pack_f16: bad arguments (cin_pad %% 8 == 0)");
    adaf_launch_pack_weight_f16(w_oihw, cout, cin, kh, kw, cin_pad, w_ohwi_f16, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ADAF_OK : hip_fail(h, e, "pack_f16 launch");
}

int adaf_cast_f32_f16(adaf_handle* h, const void* src, size_t count, void* dst, int to_f16, void* stream) {
    if (!h) return ADAF_E_BADARG;
    if (count == 0) return ADAF_OK;
    if (!src || !dst) return fail(h, ADAF_E_BADARG, "cast: null pointer");
    adaf_launch_cast(src, (long long)count, dst, to_f16 ? 1 : 0, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ADAF_OK : hip_fail(h, e, "cast launch");
}

int adaf_dwconv3x3_bn_act_f16(adaf_handle* h, const void* x_f16, int n, int hh, int ww, int c, int stride, const float* w_33c,
                              const float* scale, const float* bias, int act, void* out_f16, void* stream) {
    if (!h) return ADAF_E_BADARG;
    if (!x_f16 || !w_33c || !scale || !bias || !out_f16 || n <= 0 || hh <= 0 || ww <= 0 || c <= 0) return fail(h, ADAF_E_BADARG, "dwconv_f16: bad arguments");
    if (stride != 1 && stride != 2) return fail(h, ADAF_E_BADARG, "dwconv_f16: stride must be 1 or 2");
    if (act < ADAF_ACT_NONE || act > ADAF_ACT_RELU6) return fail(h, ADAF_E_BADARG, "dwconv_f16: activation");
    if (c % 4 || (reinterpret_cast<uintptr_t>(x_f16) & 7) || (reinterpret_cast<uintptr_t>(out_f16) & 7) || !aligned16(w_33c) || !aligned16(scale) || !aligned16(bias))
        return fail(h, ADAF_E_LAYOUT, "dwconv_f16: c %% 4 == 0 and 8 / 16-byte alignment required");
    adaf_launch_dwconv3x3_f16(x_f16, n, hh, ww, c, stride, w_33c, scale, bias, act, out_f16, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ADAF_OK : hip_fail(h, e, "dwconv_f16 launch");
}

int adaf_pack_conv_weight_f32(adaf_handle* h, const float* w_oihw, int cout, int cin, int kh, int kw, int cin_pad,
                              float* w_ohwi, void* stream) {
    if (!h) return ADAF_E_BADARG;
    if (!w_oihw || !w_ohwi || cout <= 0 || cin <= 0 || kh <= 0 || kw <= 0 || cin_pad < cin || cin_pad % 4)
        return fail(h, ADAF_E_BADARG, "pack: bad arguments");
    adaf_launch_pack_weight(w_oihw, cout, cin, kh, kw, cin_pad, w_ohwi, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ADAF_OK : hip_fail(h, e, "pack launch");
}

int adaf_fold_bn_f32(adaf_handle* h, const float* gamma, const float* beta, const float* mean, const float* var,
                     float eps, int channels, float* scale, float* bias, void* stream) {
    if (!h) return ADAF_E_BADARG;
    if (!gamma || !beta || !mean || !var || !scale || !bias || channels <= 0) return fail(h, ADAF_E_BADARG, "fold_bn: bad arguments");
    adaf_launch_fold_bn(gamma, beta, mean, var, eps, channels, scale, bias, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ADAF_OK : hip_fail(h, e, "fold_bn launch");
}

// ---- pooling / shift / glue ------------------------------------------------------------
int adaf_maxpool3x3s2_f32(adaf_handle* h, const float* x, int n, int hh, int ww, int c, float* out, void* stream) {
    if (!h) return ADAF_E_BADARG;
    if (!x || !out || n <= 0 || hh <= 0 || ww <= 0 || c <= 0) return fail(h, ADAF_E_BADARG, "maxpool: bad arguments");
    if (c % 4 || !aligned16(x) || !aligned16(out)) return fail(h, ADAF_E_LAYOUT, "maxpool: c %% 4 and 16-byte alignment required");
    adaf_launch_maxpool(x, n, hh, ww, c, out, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ADAF_OK : hip_fail(h, e, "maxpool launch");
}

int adaf_global_avgpool_f32(adaf_handle* h, const float* x, int n, int hw, int c, float* out, int ldo, void* stream) {
    if (!h) return ADAF_E_BADARG;
    if (!x || !out || n <= 0 || hw <= 0 || c <= 0) return fail(h, ADAF_E_BADARG, "avgpool: bad arguments");
    if (ldo == 0) ldo = c;
    if (c % 4 || ldo % 4 || ldo < c || !aligned16(x) || !aligned16(out)) return fail(h, ADAF_E_LAYOUT, "avgpool: c,ldo %% 4 and 16-byte alignment required");
    adaf_launch_avgpool(x, n, hw, c, out, ldo, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ADAF_OK : hip_fail(h, e, "avgpool launch");
}

int adaf_temporal_shift_f32(adaf_handle* h, const float* x, int nt, int c, int hw, int n_segment, int fold_div,
                            int layout, float* out, void* stream) {
    if (!h) return ADAF_E_BADARG;
    if (nt == 0) return ADAF_OK;
    if (!x || !out || nt < 0 || c <= 0 || hw <= 0 || n_segment <= 0 || fold_div <= 0) return fail(h, ADAF_E_BADARG, "tshift: bad arguments");
    if (nt % n_segment) return fail(h, ADAF_E_BADARG, "tshift: nt=%d not a multiple of n_segment=%d", nt, n_segment);
    if (layout != ADAF_LAYOUT_NCHW && layout != ADAF_LAYOUT_NHWC) return fail(h, ADAF_E_LAYOUT, "tshift: layout");
    if (x == out) return fail(h, ADAF_E_BADARG, "tshift: in-place shift is not supported (as in the reference, temporal_shift.py:36-38)");
    adaf_launch_tshift(x, nt, c, hw, n_segment, fold_div, layout, out, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ADAF_OK : hip_fail(h, e, "tshift launch");
}

int adaf_copy2d_f32(adaf_handle* h, const float* src, int lds, float* dst, int ldd, int rows, int cols, void* stream) {
    if (!h) return ADAF_E_BADARG;
    if (rows == 0 || cols == 0) return ADAF_OK;
    if (!src || !dst || rows < 0 || cols < 0 || lds < cols || ldd < cols) return fail(h, ADAF_E_BADARG, "copy2d: bad arguments");
    adaf_launch_copy2d(src, lds, dst, ldd, rows, cols, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ADAF_OK : hip_fail(h, e, "copy2d launch");
}

}  // extern "C"

// ======================================================================================
// ResNet-50 trunk
// ======================================================================================
struct ConvLayer {
    std::string name;      // e.g. "layer1.0.conv1"
    std::string bn;        // e.g. "layer1.0.bn1"
    int cin, cout, k, stride, pad;
    int cin_pad;
    float* w = nullptr;    // packed OHWI
    unsigned short* wsp = nullptr;   // the same as three bf16 planes (ADAF_MATH_F32_SPLIT_BF16 only)
    float* scale = nullptr;
    float* bias = nullptr;
};

struct adaf_resnet50 {
    adaf_handle* h = nullptr;
    std::map<std::string, std::pair<const float*, size_t>> params;
    std::vector<ConvLayer> convs;  // [0] = stem, then per block conv1, conv2, conv3, (downsample)
    std::vector<int> tiles;        // per conv launch override
    // layer1.0's conv1 (64 -> 64) and downsample (64 -> 256) read the same map with the same 1x1 / stride-1 geometry: their
    // filter banks and BN affines concatenated along the output channels, for one launch instead of two (run_trunk)
    float* l10_w = nullptr;
    float* l10_scale = nullptr;
    float* l10_bias = nullptr;
    int math = ADAF_MATH_F32;      // ADAF_MATH_*: which matrix pipe the (non-stem) convs use
    bool fuse = true;              // stage 1: conv2 -> conv3 (-> next conv1) in one launch; stem + max-pool in one launch
    bool fuse_stem_always = false; // (tests, set_fusion(2)) take every fused launch at every size, not only where it is the faster plan
    bool tsm_block = false;        // temporal shift in front of the WHOLE Bottleneck (shift_place = 'block') instead of its conv1 ('blockres')
    int lat_rows = -1;             // convs with at most this many GEMM rows take the small-batch form (-1 = the "latency_rows" option, 1536)
    float* stem_w = nullptr;       // filter bank in the stem kernel's layout (stem.hip)
    bool finalized = false;
};

namespace {

const int kStageBlocks[4] = {3, 4, 6, 3};
const int kStagePlanes[4] = {64, 128, 256, 512};

void build_layers(adaf_resnet50* net) {
    net->convs.clear();
    net->convs.push_back({"conv1", "bn1", 3, 64, 7, 2, 3, 4});
    int inplanes = 64;
    for (int s = 0; s < 4; ++s) {
        const int planes = kStagePlanes[s];
        for (int b = 0; b < kStageBlocks[s]; ++b) {
            char pre[32];
            snprintf(pre, sizeof(pre), "layer%d.%d.", s + 1, b);
            const int stride = (b == 0 && s > 0) ? 2 : 1;
            const std::string p(pre);
            net->convs.push_back({p + "conv1", p + "bn1", inplanes, planes, 1, 1, 0, inplanes});
            net->convs.push_back({p + "conv2", p + "bn2", planes, planes, 3, stride, 1, planes});
            net->convs.push_back({p + "conv3", p + "bn3", planes, planes * 4, 1, 1, 0, planes});
            if (b == 0) net->convs.push_back({p + "downsample.0", p + "downsample.1", inplanes, planes * 4, 1, stride, 0, inplanes});
            inplanes = planes * 4;
        }
    }
    net->tiles.assign(net->convs.size(), 0);
}

struct Launch {   // one enqueued kernel of the forward pass, for the profiler
    double flops, bytes;
    int tile;
};

// Where the trunk's patches come from when the stem gathers them itself (adaf_resnet50_forward_frames)
struct FrameSrc {
    const float* frames;    // [nframes, 3, H, W] planar or [nframes, H, W, 4] pixel-major
    bool pixel_major;
    int nframes, H, W;
    const float* act;       // [n / fpa, 2] fp32 (y, x)
    int fpa;
};

// Walks the trunk; `rec` (optional) gets one hipEvent before each launch plus one at the end.
int run_trunk(adaf_resnet50* net, const float* x4, int n, int P, int tsm_T, int tsm_div, float* feat, int ldfeat,
              void* ws, size_t ws_bytes, hipStream_t st, std::vector<hipEvent_t>* rec, std::vector<Launch>* info, float* featmap = nullptr,
              const FrameSrc* src = nullptr) {
    adaf_handle* h = net->h;
    if (src) x4 = src->frames;
    if (!net->finalized) return fail(h, ADAF_E_STATE, "resnet50: finalize() has not been called");
    if (!x4 || !feat || !ws) return fail(h, ADAF_E_BADARG, "resnet50: null pointer");
    if (n <= 0 || P < 32) return fail(h, ADAF_E_BADARG, "resnet50: need n > 0 and patch >= 32");
    if (ldfeat == 0) ldfeat = 2048;
    if (ldfeat < 2048 || ldfeat % 4 || !aligned16(feat) || !aligned16(x4) || !aligned16(ws))
        return fail(h, ADAF_E_LAYOUT, "resnet50: ldfeat >= 2048, %% 4 == 0 and 16-byte aligned buffers required");
    if (tsm_T > 0 && n % tsm_T) return fail(h, ADAF_E_BADARG, "resnet50: n=%d not a multiple of tsm_segments=%d", n, tsm_T);
    const size_t need = adaf_resnet50_workspace_bytes(net, n, P);
    if (ws_bytes < need) return fail(h, ADAF_E_NOMEM, "resnet50: workspace %zu < %zu bytes", ws_bytes, need);

    // Small problems (BASELINE config 1: B*T = 16 patches -> 576 / 144 output pixels in stages 3 / 4): a conv whose GEMM has at most
    // `lat_rows` rows is as long as ONE accumulator chain on the engine, and runs on the latency form instead (conv_lat.hip:
    // v_mfma_f32_16x16x4_f32 chains, 3.2x shorter and bit-identical).  ADAF_LATENCY_ROWS: the row limit (0 = never).
    const int lat_rows = net->lat_rows >= 0 ? net->lat_rows : adaf_options().latency_rows;
    const bool lat_ok = lat_rows > 0 && tsm_T == 0 && net->math == ADAF_MATH_F32;     // (run_trunk's tsm_T: either shift placement)
    const bool fuse = net->fuse;
    // shift_place = 'block' (STH/ops/temporal_shift.py:104-121): TemporalShift wraps the whole Bottleneck, so conv1, the downsample
    // conv AND the identity see the shifted block input.  The shifted map is materialised in a sixth slab in front of every block and
    // the block then runs exactly as a block without a shift (every fused form applies, except the next block's conv1 riding in a
    // fused tail: it needs the SHIFTED output).  'blockres' (every shipped configuration) keeps the shift inside conv1's operand load.
    // The 64-plane stage is HBM-bound layer by layer, whatever the matrix pipe: its fused launches (conv1 + downsample of layer1.0;
    // conv2 -> conv3 -> next conv1 per block) exist on the fp32 pipe only, and the opt-in split-bf16 arithmetic takes them too --
    // 2.30 ms against 2.41 ms for the ten split launches they replace (option "split_stage1_f32" = 0: A/B).  Every product of such a
    // plan is either an exact fp32 FMA chain or the 6-product bf16 form: fp32-level accuracy throughout.
    const bool stage1_f32 = net->math == ADAF_MATH_F32 || (net->math == ADAF_MATH_F32_SPLIT_BF16 && adaf_options().split_stage1_f32);
    const bool tsm_block = net->tsm_block && tsm_T > 0;
    const int tsm_c1 = tsm_block ? 0 : tsm_T;     // the temporal shift conv1's operand load carries
    const int nslab = net->tsm_block ? 6 : 5;
    const size_t slab = adaf_resnet50_workspace_bytes(net, n, P) / (nslab * sizeof(float));  // largest activation, floats
    float* buf[6];
    for (int i = 0; i < nslab; ++i) buf[i] = static_cast<float*>(ws) + i * slab;

    auto mark = [&](double flops, double bytes, int tile) {
        if (rec) {   // events are created up front by the caller: recording is the only work between launches
            (void)hipEventRecord((*rec)[info->size()], st);
            info->push_back({flops, bytes, tile});
        }
    };
    int li = 0;
    bool pooled = false;           // the last conv3 averaged its map itself
    auto conv = [&](const float* in, int hh, int ww, int act, const float* res, float* out, int tsm, int* oh, int* ow,
                    int ldo) -> int {
        const ConvLayer& L = net->convs[li];
        adaf_conv_params p;
        memset(&p, 0, sizeof(p));
        p.n = n; p.h = hh; p.w = ww; p.cin = L.cin_pad; p.cout = L.cout; p.kh = p.kw = L.k; p.stride = L.stride; p.pad = L.pad;
        p.act = act; p.tsm_segments = tsm ? tsm_T : 0; p.tsm_div = tsm_div; p.ldo = ldo;
        // the split plan's stage 1 is on the fp32 pipe BY LAYER (convs 1..11: layer1.* and layer2.0.conv1, the launches the fused forms
        // cover), whether or not the fused launches are taken for this batch size / shift / fusion setting: a patch's features must not
        // depend on the batch it came in
        const bool split_here = net->math == ADAF_MATH_F32_SPLIT_BF16 && !(stage1_f32 && li >= 1 && li <= 11);
        p.tile = net->tiles[li] ? net->tiles[li] : (split_here ? 40 : 0);
        ConvArgs a;
        int rc = make_conv_args(h, &p, in, L.w, L.scale, L.bias, res, out, &a);
        if (rc) return rc;
        a.wsp = split_here ? L.wsp : nullptr;
        const double macs = (double)a.M * L.cout * L.k * L.k * L.cin;   // algorithmic: un-padded cin
        const double bytes = 4.0 * ((double)n * hh * ww * L.cin + (double)a.M * L.cout * (res ? 2 : 1) + (double)L.cout * L.k * L.k * L.cin);
        mark(2.0 * macs, bytes, 0);
        const bool want_lat = lat_ok && a.M <= lat_rows && li > 0 && !net->tiles[li];
        int used = adaf_launch_conv_gemm(a, want_lat ? 95 : p.tile, h->cus, st);
        if (used < 0 && want_lat) used = adaf_launch_conv_gemm(a, p.tile, h->cus, st);   // the latency form declined the shape: the engine takes it
        if (used < 0) return fail(h, ADAF_E_LAUNCH, "resnet50: no kernel for tile id %d (conv launch %d)", p.tile, li);
        if (info && !info->empty()) info->back().tile = used;
        *oh = a.OH; *ow = a.OW;
        ++li;
        return ADAF_OK;
    };

    int hh, ww, rc;
    bool gathered = false;
    if (src) {
        // the patches are windows of resident frames at floor(action * (H - P)) (get_patch, ACT/models/utils.py:37-51).  The strip-walking
        // stem kernel gathers them itself -- no gather launch, no patch tensor; where it does not apply (other patch sizes, small batches,
        // fusion off, a stem tile override) the gather runs into a free workspace slab first: same values either way.
        if (net->tiles[0] == 0 && net->fuse && adaf_stem7x7_rows_ok(P, n, h->cus)) {
            const ConvLayer& L = net->convs[0];
            const int oh = conv_out(P, 7, 2, 3), ph = conv_out(oh, 3, 2, 1);
            mark(2.0 * (double)n * oh * oh * 64 * 147, 4.0 * ((double)n * P * P * 3 + (double)n * ph * ph * 64 + 64.0 * 147), 94);
            gathered = adaf_launch_stem7x7_pool_frames(src->frames, src->pixel_major, src->nframes, src->act, src->fpa, src->H, src->W, n, P,
                                                       net->stem_w, L.scale, L.bias, buf[1], h->cus, st);
            if (gathered) { ++li; hh = ww = ph; }
            else if (rec) info->pop_back();
        }
        if (!gathered) {
            mark(0.0, 4.0 * 2.0 * (double)n * P * P * 3, 0);
            for (int g = 0; g * src->nframes < n; ++g) {      // one gather per action set over the same frames
                const float* act = src->act + (size_t)g * (src->nframes / src->fpa) * 2;
                float* dst = buf[2] + (size_t)g * src->nframes * P * P * 4;
                if (src->pixel_major) adaf_launch_crop_nhwc4(src->frames, src->nframes, src->H, src->W, act, src->fpa, P, dst, nullptr, st);
                else if (adaf_launch_crop(src->frames, src->nframes, 3, src->H, src->W, act, src->fpa, P, dst, ADAF_LAYOUT_NHWC4, nullptr, st) != hipSuccess)
                    return fail(h, ADAF_E_LAUNCH, "resnet50: gather launch");
            }
            x4 = buf[2];
        }
    }
    // stem: conv7x7 s2 + BN + ReLU -> maxpool 3x3 s2
    if (gathered) {
    } else if (net->tiles[0] == 0 && net->fuse && (adaf_stem7x7_pool_pays(P) || adaf_stem7x7_rows_ok(P, n, h->cus) || net->fuse_stem_always)) {   // both in one launch: the conv map never reaches HBM
        const ConvLayer& L = net->convs[0];
        hh = ww = conv_out(P, 7, 2, 3);
        const int ph = conv_out(hh, 3, 2, 1);
        mark(2.0 * (double)n * hh * ww * 64 * 147, 4.0 * ((double)n * P * P * 3 + (double)n * ph * ph * 64 + 64.0 * 147), 90);
        adaf_launch_stem7x7_pool(x4, n, P, net->stem_w, L.scale, L.bias, buf[1], h->cus, st);
        ++li;
        hh = ww = ph;
    } else {
        if (net->tiles[0] == 0) {   // specialised stem kernel (tile override != 0 runs it on the generic engine instead)
            const ConvLayer& L = net->convs[0];
            hh = ww = conv_out(P, 7, 2, 3);
            mark(2.0 * (double)n * hh * ww * 64 * 147, 4.0 * ((double)n * P * P * 3 + (double)n * hh * ww * 64 + 64.0 * 147), 40);
            adaf_launch_stem7x7(x4, n, P, net->stem_w, L.scale, L.bias, buf[0], h->cus, st);
            ++li;
        } else if ((rc = conv(x4, P, P, ADAF_ACT_RELU, nullptr, buf[0], 0, &hh, &ww, 0))) return rc;
        const int ph = conv_out(hh, 3, 2, 1), pw = conv_out(ww, 3, 2, 1);
        mark(0.0, 4.0 * ((double)n * hh * ww * 64 + (double)n * ph * pw * 64), 0);
        adaf_launch_maxpool(buf[0], n, hh, ww, 64, buf[1], st);
        hh = ph; ww = pw;
    }
    float* cur = buf[1];
    float* nxt = buf[0];
    float* t1 = buf[2];            // conv1 output
    float* t2 = buf[3];            // conv2 output (or, after a fused launch, the NEXT block's conv1 output)
    bool c1_done = false;          // the previous fused launch already produced this block's conv1 output (in t1)
    for (int s = 0; s < 4; ++s) {
        for (int b = 0; b < kStageBlocks[s]; ++b) {
            int h1 = hh, w1 = ww, h2, w2, h3, w3;
            if (tsm_block) {       // the block's input, shifted along its clip: conv1, downsample and identity all read this copy
                const int cin = net->convs[li].cin;
                mark(0.0, 8.0 * (double)n * hh * ww * cin, 0);
                adaf_launch_tshift(cur, n, cin, hh * ww, tsm_T, tsm_div, ADAF_LAYOUT_NHWC, buf[5], st);
                float* t = cur; cur = buf[5]; buf[5] = t;
            }
            const int i_c2 = li + 1, i_c3 = li + 2, i_ds = li + 3;
            const int i_next = li + 3 + (b == 0 ? 1 : 0);          // the next block's conv1 (or convs.size())
            // conv1 (1x1, optional fused temporal shift) -> conv2 (3x3, stride) -> conv3 (1x1) + identity
            bool ds_done = false;
            if (s == 0 && b == 0 && !c1_done && fuse && net->l10_w && tsm_c1 == 0 && stage1_f32 &&
                !net->tiles[li] && !net->tiles[i_ds]) {
                // layer1.0: conv1 and the downsample conv in ONE launch (same input, same 1x1 geometry; N = 64 + 256): the
                // pooled map is read once instead of twice and a 0.07 ms launch disappears.  128x64 tiles: column tile 0 is conv1.
                const ConvLayer &C1 = net->convs[li], &DS = net->convs[i_ds];
                adaf_conv_params p;
                memset(&p, 0, sizeof(p));
                p.n = n; p.h = hh; p.w = ww; p.cin = C1.cin_pad; p.cout = C1.cout + DS.cout; p.kh = p.kw = 1; p.stride = 1; p.pad = 0;
                p.act = ADAF_ACT_RELU;
                ConvArgs am;
                if ((rc = make_conv_args(h, &p, cur, net->l10_w, net->l10_scale, net->l10_bias, nullptr, t1, &am))) return rc;
                am.ldo = C1.cout;                       // conv1's output rows are 64 wide
                am.split_n = C1.cout;
                am.out_b = buf[4] - C1.cout;            // column n of the merged GEMM is channel n - 64 of the downsample output
                am.ldo_b = DS.cout;
                am.act_b = ADAF_ACT_NONE;
                const double M = (double)am.M;
                mark(2.0 * M * (C1.cout + DS.cout) * C1.cin, 4.0 * (M * C1.cin + M * (C1.cout + DS.cout) + (double)(C1.cout + DS.cout) * C1.cin), 93);
                if (adaf_launch_conv_gemm(am, 32, h->cus, st) < 0) return fail(h, ADAF_E_LAUNCH, "resnet50: merged layer1.0 launch");
                h1 = am.OH; w1 = am.OW;
                ++li;
                ds_done = true;
            } else if (!c1_done) {
                if ((rc = conv(cur, hh, ww, ADAF_ACT_RELU, nullptr, t1, tsm_c1 > 0, &h1, &w1, 0))) return rc;
            } else ++li;
            c1_done = false;
            const float* identity = cur;
            if (b == 0 && ds_done) identity = buf[4];
            else if (b == 0) {
                li = i_ds;
                int hd, wd;
                if ((rc = conv(cur, hh, ww, ADAF_ACT_NONE, nullptr, buf[4], 0, &hd, &wd, 0))) return rc;
                identity = buf[4];
            }
            li = i_c2;
            const ConvLayer& L2 = net->convs[i_c2];
            // (below ~1.5 row tiles of 128 pixels per CU the fused launch is a few dozen blocks that each run conv2, eight conv3 passes and
            //  the next conv1 one after the other -- 48-55 us at 8 patches against ~30 us for the three launches it replaces, each spread over
            //  more CUs; measured crossover between 64 and 96 patches of 96^2, tools/lat_plan_probe.py.  Bit-identical either way.)
            const bool fusable = fuse && stage1_f32 && L2.cin == 64 && L2.cout == 64 && L2.stride == 1 &&
                                 !net->tiles[i_c2] && !net->tiles[i_c3] && (net->fuse_stem_always || (long long)n * h1 * w1 * 2 >= 3ll * 128 * h->cus);
            if (fusable) {
                const ConvLayer& L3 = net->convs[i_c3];
                adaf_conv_params p;
                memset(&p, 0, sizeof(p));
                p.n = n; p.h = h1; p.w = w1; p.cin = L2.cin_pad; p.cout = L2.cout; p.kh = p.kw = L2.k; p.stride = 1; p.pad = L2.pad;
                p.act = ADAF_ACT_RELU;
                ConvArgs a2;
                if ((rc = make_conv_args(h, &p, t1, L2.w, L2.scale, L2.bias, nullptr, t2, &a2))) return rc;
                // the next block's conv1 rides along unless it carries a temporal shift or a tile override
                // ('block' placement: the next block reads a shifted COPY of this block's output, so its conv1 cannot ride; 'blockres': it rides
                //  with the shift as a row offset inside the tile, whole clips per tile -- adaf_fused_tail_shift_ok)
                const ConvLayer* Ln = ((tsm_T == 0 || tsm_c1 > 0) && i_next < (int)net->convs.size() && !net->tiles[i_next]) ? &net->convs[i_next] : nullptr;
                if (Ln && !(Ln->k == 1 && Ln->stride == 1 && Ln->cin == L3.cout && (Ln->cout == 64 || Ln->cout == 128))) Ln = nullptr;
                const int tsm_n1 = (Ln && tsm_c1 > 0) ? tsm_c1 : 0, fold_n1 = Ln ? Ln->cin / (tsm_div > 0 ? tsm_div : 8) : 0;
                if (tsm_n1 && !adaf_fused_tail_shift_ok(a2, L3.cout, L3.cout, tsm_n1, fold_n1)) Ln = nullptr;
                const double M = (double)a2.M;
                double macs = M * 64 * 9 * 64 + M * L3.cout * 64 + (Ln ? M * Ln->cout * L3.cout : 0.0);
                double bytes = 4.0 * (M * 64 + 2.0 * M * L3.cout + (Ln ? M * Ln->cout : 0.0) + 64.0 * 576 + 64.0 * L3.cout +
                                      (Ln ? (double)Ln->cout * L3.cout : 0.0));
                mark(2.0 * macs, bytes, Ln ? 92 : 91);
                if (adaf_launch_fused_tail(a2, L3.w, L3.scale, L3.bias, identity, L3.cout, nxt, L3.cout, Ln ? Ln->w : nullptr,
                                           Ln ? Ln->scale : nullptr, Ln ? Ln->bias : nullptr, t2, Ln ? Ln->cout : 0, st, Ln ? tsm_n1 : 0, fold_n1) < 0)
                    return fail(h, ADAF_E_LAUNCH, "resnet50: fused bottleneck tail rejected the shape");
                h3 = a2.OH; w3 = a2.OW;
                if (Ln) { float* t = t1; t1 = t2; t2 = t; c1_done = true; }
            } else {
                if ((rc = conv(t1, h1, w1, ADAF_ACT_RELU, nullptr, t2, 0, &h2, &w2, 0))) return rc;
                const bool last = s == 3 && b == kStageBlocks[3] - 1;
                if (last && fuse && !rec && !featmap && net->math == ADAF_MATH_F32 && !net->tiles[li] && !(lat_ok && n * h2 * w2 <= lat_rows)) {
                    // the trunk's last conv3: the global average pool rides in its epilogue (conv_epilogue_pool) -- no 2048-channel map,
                    // no pooling launch -- when whole images fill its row tiles (3x3 / 4x4 / 5x5 maps); bit-identical to conv + pool
                    const ConvLayer& L3 = net->convs[li];
                    adaf_conv_params p;
                    memset(&p, 0, sizeof(p));
                    p.n = n; p.h = h2; p.w = w2; p.cin = L3.cin_pad; p.cout = L3.cout; p.kh = p.kw = 1; p.stride = 1; p.pad = 0;
                    p.act = ADAF_ACT_RELU;
                    ConvArgs a3;
                    if ((rc = make_conv_args(h, &p, t2, L3.w, L3.scale, L3.bias, identity, nxt, &a3))) return rc;
                    // (the profiled pass -- one event in front of every launch -- keeps conv + pool: its per-launch table stays comparable)
                    if (adaf_launch_conv_pool(a3, h2 * w2, feat, ldfeat, st)) {
                        pooled = true;
                        h3 = h2; w3 = w2;
                        ++li;
                    }
                }
                if (!pooled && (rc = conv(t2, h2, w2, ADAF_ACT_RELU, identity, nxt, 0, &h3, &w3, 0))) return rc;
            }
            li = i_next;
            hh = h3; ww = w3;
            float* t = cur; cur = nxt; nxt = t;
        }
    }
    if (featmap)        // get_featmap(pooled=False): the last block's map leaves the workspace (NHWC)
        (void)hipMemcpyAsync(featmap, cur, (size_t)n * hh * ww * 2048 * sizeof(float), hipMemcpyDeviceToDevice, st);
    if (!pooled) {
        mark(0.0, 4.0 * ((double)n * hh * ww * 2048 + (double)n * 2048), 0);
        adaf_launch_avgpool(cur, n, hh * ww, 2048, feat, ldfeat, st);
    }
    if (rec) (void)hipEventRecord((*rec)[info->size()], st);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ADAF_OK : hip_fail(h, e, "resnet50 forward");
}

}  // namespace

extern "C" {

int adaf_resnet50_create(adaf_handle* h, adaf_resnet50** out) {
    if (!h || !out) return ADAF_E_BADARG;
    adaf_resnet50* net = new adaf_resnet50();
    net->h = h;
    build_layers(net);
    *out = net;
    return ADAF_OK;
}

int adaf_resnet50_destroy(adaf_resnet50* net) {
    if (!net) return ADAF_OK;
    if (net->stem_w) (void)hipFree(net->stem_w);
    if (net->l10_w) (void)hipFree(net->l10_w);
    if (net->l10_scale) (void)hipFree(net->l10_scale);
    if (net->l10_bias) (void)hipFree(net->l10_bias);
    for (auto& L : net->convs) {
        if (L.w) (void)hipFree(L.w);
        if (L.wsp) (void)hipFree(L.wsp);
        if (L.scale) (void)hipFree(L.scale);
        if (L.bias) (void)hipFree(L.bias);
    }
    delete net;
    return ADAF_OK;
}

int adaf_resnet50_set_param(adaf_resnet50* net, const char* name, const float* dev_ptr, size_t numel) {
    if (!net || !name || !dev_ptr) return ADAF_E_BADARG;
    net->params[name] = std::make_pair(dev_ptr, numel);
    net->finalized = false;
    return ADAF_OK;
}

// Three bf16 planes of every packed filter bank except the stem's (idempotent; used by the split tiles 6x).
static int split_weights(adaf_resnet50* net, void* stream) {
    adaf_handle* h = net->h;
    hipStream_t st = (hipStream_t)stream;
    for (size_t i = 1; i < net->convs.size(); ++i) {
        ConvLayer& L = net->convs[i];
        const size_t wn = (size_t)L.cout * L.k * L.k * L.cin_pad;
        if (!L.wsp && hipMalloc(reinterpret_cast<void**>(&L.wsp), 3 * wn * sizeof(unsigned short)) != hipSuccess)
            return fail(h, ADAF_E_NOMEM, "resnet50: hipMalloc split weights");
        adaf_launch_split_weight(L.w, wn, L.wsp, st);
    }
    hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess) return hip_fail(h, e, "resnet50 split weights");
    return ADAF_OK;
}

int adaf_resnet50_finalize(adaf_resnet50* net, void* stream) {
    if (!net) return ADAF_E_BADARG;
    adaf_handle* h = net->h;
    hipStream_t st = (hipStream_t)stream;
    auto get = [&](const std::string& key, size_t numel, const float** p) -> int {
        auto it = net->params.find(key);
        if (it == net->params.end()) return fail(h, ADAF_E_STATE, "resnet50: missing parameter '%s'", key.c_str());
        if (it->second.second != numel)
            return fail(h, ADAF_E_BADARG, "resnet50: '%s' has %zu elements, expected %zu", key.c_str(), it->second.second, numel);
        *p = it->second.first;
        return ADAF_OK;
    };
    for (auto& L : net->convs) {
        const float *w, *g, *b, *m, *v;
        int rc;
        if ((rc = get(L.name + ".weight", (size_t)L.cout * L.cin * L.k * L.k, &w))) return rc;
        if ((rc = get(L.bn + ".weight", L.cout, &g))) return rc;
        if ((rc = get(L.bn + ".bias", L.cout, &b))) return rc;
        if ((rc = get(L.bn + ".running_mean", L.cout, &m))) return rc;
        if ((rc = get(L.bn + ".running_var", L.cout, &v))) return rc;
        const size_t wn = (size_t)L.cout * L.k * L.k * L.cin_pad;
        if (!L.w && hipMalloc(reinterpret_cast<void**>(&L.w), wn * sizeof(float)) != hipSuccess) return fail(h, ADAF_E_NOMEM, "resnet50: hipMalloc weights");
        if (!L.scale && hipMalloc(reinterpret_cast<void**>(&L.scale), L.cout * sizeof(float)) != hipSuccess) return fail(h, ADAF_E_NOMEM, "resnet50: hipMalloc scale");
        if (!L.bias && hipMalloc(reinterpret_cast<void**>(&L.bias), L.cout * sizeof(float)) != hipSuccess) return fail(h, ADAF_E_NOMEM, "resnet50: hipMalloc bias");
        adaf_launch_pack_weight(w, L.cout, L.cin, L.k, L.k, L.cin_pad, L.w, st);
        adaf_launch_fold_bn(g, b, m, v, 1e-5f, L.cout, L.scale, L.bias, st);
        if (&L == &net->convs[0]) {
            if (!net->stem_w && hipMalloc(reinterpret_cast<void**>(&net->stem_w), adaf_stem_weight_floats() * sizeof(float)) != hipSuccess)
                return fail(h, ADAF_E_NOMEM, "resnet50: hipMalloc stem weights");
            adaf_launch_pack_stem_weight(w, net->stem_w, st);
        }
    }
    {   // conv1 ++ downsample of layer1.0 (convs[1] and convs[4]: 1x1, stride 1, 64 input channels)
        const ConvLayer &C1 = net->convs[1], &DS = net->convs[4];
        if (C1.k == 1 && DS.k == 1 && C1.stride == 1 && DS.stride == 1 && C1.cin_pad == DS.cin_pad && C1.cout % 64 == 0) {
            const size_t n1 = (size_t)C1.cout * C1.cin_pad, n2 = (size_t)DS.cout * DS.cin_pad;
            const int cm = C1.cout + DS.cout;
            if ((!net->l10_w && hipMalloc(reinterpret_cast<void**>(&net->l10_w), (n1 + n2) * sizeof(float)) != hipSuccess) ||
                (!net->l10_scale && hipMalloc(reinterpret_cast<void**>(&net->l10_scale), cm * sizeof(float)) != hipSuccess) ||
                (!net->l10_bias && hipMalloc(reinterpret_cast<void**>(&net->l10_bias), cm * sizeof(float)) != hipSuccess))
                return fail(h, ADAF_E_NOMEM, "resnet50: hipMalloc merged layer1.0 filters");
            (void)hipMemcpyAsync(net->l10_w, C1.w, n1 * sizeof(float), hipMemcpyDeviceToDevice, st);
            (void)hipMemcpyAsync(net->l10_w + n1, DS.w, n2 * sizeof(float), hipMemcpyDeviceToDevice, st);
            (void)hipMemcpyAsync(net->l10_scale, C1.scale, C1.cout * sizeof(float), hipMemcpyDeviceToDevice, st);
            (void)hipMemcpyAsync(net->l10_scale + C1.cout, DS.scale, DS.cout * sizeof(float), hipMemcpyDeviceToDevice, st);
            (void)hipMemcpyAsync(net->l10_bias, C1.bias, C1.cout * sizeof(float), hipMemcpyDeviceToDevice, st);
            (void)hipMemcpyAsync(net->l10_bias + C1.cout, DS.bias, DS.cout * sizeof(float), hipMemcpyDeviceToDevice, st);
        }
    }
    hipError_t e = hipStreamSynchronize(st);
    if (e != hipSuccess) return hip_fail(h, e, "resnet50 finalize");
    net->finalized = true;
    if (net->math == ADAF_MATH_F32_SPLIT_BF16) return split_weights(net, stream);
    return ADAF_OK;
}

size_t adaf_resnet50_workspace_bytes(const adaf_resnet50* net, int n, int patch) {
    if (n <= 0 || patch <= 0) return 0;
    // five slabs (block input, block output, two bottleneck temporaries, downsample branch), each as
    // large as the biggest activation: the stem output or the first stage's 256-channel map; a sixth for the
    // shifted block input when the temporal shift wraps whole blocks (adaf_resnet50_set_shift_place)
    const int s1 = conv_out(patch, 7, 2, 3), s2 = conv_out(s1, 3, 2, 1);
    const size_t a = (size_t)s1 * s1 * 64, b = (size_t)s2 * s2 * 256;
    return (size_t)((net && net->tsm_block) ? 6 : 5) * n * (a > b ? a : b) * sizeof(float);
}

int adaf_resnet50_forward(adaf_resnet50* net, const float* patches_nhwc4, int n, int patch, int tsm_segments,
                          int tsm_div, float* feat, int ldfeat, void* ws, size_t ws_bytes, void* stream) {
    if (!net) return ADAF_E_BADARG;
    return run_trunk(net, patches_nhwc4, n, patch, tsm_segments, tsm_div, feat, ldfeat, ws, ws_bytes, (hipStream_t)stream,
                     nullptr, nullptr);
}

int adaf_resnet50_forward_frames(adaf_resnet50* net, const float* frames, int frames_layout, int n_frames, int height, int width,
                                 const float* action_yx, int n_actions, int frames_per_action, int patch, int tsm_segments, int tsm_div,
                                 float* feat, int ldfeat, void* ws, size_t ws_bytes, void* stream) {
    if (!net) return ADAF_E_BADARG;
    adaf_handle* h = net->h;
    if (!frames || !action_yx || n_frames <= 0 || n_actions <= 0 || frames_per_action <= 0)
        return fail(h, ADAF_E_BADARG, "resnet50 forward_frames: null pointer or empty batch");
    if (frames_layout != ADAF_LAYOUT_NCHW && frames_layout != ADAF_LAYOUT_NHWC4)
        return fail(h, ADAF_E_LAYOUT, "resnet50 forward_frames: frames must be NCHW (3 planes) or NHWC4");
    if (n_frames % frames_per_action) return fail(h, ADAF_E_BADARG, "resnet50 forward_frames: n_frames %% frames_per_action != 0");
    const int per_set = n_frames / frames_per_action;
    if (n_actions % per_set) return fail(h, ADAF_E_BADARG, "resnet50 forward_frames: n_actions=%d is not a multiple of n_frames / frames_per_action=%d", n_actions, per_set);
    // (get_patch takes its size from the frames' HEIGHT and scales both axes by H - P, ACT/models/utils.py:40-42: frames wider than high work as in
    //  adaf_crop_gather_f32 -- x is clamped to W - P; narrower ones would read past a row)
    if (width < height) return fail(h, ADAF_E_BADARG, "resnet50 forward_frames: width %d < height %d (get_patch scales both axes by H - P)", width, height);
    if (patch > height || patch < 32) return fail(h, ADAF_E_BADARG, "resnet50 forward_frames: patch %d outside [32, %d]", patch, height);
    if (!aligned16(frames)) return fail(h, ADAF_E_LAYOUT, "resnet50 forward_frames: frames must be 16-byte aligned");
    FrameSrc src{frames, frames_layout == ADAF_LAYOUT_NHWC4, n_frames, height, width, action_yx, frames_per_action};
    const int n = (n_actions / per_set) * n_frames;        // one patch per (action set, frame)
    return run_trunk(net, frames, n, patch, tsm_segments, tsm_div, feat, ldfeat, ws, ws_bytes, (hipStream_t)stream, nullptr, nullptr, nullptr, &src);
}

int adaf_resnet50_map_size(int patch) {
    if (patch < 32) return 0;
    int s = (patch + 6 - 7) / 2 + 1;          // conv1 7x7 / 2 / pad 3
    s = (s + 2 - 3) / 2 + 1;                  // max-pool 3x3 / 2 / pad 1
    for (int i = 0; i < 3; ++i) s = (s + 2 - 3) / 2 + 1;      // layer2-4: 3x3 / 2 / pad 1
    return s;
}

int adaf_resnet50_forward_map(adaf_resnet50* net, const float* patches_nhwc4, int n, int patch, int tsm_segments, int tsm_div,
                              float* featmap_nhwc, float* feat, int ldfeat, void* ws, size_t ws_bytes, void* stream) {
    if (!net) return ADAF_E_BADARG;
    if (!featmap_nhwc || !aligned16(featmap_nhwc)) return fail(net->h, ADAF_E_BADARG, "resnet50: forward_map needs a 16-byte aligned map buffer");
    return run_trunk(net, patches_nhwc4, n, patch, tsm_segments, tsm_div, feat, ldfeat, ws, ws_bytes, (hipStream_t)stream, nullptr, nullptr,
                     featmap_nhwc);
}

int adaf_resnet50_launch_count(const adaf_resnet50* net) { return net ? (int)net->convs.size() + 2 + (net->tsm_block ? 16 : 0) : 0; }

int adaf_resnet50_forward_profiled(adaf_resnet50* net, const float* patches_nhwc4, int n, int patch, int tsm_segments,
                                   int tsm_div, float* feat, int ldfeat, void* ws, size_t ws_bytes, void* stream,
                                   float* launch_ms, double* launch_flops, double* launch_bytes, int* launch_tile) {
    if (!net || !launch_ms || !launch_flops || !launch_bytes || !launch_tile) return ADAF_E_BADARG;
    std::vector<hipEvent_t> ev(adaf_resnet50_launch_count(net) + 1);
    for (auto& e : ev) (void)hipEventCreate(&e);
    std::vector<Launch> info;
    int rc = run_trunk(net, patches_nhwc4, n, patch, tsm_segments, tsm_div, feat, ldfeat, ws, ws_bytes, (hipStream_t)stream,
                       &ev, &info);
    if (rc == ADAF_OK) {
        hipError_t e = hipStreamSynchronize((hipStream_t)stream);
        if (e != hipSuccess) rc = hip_fail(net->h, e, "resnet50 profiled forward");
    }
    if (rc == ADAF_OK && info.size() + 1 <= ev.size()) {
        for (size_t i = info.size(); i + 1 < ev.size(); ++i) {   // fused plans use fewer launches than the table holds
            launch_ms[i] = 0.f; launch_flops[i] = 0.0; launch_bytes[i] = 0.0; launch_tile[i] = -1;
        }
        for (size_t i = 0; i < info.size(); ++i) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ev[i], ev[i + 1]);
            launch_ms[i] = ms;
            launch_flops[i] = info[i].flops;
            launch_bytes[i] = info[i].bytes;
            launch_tile[i] = info[i].tile;
        }
    }
    for (auto e : ev) (void)hipEventDestroy(e);
    return rc;
}

int adaf_resnet50_set_tiles(adaf_resnet50* net, const int* tile, int count) {
    if (!net || !tile || count != (int)net->convs.size()) return ADAF_E_BADARG;
    for (int i = 0; i < count; ++i) {
        if (tile[i] < 0 || tile[i] > 80 || (tile[i] && !adaf_conv_tile_exists(tile[i])))
            return fail(net->h, ADAF_E_BADARG, "set_tiles: no kernel variant with id %d", tile[i]);
        net->tiles[i] = tile[i];
    }
    return ADAF_OK;
}

int adaf_resnet50_set_fusion(adaf_resnet50* net, int on) {
    if (!net) return ADAF_E_BADARG;
    net->fuse = on != 0;
    net->fuse_stem_always = on == 2;
    return ADAF_OK;
}

int adaf_resnet50_set_shift_place(adaf_resnet50* net, int place) {
    if (!net) return ADAF_E_BADARG;
    if (place != ADAF_SHIFT_BLOCKRES && place != ADAF_SHIFT_BLOCK) return fail(net->h, ADAF_E_BADARG, "set_shift_place: unknown placement %d", place);
    net->tsm_block = place == ADAF_SHIFT_BLOCK;
    return ADAF_OK;
}

int adaf_resnet50_set_latency_rows(adaf_resnet50* net, int rows) {
    if (!net) return ADAF_E_BADARG;
    net->lat_rows = rows;          // < 0: back to the default
    return ADAF_OK;
}

int adaf_resnet50_set_math(adaf_resnet50* net, int mode) {
    if (!net) return ADAF_E_BADARG;
    if (mode != ADAF_MATH_F32 && mode != ADAF_MATH_F32_SPLIT_BF16) return fail(net->h, ADAF_E_BADARG, "set_math: unknown mode %d", mode);
    net->math = mode;
    if (mode == ADAF_MATH_F32_SPLIT_BF16 && net->finalized) return split_weights(net, nullptr);
    return ADAF_OK;
}

// ======================================================================================
// GRU classifier / linear + temporal mean
// ======================================================================================
size_t adaf_gru_cls_workspace_bytes(int batch, int steps, int hidden) {
    if (batch <= 0 || steps <= 0 || hidden <= 0) return 0;
    // gi [B*T, 3H] + gh [B, 3H] + hidden states [B, T, H]
    return ((size_t)batch * steps * 3 * hidden + (size_t)batch * 3 * hidden + (size_t)batch * steps * hidden) * sizeof(float);
}

static int linear_launch(adaf_handle* h, const float* x, int rows, int ldx, int in, int out_dim, const float* w,
                         const float* bias, float* out, int ldo, hipStream_t st) {
    adaf_conv_params p;
    memset(&p, 0, sizeof(p));
    p.n = rows; p.h = 1; p.w = 1; p.cin = in; p.cout = out_dim; p.kh = p.kw = 1; p.stride = 1; p.pad = 0;
    p.act = ADAF_ACT_NONE; p.ldx = ldx; p.ldo = ldo;
    ConvArgs a;
    int rc = make_conv_args(h, &p, x, w, nullptr, bias, nullptr, out, &a);
    if (rc) return rc;
    // a few rows (config 1's GRU projection: 16 x 3328 -> 3072) are one accumulator chain per block on the engine: the
    // small-batch form's chain is 3.2x shorter and bit-identical (conv_lat.hip; ADAF_LATENCY_LINEAR_ROWS, 0 = never)
    const int lat_rows = adaf_options().latency_linear_rows;
    if (rows <= lat_rows && in >= 512 && adaf_launch_conv_gemm(a, 95, h->cus, st) > 0) return ADAF_OK;
    if (adaf_launch_conv_gemm(a, 0, h->cus, st) < 0) return fail(h, ADAF_E_LAUNCH, "linear: no kernel for this shape");
    return ADAF_OK;
}

// h_t for every step: hs[b, t, :] (row stride between steps of one clip = hidden, between clips = T*hidden); with fc_w the
// per-step classifier rides along (logits_all [B*T, C], last [B, C]).
static int gru_scan(adaf_handle* h, const float* x, int ldx, int batch, int steps, int feat, int hidden,
                    const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, const float* h0, float* gi,
                    float* gh, float* hs, const float* fc_w, const float* fc_b, int classes, float* logits_all, float* last,
                    hipStream_t st) {
    int rc;
    // all input projections at once: gi[b*T+t, :] = W_ih x[b,t] + b_ih
    if ((rc = linear_launch(h, x, batch * steps, ldx, feat, 3 * hidden, w_ih, b_ih, gi, 0, st))) return rc;
    // While `st` is being captured into a HIP graph (GFV.capture_hot_path, the small-batch latency mode) the slot events below cannot be part
    // of the capture (an event recorded outside a capture cannot be waited on inside one), so a captured persistent scan would sit OUTSIDE the
    // throttle that keeps the grid barrier's blocks co-resident: two graphs replayed side by side, or a graph beside eager hot paths, could
    // starve each other into the barrier's time-out (NaN-poisoned logits).  A capture therefore takes the launch-per-step form, which has no
    // grid barrier, unless the option "gru_graph_persistent" says the caller guarantees exclusive use (HotPathGraph then checks the time-out
    // counter every few replays).
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cap);
    const bool capturing = cap != hipStreamCaptureStatusNone;
    if (h->gru_persistent && steps + 1 <= batch * 3 * hidden && (!capturing || adaf_options().gru_graph_persistent) &&
        adaf_gru_scan_persistent_ok(batch, hidden, fc_w ? classes : 0, h->scan_resident)) {
        // the whole recurrence (+ classifier) in one kernel; `gh` only lends its first steps+1 words to the grid barrier
        // a scan cut into two slices (batch > 32) takes two sets of blocks, i.e. two of the slots the resident-block budget is made of
        const AdafGruScanPlan plan = adaf_gru_scan_plan(batch, steps, (size_t)batch * 3 * hidden, h->scan_resident);
        const int need = plan.groups > h->scan_slots ? h->scan_slots : plan.groups;
        int slots[2] = {h->scan_next, (h->scan_next + 1) % h->scan_slots};
        if (!capturing) {
            h->scan_next = (h->scan_next + need) % h->scan_slots;
            for (int i = 0; i < need; ++i)
                if (h->scan_used[slots[i]]) (void)hipStreamWaitEvent(st, h->scan_done[slots[i]], 0);   // the scans that held these slots have finished
        }
        hipError_t e = adaf_launch_gru_scan_persistent(gi, w_hh, b_hh, h0, hs, reinterpret_cast<unsigned*>(gh), plan, batch, steps, fc_w, fc_b,
                                                       logits_all, last, classes, h->gru_persistent == 2, h->scan_timeouts, st);
        if (e != hipSuccess) return hip_fail(h, e, "gru scan launch");
        if (!capturing) {
            for (int i = 0; i < need; ++i) {
                (void)hipEventRecord(h->scan_done[slots[i]], st);
                h->scan_used[slots[i]] = true;
            }
        }
        return ADAF_OK;
    }
    for (int t = 0; t < steps; ++t) {
        const float* hprev = t ? hs + (size_t)(t - 1) * hidden : h0;
        const int ldprev = t ? steps * hidden : hidden;
        if (hprev) {  // gh = W_hh h_{t-1}; rows are strided views into hs
            if ((rc = linear_launch(h, hprev, batch, ldprev, hidden, 3 * hidden, w_hh, nullptr, gh, 0, st))) return rc;
        }
        adaf_launch_gru_gates(gi + (size_t)t * 3 * hidden, steps * 3 * hidden, hprev ? gh : nullptr, b_hh, hprev, ldprev,
                              hs + (size_t)t * hidden, steps * hidden, batch, hidden, st);
    }
    if (fc_w) {   // logits for every step, then the last step's rows
        if ((rc = linear_launch(h, hs, batch * steps, hidden, hidden, classes, fc_w, fc_b, logits_all, 0, st))) return rc;
        if (last) adaf_launch_copy2d(logits_all + (size_t)(steps - 1) * classes, steps * classes, last, classes, batch, classes, st);
    }
    return ADAF_OK;
}

int adaf_gru_seq_forward_f32(adaf_handle* h, const float* x, int ldx, int batch, int steps, int feat, int hidden,
                             const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, const float* h0,
                             float* hs, void* ws, size_t ws_bytes, void* stream) {
    if (!h) return ADAF_E_BADARG;
    if (batch == 0) return ADAF_OK;
    if (!x || !w_ih || !w_hh || !b_ih || !b_hh || !hs || !ws) return fail(h, ADAF_E_BADARG, "gru_seq: null pointer");
    if (batch < 0 || steps <= 0 || feat <= 0 || hidden <= 0) return fail(h, ADAF_E_BADARG, "gru_seq: non-positive extent");
    if (ldx == 0) ldx = feat;
    if (feat % 4 || hidden % 4 || ldx % 4) return fail(h, ADAF_E_LAYOUT, "gru_seq: feat, hidden, ldx must be multiples of 4");
    if (h0 && !aligned16(h0)) return fail(h, ADAF_E_LAYOUT, "gru_seq: h0 must be 16-byte aligned");
    if (ws_bytes < adaf_gru_cls_workspace_bytes(batch, steps, hidden)) return fail(h, ADAF_E_NOMEM, "gru_seq: workspace too small");
    float* gi = static_cast<float*>(ws);
    float* gh = gi + (size_t)batch * steps * 3 * hidden;
    int rc = gru_scan(h, x, ldx, batch, steps, feat, hidden, w_ih, w_hh, b_ih, b_hh, h0, gi, gh, hs, nullptr, nullptr, 0, nullptr,
                      nullptr, (hipStream_t)stream);
    if (rc) return rc;
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ADAF_OK : hip_fail(h, e, "gru_seq forward");
}

int adaf_gru_cls_forward_f32(adaf_handle* h, const float* x, int ldx, int batch, int steps, int feat, int hidden,
                             int classes, const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh,
                             const float* fc_w, const float* fc_b, float* logits_all, float* last, void* ws,
                             size_t ws_bytes, void* stream) {
    if (!h) return ADAF_E_BADARG;
    if (batch == 0) return ADAF_OK;
    if (!x || !w_ih || !w_hh || !b_ih || !b_hh || !fc_w || !fc_b || !logits_all || !last || !ws)
        return fail(h, ADAF_E_BADARG, "gru_cls: null pointer");
    if (batch < 0 || steps <= 0 || feat <= 0 || hidden <= 0 || classes <= 0) return fail(h, ADAF_E_BADARG, "gru_cls: non-positive extent");
    if (ldx == 0) ldx = feat;
    if (feat % 4 || hidden % 4 || ldx % 4) return fail(h, ADAF_E_LAYOUT, "gru_cls: feat, hidden, ldx must be multiples of 4");
    if (ws_bytes < adaf_gru_cls_workspace_bytes(batch, steps, hidden)) return fail(h, ADAF_E_NOMEM, "gru_cls: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* gi = static_cast<float*>(ws);
    float* gh = gi + (size_t)batch * steps * 3 * hidden;
    float* hs = gh + (size_t)batch * 3 * hidden;  // [B, T, H]
    int rc = gru_scan(h, x, ldx, batch, steps, feat, hidden, w_ih, w_hh, b_ih, b_hh, nullptr, gi, gh, hs, fc_w, fc_b, classes,
                      logits_all, last, st);
    if (rc) return rc;
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ADAF_OK : hip_fail(h, e, "gru_cls forward");
}

int adaf_fc_meanpool_forward_f32(adaf_handle* h, const float* feat, int batch, int steps, int feat_dim, int classes,
                                 const float* fc_w, const float* fc_b, const float* global_logit, int global_steps,
                                 float* out, void* ws, size_t ws_bytes, void* stream) {
    if (!h) return ADAF_E_BADARG;
    if (batch == 0) return ADAF_OK;
    if (!feat || !fc_w || !fc_b || !out || !ws) return fail(h, ADAF_E_BADARG, "fc_meanpool: null pointer");
    if (batch < 0 || steps <= 0 || feat_dim <= 0 || classes <= 0 || (global_logit && global_steps <= 0))
        return fail(h, ADAF_E_BADARG, "fc_meanpool: non-positive extent");
    if (feat_dim % 4) return fail(h, ADAF_E_LAYOUT, "fc_meanpool: feat_dim %% 4");
    if (ws_bytes < (size_t)batch * steps * classes * sizeof(float)) return fail(h, ADAF_E_NOMEM, "fc_meanpool: workspace too small");
    hipStream_t st = (hipStream_t)stream;
    float* logit = static_cast<float*>(ws);
    int rc = linear_launch(h, feat, batch * steps, feat_dim, feat_dim, classes, fc_w, fc_b, logit, 0, st);
    if (rc) return rc;
    adaf_launch_segment_mean(logit, batch, steps, classes, global_logit, global_steps, out, st);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ADAF_OK : hip_fail(h, e, "fc_meanpool forward");
}


int adaf_pack_dw_weight_f32(adaf_handle* h, const float* w_c133, int channels, float* w_33c, void* stream) {
    if (!h) return ADAF_E_BADARG;
    if (!w_c133 || !w_33c || channels <= 0) return fail(h, ADAF_E_BADARG, "pack_dw: bad arguments");
    adaf_launch_pack_dw_weight(w_c133, channels, w_33c, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ADAF_OK : hip_fail(h, e, "pack_dw launch");
}

int adaf_dwconv3x3_bn_act_f32(adaf_handle* h, const float* x, int n, int hh, int ww, int c, int stride,
                              const float* w_33c, const float* scale, const float* bias, int act, float* out,
                              void* stream) {
    if (!h) return ADAF_E_BADARG;
    if (!x || !w_33c || !scale || !bias || !out || n <= 0 || hh <= 0 || ww <= 0 || c <= 0) return fail(h, ADAF_E_BADARG, "dwconv: bad arguments");
    if (stride != 1 && stride != 2) return fail(h, ADAF_E_BADARG, "dwconv: stride must be 1 or 2");
    if (act < ADAF_ACT_NONE || act > ADAF_ACT_RELU6) return fail(h, ADAF_E_BADARG, "dwconv: activation");
    if (c % 4 || !aligned16(x) || !aligned16(out) || !aligned16(w_33c) || !aligned16(scale) || !aligned16(bias))
        return fail(h, ADAF_E_LAYOUT, "dwconv: c %% 4 == 0 and 16-byte alignment required");
    adaf_launch_dwconv3x3(x, n, hh, ww, c, stride, w_33c, scale, bias, act, out, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ADAF_OK : hip_fail(h, e, "dwconv launch");
}

int adaf_grid_actions_f32(adaf_handle* h, const float* logits, int rows, int n_actions, const float* table_yx,
                          int64_t* idx_out, float* action_out, void* stream) {
    if (!h) return ADAF_E_BADARG;
    if (rows == 0) return ADAF_OK;
    if (!logits || rows < 0 || n_actions <= 0 || (!action_out && !idx_out) || (action_out && !table_yx))
        return fail(h, ADAF_E_BADARG, "grid_actions: bad arguments");
    adaf_launch_grid_actions(logits, rows, n_actions, table_yx, reinterpret_cast<long long*>(idx_out), action_out, (hipStream_t)stream);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? ADAF_OK : hip_fail(h, e, "grid_actions launch");
}

}  // extern "C"
