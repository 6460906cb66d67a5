/*
 * adafocus.h -- C ABI of the MI355X-native AdaFocus offline-inference hot path.
 *
 * The reference (blackfeather-wang/AdaFocus) is 100 % Python on PyTorch and exposes no
 * FFI/plugin registry; its boundary for this path is the Python
 * surface of models/utils.py, models/gfv_net.py, models/resnet.py, ops/temporal_shift.py.
 * This header is the flat C layer that surface is re-implemented on.  Each entry point cites
 * the reference interface it replaces.  Path aliases:
 *   ACT/ = "Experiments on ActivityNet, FCVID and Mini-Kinetics/"
 *   STH/ = "Experiments on Something-Something V1&V2/"
 *
 * Conventions
 *   - every pointer named in a compute call is a DEVICE pointer owned by the caller
 *     (PyTorch tensors' data_ptr()); the library allocates only its handle and the packed
 *     weight copies made by adaf_resnet50_finalize();
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls enqueue work
 *     and return, they never synchronise the device;
 *   - all functions return ADAF_OK (0) or a negative ADAF_E_* code; the message is available
 *     from adaf_last_error(); a handle is not thread-safe (one per device per process, like the
 *     reference's one Python thread per rank);
 *   - activations are fp32 NHWC ("pixel-major"): element (n,y,x,c) at ((n*H+y)*W+x)*ld + c.
 */
#ifndef ADAFOCUS_H
#define ADAFOCUS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADAF_VERSION 303

enum {
    ADAF_OK = 0,
    ADAF_E_BADARG = -1, /* argument out of the documented domain */
    ADAF_E_LAYOUT = -2, /* unsupported layout / alignment */
    ADAF_E_ARCH = -3,   /* device is not gfx950 or no device */
    ADAF_E_LAUNCH = -4, /* HIP runtime error while enqueueing */
    ADAF_E_STATE = -5,  /* object not finalized / missing parameter */
    ADAF_E_NOMEM = -6   /* workspace too small or allocation failure */
};

enum { ADAF_LAYOUT_NCHW = 0, ADAF_LAYOUT_NHWC = 1, ADAF_LAYOUT_NHWC4 = 2 /* C=3 padded to 4 with a zero lane */ };
enum { ADAF_ACT_NONE = 0, ADAF_ACT_RELU = 1, ADAF_ACT_RELU6 = 2, ADAF_ACT_SIGMOID = 3, ADAF_ACT_SWISH = 4 /* x * sigmoid(x) */ };

typedef struct adaf_handle adaf_handle;
typedef struct adaf_resnet50 adaf_resnet50;

/* ---- lifetime ------------------------------------------------------------------------- */
int adaf_version(void);
/* Binds to HIP device `device`; fails with ADAF_E_ARCH unless it is a gfx950 part. */
int adaf_create(int device, adaf_handle** out);
int adaf_destroy(adaf_handle* h);
const char* adaf_last_error(const adaf_handle* h);
/* Number of compute units of the bound device (256 on MI355X). */
int adaf_device_cus(const adaf_handle* h);
/* GRU scans (classifier a7, policy a11) as one persistent kernel with a grid barrier per step when hidden == 1024,
 * batch <= 256 and the runtime's occupancy query says the whole grid can be co-resident.
 *   mode 0 = two launches per step (+ a GEMM for the classifier), 1 = persistent kernel (default),
 *   mode 2 = persistent kernel launched with hipLaunchCooperativeKernel (the runtime itself guarantees co-residency).
 * All forms are deterministic; they differ in summation order. */
int adaf_set_gru_persistent(adaf_handle* h, int mode);
/* The persistent scan's grid barrier is a bounded spin: a block that never sees its peers arrive (a grid that could not
 * become co-resident -- never observed; the launcher budgets the resident blocks) poisons its outputs with NaN instead of
 * hanging the device, and bumps a device counter.  This call SYNCHRONISES the device and returns the number of such
 * blocks since adaf_create: 0 means every scan so far completed normally.  Evaluation loops check it once at the end
 * (adafocus_amd/evaluate.py) so that a starved scan can never pass as a result. */
int adaf_gru_scan_timeouts(adaf_handle* h, unsigned* count_out);
/* k x k convolutions on small maps: tiles of the SAME output pixel over consecutive images, so the filter taps that only
 * multiply zero padding are skipped for the whole tile (40 % of the products of a 3x3 conv on a 3x3 map, 21 % on 6x6).
 * Bit-identical to the row-major tiles (a skipped slice contributes exact zeros).  Default on; off for A/B and tests. */
int adaf_set_conv_pos_major(adaf_handle* h, int on);
/* Process-wide tuning / A-B switches: GLOBAL state of the library (not of a handle: the kernels' launchers read them), hence no handle in the
 * signature.  The defaults are the plan every reported number is measured with; INTEGRATION.md lists them.  Keys:
 *   "conv_pool" 0|1, "mb_strip" 0|1 (strip-walking front kernels of the glancer), "mbv2_chunk", "latency_rows", "latency_linear_rows",
 *   "effnet_plan" (ADAF_EF_PLAN_* bits), "effnet_chunk", "gru_scan_slices" 1|2, "effnet_fused_blocks" (bit b = MBConv block b may use the fused
 *   expand + depthwise launch), "stem_rows" 0|1|2, "split_stage1_f32" 0|1 (see adaf_resnet50_set_math), "split_lean" 0|1 (the split tiles' lean K loop; bit-identical A/B), "tsm_lean" 0|1 (a temporally shifted conv1 on the lean K loop; bit-identical A/B), "gru_graph_persistent" 0|1 (1 = a
 *   stream capture keeps the persistent GRU scan; default 0: captured scans take the launch-per-step form, which has no grid barrier).
 * adaf_set_global_option returns ADAF_E_BADARG for an unknown key or a value out of range; adaf_get_global_option returns the current value
 * (NaN for an unknown key).  Not thread-safe against concurrent launches: set options between forwards.
 * (Round 6 removed the switches that had measured as no-gain and had no user: conv_lean, pm_fill, resize_lds_kb, mb_wave, dw3_variant,
 *  gru_barrier, and adaf_mobilenetv2_set_dtype.) */
enum {
    ADAF_EF_PLAN_WHOLE_BLOCK = 1,    /* whole-image MBConv kernels where a block is eligible (see adaf_effnet_set_fusion) */
    ADAF_EF_PLAN_TINY_DW = 2,        /* register-resident depthwise kernel for maps up to 5 x 5 */
    ADAF_EF_PLAN_STRIP_PROJECT = 4,  /* strip kernel for the narrow gated project convs (K <= 64, N <= 32) */
    ADAF_EF_PLAN_STRIP_EXPAND = 8,   /* strip kernel for the narrow-input expand convs (K <= 64) */
    ADAF_EF_PLAN_OWN_STEM = 16,      /* EfficientNet's own 3x3 / stride-2 stem kernel instead of the generic engine */
    ADAF_EF_PLAN_FUSED_EXPAND = 32,  /* fp16 storage: the expand conv computed inside the depthwise launch (no expanded map in HBM) */
    ADAF_EF_PLAN_PACKED_STEM = 64,   /* stems of up to 48 channels: k packed to 28 (no zero channel), columns 32-47 on a 16-column MFMA tile */
    ADAF_EF_PLAN_HEAD_POOL = 128,    /* fp16 storage, pooled features only: the global average pool in the head conv's epilogue (no fp32 map) */
    ADAF_EF_PLAN_PAIR_CHUNKS = 256   /* two chunks of patches side by side on two streams (a batch of >= 512 patches that fits one chunk: two halves) */
};
int adaf_set_global_option(const char* key, double value);
double adaf_get_global_option(const char* key);


/* ---- a1: patch gather -------------------------------------------------------------------
 * Replaces get_patch(images, action_sequence, patch_size) -- ACT/models/utils.py:37-51
 * (= STH/models/utils.py:44-58) and PatchSampler.sample -- ACT/models/gfv_net.py:363-374.
 *   frames      [n_frames, channels, height, width] fp32, NCHW (the loader's layout)
 *   action_yx   [n_actions, 2] fp32 in [0,1]; column 0 = row (y) fraction, column 1 = x fraction
 *   frames_per_action  frame f uses action f / frames_per_action (1 = ActivityNet per-frame
 *               coords; T = Something-Something one (y,x) per clip applied to T frames);
 *               n_actions * frames_per_action must equal n_frames
 *   coords: y0 = (int)floorf(a_y * (float)(height - patch)), x0 likewise with the SAME
 *               (height - patch) factor (utils.py:40 uses images.size(2) for both axes);
 *               bit-exact with the reference for a in [0,1]; clamped to the frame for memory
 *               safety outside that range.  Requires width >= height.
 *   out         out_layout NCHW  -> [n_frames, channels, patch, patch]  (reference layout)
 *               out_layout NHWC  -> [n_frames, patch, patch, channels]
 *               out_layout NHWC4 -> [n_frames, patch, patch, 4], channels must be 3, lane 3 = 0
 *   coords_out  optional [n_actions, 2] int32 (y0, x0) for parity checks; may be NULL
 */
int adaf_crop_gather_f32(adaf_handle* h, const float* frames, int n_frames, int channels, int height, int width,
                         const float* action_yx, int n_actions, int frames_per_action, int patch, float* out,
                         int out_layout, int32_t* coords_out, void* stream);

/* Same gather from frames that are already pixel-major (N, H, W, 4) -- the layout
 * adaf_ingest_u8_f32 emits and the glancer consumes; output (N, P, P, 4).  Coordinates as above. */
int adaf_crop_gather_nhwc4_f32(adaf_handle* h, const float* frames_nhwc4, int n_frames, int height, int width,
                               const float* action_yx, int n_actions, int frames_per_action, int patch, float* out_nhwc4,
                               int32_t* coords_out, void* stream);

/* ---- N1: crop-and-resize ---------------------------------------------------------------------
 * (y, x, size) per action -> a patch x patch tensor: the window [y0, y0+S) x [x0, x0+S) with
 * (y0, x0) = floor(action * (height - S)) -- get_patch's expression (ACT/models/utils.py:40-42) with patch_size = S --
 * resampled to patch x patch by the bilinear rule of torchvision.transforms.Resize on tensors
 * (= F.interpolate(mode='bilinear', align_corners=False)), the transform the reference CONSTRUCTS as `self.down`
 * (ACT/models/gfv_net.py:58, STH/models/gfv_net.py:69) but never calls on its evaluation path; the live crop is the
 * fixed-size slice above.  With S == patch the result is bit-identical to adaf_crop_gather_f32 (all interpolation
 * weights are exactly 0 / 1 and zero-weight taps are not read).
 *   frames      in_layout NCHW [n_frames, channels, height, width] or NHWC4 [n_frames, height, width, 4]
 *   size_px     [n_actions] int32 DEVICE array of window sizes (clamped to [1, height]) or NULL -> size_default for all
 *   out         out_layout as adaf_crop_gather_f32;  coords_out optional [n_actions, 2] int32 (y0, x0) */
int adaf_crop_resize_f32(adaf_handle* h, const float* frames, int in_layout, int n_frames, int channels, int height, int width,
                         const float* action_yx, int n_actions, int frames_per_action, const int32_t* size_px,
                         int size_default, int patch, float* out, int out_layout, int32_t* coords_out, void* stream);

/* Nearest-neighbour resize of whole frames: F.interpolate(images, (glance_size, glance_size)) with the default mode --
 * the glancer's input when glance_size != input_size (ACT/main_dist.py:331-332, STH/evaluate.py:188).
 * src = min(int(floorf(dst * (float)in / out)), in - 1) (ATen's rule); a copy, hence bit-exact.  Layouts as above. */
int adaf_resize_nearest_f32(adaf_handle* h, const float* frames, int in_layout, int n_frames, int channels, int height, int width,
                            int out_h, int out_w, float* out, int out_layout, void* stream);

/* ---- f1: frame ingest ---------------------------------------------------------------------
 * Stack + ToTorchFormatTensor + GroupNormalize -- ACT/ops/transforms.py:305-336,64-77 -- fused:
 *   clips_hwc [n_clips, height, width, frames*3] uint8 (the loader's stacked clip; frame t = channels 3t..3t+2)
 *   out_nhwc4 [n_clips*frames, height, width, 4] fp32, value = ((u8 / 255) - mean[c]) / std[c] in IEEE fp32
 *   (the reference's op order, bit-exact), lane 3 = 0.  mean3 / std3 are HOST arrays of 3 floats. */
int adaf_ingest_u8_f32(adaf_handle* h, const uint8_t* clips_hwc, int n_clips, int frames, int height, int width,
                       const float* mean3, const float* std3, float* out_nhwc4, void* stream);

/* ---- a4: fused conv + BN(eval) + residual + activation, implicit GEMM on fp32 MFMA ------
 * Replaces the nn.Conv2d -> nn.BatchNorm2d -> (+identity) -> ReLU sequences of
 * Bottleneck.forward -- ACT/models/resnet.py:94-114, the stem (:212-214), the pointwise
 * convs of InvertedResidual -- ACT/models/mobilenet.py:42-68, and nn.Linear (h = w = 1).
 * out[m, co] = act( dot(x_window[m,:], w[co,:]) * scale[co] + bias[co] + residual[m, co] )
 */
typedef struct adaf_conv_params {
    int n, h, w, cin;   /* input: n images of h x w pixels, cin channels (cin % 4 == 0) */
    int cout, kh, kw;   /* filter bank [cout][kh][kw][cin] (see adaf_pack_conv_weight_f32) */
    int stride, pad;
    int act;            /* ADAF_ACT_* */
    int tsm_segments;   /* > 0: temporal shift (STH/ops/temporal_shift.py:28-46) of the INPUT, fused into the
                           operand load; images are (clip, t) with t fastest, n % tsm_segments == 0;
                           only for kh = kw = 1, stride 1, pad 0 */
    int tsm_div;        /* fold = cin / tsm_div, must be a multiple of 4 */
    int ldx, ldo, ldr;  /* pixel strides in floats of x / out / residual; 0 = dense (cin / cout / cout) */
    int tile;           /* 0 = choose automatically; otherwise force a kernel variant (tuning / tests):
                           1..4 = 128x128, 128x64, 64x64, 64x128 block tiles with register staging,
                           21..24 = the same tiles with direct-to-LDS loads, 5 / 25..27 = larger experimental tiles,
                           31..34 / 37 = direct-to-LDS with the loads issued between the MFMA groups (the default form),
                           40 = choose automatically among the split tiles, 41..47 / 51..54 = split tiles: fp32
                           operands decomposed into three bf16 parts after the LDS read and multiplied on the bf16
                           matrix pipe with 6 (4x) or 9 (5x) products per element pair, fp32 accumulate
                           (see ADAF_MATH_F32_SPLIT_BF16)
                           95 = the small-batch form (csrc/conv_lat.hip): 32x32 block tiles of four 16x16 wave tiles on
                           v_mfma_f32_16x16x4_f32, whose accumulator chain is 3.2x shorter per k than the 32x32x2 chain of
                           every other fp32 tile and visits k in the same order -- bit-identical results; fp32 only,
                           cin % 64 == 0, no temporal shift (anything else: ADAF_E_LAUNCH) */
} adaf_conv_params;

#define ADAF_CONV_TILES 4

int adaf_conv2d_bn_act_f32(adaf_handle* h, const adaf_conv_params* p, const float* x, const float* w_ohwi,
                           const float* scale, const float* bias, const float* residual, float* out,
                           void* stream);
/* Same contract, one thread per output element, no MFMA: the on-device cross-check used by
 * the tests (never by the model path). */
int adaf_conv2d_naive_f32(adaf_handle* h, const adaf_conv_params* p, const float* x, const float* w_ohwi,
                          const float* scale, const float* bias, const float* residual, float* out,
                          void* stream);
/* OIHW (PyTorch) -> OHWI with the input-channel axis zero-padded to cin_pad (>= cin, % 4 == 0). */
int adaf_pack_conv_weight_f32(adaf_handle* h, const float* w_oihw, int cout, int cin, int kh, int kw, int cin_pad,
                              float* w_ohwi, void* stream);
/* Eval-mode BatchNorm as an affine map: scale = gamma / sqrt(var + eps), bias = beta - mean * scale
 * (nn.BatchNorm2d in eval mode, ACT/models/resnet.py:85-91). */
int adaf_fold_bn_f32(adaf_handle* h, const float* gamma, const float* beta, const float* mean, const float* var,
                     float eps, int channels, float* scale, float* bias, void* stream);

/* ---- N2: half-precision STORAGE variants (BASELINE config 5) -------------------------------------------------
 * Activations and 1x1 weights as fp16 in HBM, products on v_mfma_f32_32x32x16_f16, accumulation / BN affine /
 * activation in fp32, fp16 (or fp32) store.  The reference computes in fp32 everywhere (no autocast in its validate
 * loops) and has NO half-precision or EfficientNet implementation on a live path: these entry
 * points have no reference call site and their parity is pinned only against the fp32 goldens at fp16 tolerance.
 *   x_dtype F16: x [n,h,w,cin] and w [cout][kh][kw][cin] hold fp16 (adaf_pack_conv_weight_f16), cin %% 8 == 0
 *               (k x k filters: cin %% 64 == 0); residual, if any, fp16; out fp16 or fp32.
 *   x_dtype F32: fp32 operands as adaf_conv2d_bn_act_f32 with an fp16 store (the 3-channel stem); no residual.
 *   adaf_conv_params.tile: 0 = automatic, 81..84 = 128x128 / 128x64 / 64x64 / 64x128, 88 = 128x32. */
enum { ADAF_DTYPE_F32 = 0, ADAF_DTYPE_F16 = 1 };
int adaf_conv2d_bn_act_f16(adaf_handle* h, const adaf_conv_params* p, const void* x, int x_dtype, const void* w_ohwi,
                           const float* scale, const float* bias, const void* residual_f16, void* out, int out_dtype,
                           void* stream);
int adaf_pack_conv_weight_f16(adaf_handle* h, const float* w_oihw, int cout, int cin, int kh, int kw, int cin_pad,
                              void* w_ohwi_f16, void* stream);
/* Element-wise conversion, round-to-nearest-even: to_f16 != 0: fp32 -> fp16, else fp16 -> fp32. */
int adaf_cast_f32_f16(adaf_handle* h, const void* src, size_t count, void* dst, int to_f16, void* stream);
/* Depthwise 3x3 with fp16 activations (taps, BN affine and the sum in fp32). */
int adaf_dwconv3x3_bn_act_f16(adaf_handle* h, const void* x_f16, int n, int hh, int ww, int c, int stride, const float* w_33c,
                              const float* scale, const float* bias, int act, void* out_f16, void* stream);

/* ---- pooling ---------------------------------------------------------------------------
 * nn.MaxPool2d(3, 2, 1) -- ACT/models/resnet.py:141,215; nn.AdaptiveAvgPool2d((1,1)) -- :150,222. */
int adaf_maxpool3x3s2_f32(adaf_handle* h, const float* x, int n, int hh, int ww, int c, float* out, void* stream);
int adaf_global_avgpool_f32(adaf_handle* h, const float* x, int n, int hw, int c, float* out, int ldo, void* stream);

/* ---- a6: stand-alone temporal shift ----------------------------------------------------
 * TemporalShift.shift(x, n_segment, fold_div) -- STH/ops/temporal_shift.py:28-46.
 * x, out: [n_clips * n_segment, c, hw] (layout NCHW) or [n_clips * n_segment, hw, c] (NHWC). */
int adaf_temporal_shift_f32(adaf_handle* h, const float* x, int nt, int c, int hw, int n_segment, int fold_div,
                            int layout, float* out, void* stream);

/* ---- a4/a5: ResNet-50 trunk as one object ----------------------------------------------
 * Replaces ResNet.get_featmap(x, pooled=True) -- ACT/models/resnet.py:211-225 -- and
 * TSN.forward(input, no_reshape=True) -- STH/models/tsn.py:215-241 (TSM on every Bottleneck
 * conv1, STH/ops/temporal_shift.py:123-140).  Parameter names are torchvision's
 * ("conv1.weight", "bn1.running_var", "layer3.4.conv2.weight", "layer2.0.downsample.1.bias").
 */
int adaf_resnet50_create(adaf_handle* h, adaf_resnet50** out);
int adaf_resnet50_destroy(adaf_resnet50* net);
/* Registers a device pointer to a parameter / buffer in PyTorch layout; the data is read by
 * adaf_resnet50_finalize() and not needed afterwards. */
int adaf_resnet50_set_param(adaf_resnet50* net, const char* name, const float* dev_ptr, size_t numel);
/* Packs weights (OIHW -> OHWI, stem cin 3 -> 4), folds BN, on `stream`; synchronises that stream. */
int adaf_resnet50_finalize(adaf_resnet50* net, void* stream);
size_t adaf_resnet50_workspace_bytes(const adaf_resnet50* net, int n, int patch);
/* patches_nhwc4 [n, patch, patch, 4] (adaf_crop_gather_f32 with ADAF_LAYOUT_NHWC4);
 * feat[i * ldfeat + c], c < 2048: the pooled feature (get_featmap(..., pooled=True).view(n,-1)).
 * tsm_segments = 0 for the ActivityNet model. */
int adaf_resnet50_forward(adaf_resnet50* net, const float* patches_nhwc4, int n, int patch, int tsm_segments,
                          int tsm_div, float* feat, int ldfeat, void* ws, size_t ws_bytes, void* stream);
/* The same forward with the patch gather folded in: the trunk's patches are the P x P windows of resident frames at
 * floor(action * (H - P)) -- get_patch(images, action_sequence, patch_size), ACT/models/utils.py:37-51 (= STH/models/utils.py:44-58),
 * adaf_crop_gather_f32's coordinate arithmetic -- and the stem gathers them itself (csrc/stem.hip stem7x7_pool_rows_kernel): no gather
 * launch and no patch tensor between Focuser.forward's two calls (ACT/models/gfv_net.py:325-331).  frames: [n_frames, 3, H, W]
 * (ADAF_LAYOUT_NCHW, the loader's layout) or [n_frames, H, W, 4] (ADAF_LAYOUT_NHWC4, adaf_ingest_u8's output), square.  action_yx:
 * [n_actions, 2] fp32; n_actions = k * n_frames / frames_per_action: action set j (k > 1: the Something-Something reward baseline rides
 * in the same pass) cuts patch j * n_frames + f from frame f with action j * n_frames / fpa + f / fpa.  feat: [k * n_frames, 2048].
 * Where the gathering stem does not apply (patch sizes other than 64 / 96 / 128 / 144, fewer images than CUs, fusion off) the gather
 * runs into a workspace slab first: the same values either way (tests/test_hip_parity_r5.py).  Workspace: adaf_resnet50_workspace_bytes
 * for k * n_frames patches. */
int adaf_resnet50_forward_frames(adaf_resnet50* net, const float* frames, int frames_layout, int n_frames, int height, int width,
                                 const float* action_yx, int n_actions, int frames_per_action, int patch, int tsm_segments, int tsm_div,
                                 float* feat, int ldfeat, void* ws, size_t ws_bytes, void* stream);
/* ResNet.get_featmap(x, pooled=False) (ACT/models/resnet.py:211-225, the branch `return x` before avgpool): the same pass, with the last
 * block's map written to featmap_nhwc [n][s][s][2048] (s = adaf_resnet50_map_size(patch): 3 at 96^2, 4 at 128^2) and the pooled feature to
 * `feat` as in adaf_resnet50_forward (the pool runs as its own launch here: same values). */
int adaf_resnet50_map_size(int patch);
int adaf_resnet50_forward_map(adaf_resnet50* net, const float* patches_nhwc4, int n, int patch, int tsm_segments, int tsm_div,
                              float* featmap_nhwc, float* feat, int ldfeat, void* ws, size_t ws_bytes, void* stream);
/* Same as forward, but brackets every launch with HIP events on `stream` and reports per-launch
 * milliseconds, algorithmic FLOPs (2*MAC, 0 for non-conv launches) and algorithmic bytes.
 * Arrays must hold adaf_resnet50_launch_count() entries.  Synchronises the stream. */
int adaf_resnet50_launch_count(const adaf_resnet50* net);
int adaf_resnet50_forward_profiled(adaf_resnet50* net, const float* patches_nhwc4, int n, int patch,
                                   int tsm_segments, int tsm_div, float* feat, int ldfeat, void* ws, size_t ws_bytes,
                                   void* stream, float* launch_ms, double* launch_flops, double* launch_bytes,
                                   int* launch_tile);
/* Kernel-variant override table for tuning: tile[i] as in adaf_conv_params.tile for conv launch i (0 = auto). */
int adaf_resnet50_set_tiles(adaf_resnet50* net, const int* tile, int count);
/* Layer fusion inside the trunk (default on; off = one launch per layer, for A/B and the bit-identity tests):
 *   - stage 1 (64 planes): conv2 3x3 -> conv3 1x1 + identity + ReLU -> the NEXT block's conv1 1x1 in one launch per
 *     128-pixel tile (csrc/conv_gemm.hip conv_fused_tail_kernel); a next conv1 that carries a temporal shift rides along when whole clips fill
 *     the position-major 128-image tiles (clip lengths dividing 128; round 6), else it stays a launch of its own;
 *   - stem conv 7x7/2 + BN + ReLU + max-pool 3x3/2 in one launch (csrc/stem.hip) at the patch sizes where that is the
 *     faster plan (on = 2: at every size, for tests); the stage-1 launches likewise only from ~1.5 row tiles of 128 pixels per CU
 *     upwards (below that the three separate launches are faster: small batches; on = 2: always);
 *   - the global average pool (ACT/models/resnet.py:222-223) in the epilogue of the last conv3 when whole images fill its
 *     128-row tiles (3x3 / 4x4 / 5x5 final maps): the 2048-channel map is never written (conv_gemm.hip conv_epilogue_pool).
 * Results are bit-identical to the unfused launches (same k order in every product).  ADAF_MATH_F32, and the fp32-pipe layers of the hybrid ADAF_MATH_F32_SPLIT_BF16 plan (see adaf_resnet50_set_math). */
int adaf_resnet50_set_fusion(adaf_resnet50* net, int on);
/* Small batches (BASELINE config 1 is B = 2, T = 8: 16 patches).  A conv launch whose GEMM has at most `rows` output pixels
 * (default 1536, env ADAF_LATENCY_ROWS; 0 = never) runs on the small-batch form (tile id 95 of adaf_conv_params.tile) instead
 * of the engine's batched tiles: at 16 patches stages 3 and 4 are 576 / 144 rows, a few dozen blocks whose duration is one
 * accumulator chain.  Bit-identical to the batched plan: a clip's logits do not depend on the size of the batch it came in
 * (tests/test_hip_parity_r3.py).  ADAF_MATH_F32 without temporal shift only.  No reference counterpart (the reference's
 * published latency is a CPU bs = 1 figure). */
int adaf_resnet50_set_latency_rows(adaf_resnet50* net, int rows);
/* Where the temporal shift sits (make_temporal_shift(net, n_segment, n_div, place), STH/ops/temporal_shift.py:99-142):
 *   ADAF_SHIFT_BLOCKRES (default; every shipped configuration, the STH/conf yaml files: `shift_place: blockres`): TemporalShift wraps the
 *                       conv1 of every Bottleneck (:123-140) -- fused into that conv's operand load, no shifted tensor exists;
 *   ADAF_SHIFT_BLOCK    TemporalShift wraps the WHOLE Bottleneck (:104-121): conv1, the downsample conv and the identity all see
 *                       the shifted block input.  The shifted map is materialised once per block (adaf_temporal_shift_f32's
 *                       kernel) in a sixth workspace slab -- adaf_resnet50_workspace_bytes(net, ...) grows accordingly, so set
 *                       the placement before sizing the workspace.
 * Only read when a forward is called with tsm_segments > 0. */
enum { ADAF_SHIFT_BLOCKRES = 0, ADAF_SHIFT_BLOCK = 1 };
int adaf_resnet50_set_shift_place(adaf_resnet50* net, int place);
/* Arithmetic of the trunk's convolutions (no reference counterpart; the reference is plain fp32).
 *   ADAF_MATH_F32            (default) v_mfma_f32_32x32x2_f32: an exact fp32 FMA chain per output.
 *   ADAF_MATH_F32_SPLIT_BF16 (opt-in)  every fp32 operand x is decomposed EXACTLY into bf16 parts h + m + l (round-to-
 *                            nearest: h = bf16(x), m = bf16(x-h), l = x-h-m) and x*y is accumulated in fp32 from the
 *                            six bf16 products of magnitude >= 2^-24 |xy| on v_mfma_f32_32x32x16_bf16.  Inputs, outputs,
 *                            weights and the accumulator stay fp32; error against fp64: per convolution not larger than
 *                            the default's, whole trunk 5.9e-7 vs 3.8e-7 rms.  The plan is HYBRID (round 5): the stem and
 *                            convolutions 1-11 (layer1.* and layer2.0.conv1 -- HBM-bound whatever the matrix pipe) stay on the fp32
 *                            pipe, the rest runs the six-product form; the global option "split_stage1_f32" = 0 puts every conv but
 *                            the stem on split tiles (NOT bit-identical to the default hybrid plan, both fp32-accurate; results of
 *                            split_bf16 callers changed by <= 2e-5 when the hybrid plan became the default).  adaf_resnet50_set_fusion
 *                            applies to the fp32-pipe launches of either mode. */
enum { ADAF_MATH_F32 = 0, ADAF_MATH_F32_SPLIT_BF16 = 1 };
int adaf_resnet50_set_math(adaf_resnet50* net, int mode);

/* ---- a10: MobileNetV2 building blocks and the glancer as one object ---------------------
 * Depthwise 3x3 (pad 1) + BN(eval) + ReLU6 -- the middle conv of InvertedResidual
 * (ACT/models/mobilenet.py:52-62, STH/models/mobilenetv2.py:36-63).  x [n,hh,ww,c] NHWC, c % 4 == 0,
 * w_33c [3][3][c] (adaf_pack_dw_weight_f32 from PyTorch's [c,1,3,3]). */
int adaf_pack_dw_weight_f32(adaf_handle* h, const float* w_c133, int channels, float* w_33c, void* stream);
int adaf_dwconv3x3_bn_act_f32(adaf_handle* h, const float* x, int n, int hh, int ww, int c, int stride,
                              const float* w_33c, const float* scale, const float* bias, int act, float* out,
                              void* stream);
/* MobileNetV2.features + mean -- ACT/models/mobilenet.py:146-148 (get_featmap), STH/models/mobilenetv2.py:116-121.
 * Parameter names are layout-neutral ("stem", "b1".."b17" with ".expand/.dw/.project", "head"; each with
 * ".weight", ".bn.weight", ".bn.bias", ".bn.running_mean", ".bn.running_var"); the Python mirror maps both
 * reference key layouts onto them.  tsm_segments > 0: temporal shift in front of the expand conv of every
 * residual block (STH/models/gfv_net.py:238-241).
 *   frames_nhwc4 [n, size, size, 4]; featmap [n, size/32, size/32, 1280] NHWC; featvec [n, 1280] (row stride ldvec)
 *   or NULL. */
typedef struct adaf_mobilenetv2 adaf_mobilenetv2;
int adaf_mobilenetv2_create(adaf_handle* h, adaf_mobilenetv2** out);
int adaf_mobilenetv2_destroy(adaf_mobilenetv2* net);
int adaf_mobilenetv2_set_param(adaf_mobilenetv2* net, const char* name, const float* dev_ptr, size_t numel);
int adaf_mobilenetv2_finalize(adaf_mobilenetv2* net, void* stream);
size_t adaf_mobilenetv2_workspace_bytes(const adaf_mobilenetv2* net, int n, int size, int tsm_segments);
int adaf_mobilenetv2_forward(adaf_mobilenetv2* net, const float* frames_nhwc4, int n, int size, int tsm_segments,
                             int tsm_div, float* featmap, float* featvec, int ldvec, void* ws, size_t ws_bytes,
                             void* stream);
/* Bits of `on` (default 1): bit 0 -- fused kernels: stem + block 1 in one launch; expand 1x1 -> depthwise 3x3 in one kernel
 * (the 6x-expanded map stays on chip) for the blocks whose shape allows it (cin % 8 == 0, cin <= 32, map >= 28^2: b2..b7 at
 * 224^2); and, unless bit 3 (value 8) is set, the WHOLE stride-1 block (expand -> depthwise -> project + identity) in one
 * kernel where cout <= 32 and hidden <= 192 (b3, b5, b6).  bit 2 (value 4): one frame chunk at a time instead of two side by
 * side.  0 = the three-launch form.  Every combination is bit-identical (tests, A/B).
 * Round 6 (global option "mb_strip", default 1; frames whose stem output side is a multiple of 14, e.g. 224^2): the fused kernels are STRIP-WALKING
 * forms (csrc/mbstrip.hip) -- stem + block 1, the whole blocks b2 .. b6 INCLUDING the stride-2 ones with their project conv, and expand -> depthwise
 * of the 14 x 14 blocks b8 .. b13 (bit 3 set: b2 .. b6 and b8 .. b13 fall back to the wave-private kernels / separate launches).  Same bits as every
 * other combination. */
int adaf_mobilenetv2_set_fusion(adaf_mobilenetv2* net, int on);

/* ---- N2 / BASELINE config 5: EfficientNet (MBConv with squeeze-and-excite) as the local CNN --------------------------
 * PARITY UNPINNED: the reference has no EfficientNet on a live path.  It names the third-party package
 * `efficientnet_pytorch` in dead AR-Net code (STH/ops/models_ada.py:6,69-75; not vendored, no version pin) and lists
 * "efficientnet-b3" with feature dimension 1536 and the prior (1.80 GFLOPs, 12 M parameters) in
 * STH/ops/net_flops_table.py:17,29.  These entry points implement the PUBLISHED algorithm of that package
 * (efficientnet_pytorch 0.7.x: model.py MBConvBlock.forward / EfficientNet.extract_features; utils.py round_filters,
 * round_repeats, Conv2dStaticSamePadding, MemoryEfficientSwish; BN eps 1e-3, se_ratio 0.25), restated on the CPU by
 * oracle/ref_effnet.py.
 *
 * Building blocks (also usable on their own; x_dtype / dtype = ADAF_DTYPE_F32 | ADAF_DTYPE_F16 storage, math in fp32):
 *   adaf_dwconv_same_bn_act   depthwise k x k (k = 3 | 5, stride 1 | 2) with TensorFlow-SAME padding
 *                             (pad_before = total / 2, the rest after -- Conv2dStaticSamePadding for an input of the size at
 *                             hand) + BN affine + activation; x, out [n,h,w,c] / [n,ceil(h/s),ceil(w/s),c] NHWC,
 *                             w_kkc [k*k][c] (adaf_pack_dw_weight_kxk_f32 from PyTorch's [c,1,k,k]); optionally the
 *                             squeeze pool_mean [n,c] = mean over the output pixels (fp32, summed in a fixed order;
 *                             needs ws of adaf_dwconv_same_workspace_bytes)
 *   adaf_se_gate_f32          gate[n,c] = sigmoid(W_e swish(W_r pool_mean[n] + b_r) + b_e); W_r [squeezed,c], W_e [c,squeezed]
 *                             (_se_reduce / _se_expand weights in PyTorch layout)
 *   adaf_conv1x1_gated_bn     out[m,co] = (sum_k x[m,k] gate[m / hw, k] w[co,k]) * scale[co] + bias[co] (+ residual[m,co]):
 *                             `sigmoid(x_squeezed) * x` followed by _project_conv + _bn2 (+ the identity skip); w [cout,cin]
 *                             in x's element type; gate may be NULL (plain 1x1 conv + BN) */
int adaf_pack_dw_weight_kxk_f32(adaf_handle* h, const float* w_c1kk, int channels, int k, float* w_kkc, void* stream);
size_t adaf_dwconv_same_workspace_bytes(int n, int hh, int ww, int c, int k, int stride, int dtype);
int adaf_dwconv_same_bn_act(adaf_handle* h, const void* x, int dtype, int n, int hh, int ww, int c, int k, int stride, const float* w_kkc,
                            const float* scale, const float* bias, int act, void* out, float* pool_mean, void* ws, size_t ws_bytes,
                            void* stream);
int adaf_se_gate_f32(adaf_handle* h, const float* pool_mean, int n, int c, const float* w_reduce, const float* b_reduce, int squeezed,
                     const float* w_expand, const float* b_expand, float* gate, void* stream);
int adaf_conv1x1_gated_bn(adaf_handle* h, const void* x, int dtype, int n_images, int hw, int cin, const float* gate, const void* w,
                          int cout, const float* scale, const float* bias, const void* residual, void* out, void* stream);
/* The network.  width / depth coefficients of utils.py efficientnet_params(): B0 (1.0, 1.0) ... B3 (1.2, 1.4) ... B7 (2.0, 3.1).
 * Parameter names are layout-neutral: "stem", "b<i>.expand" (absent when the block's expand ratio is 1), "b<i>.dw",
 * "b<i>.project", "head", each with ".weight", ".bn.weight", ".bn.bias", ".bn.running_mean", ".bn.running_var";
 * "b<i>.se_reduce" / "b<i>.se_expand" with ".weight", ".bias"; i counts MBConv blocks from 0 like `_blocks.<i>`.  The Python
 * mirror (adafocus_amd/efficientnet.py) maps efficientnet_pytorch's state-dict keys onto them.
 *   adaf_effnet_block_info: info8 = {kernel, stride, expand ratio, cin, cout, hidden, squeezed, stem channels}
 *   adaf_effnet_set_dtype:  ADAF_DTYPE_F16 = activations and 1x1 filters as fp16 in HBM (BASELINE config 5's "fp16");
 *                           takes effect at the next finalize()
 *   forward: frames_nhwc4 [n,size,size,4] fp32 (the gather's output); pad_size = the image size the SAME padding is
 *            computed for (EfficientNet.from_name(..., image_size=pad_size)); 0 = size itself;
 *            featmap [n,s,s,feature_dim] fp32 NHWC (extract_features) or NULL; featvec [n, ldvec] fp32 = _avg_pooling +
 *            flatten, or NULL; upto_block >= 0 (tests): stop after that many MBConv blocks and copy the block output
 *            [n,h,w,c] in the storage dtype to block_out instead (featmap / featvec untouched). */
typedef struct adaf_effnet adaf_effnet;
int adaf_effnet_create(adaf_handle* h, float width_coefficient, float depth_coefficient, adaf_effnet** out);
int adaf_effnet_destroy(adaf_effnet* net);
int adaf_effnet_feature_dim(const adaf_effnet* net);
int adaf_effnet_block_count(const adaf_effnet* net);
int adaf_effnet_block_info(const adaf_effnet* net, int block, int* info8);
int adaf_effnet_set_dtype(adaf_effnet* net, int dtype);
/* on (DEFAULT): fp16 storage only -- the stride-1 MBConv blocks whose map is at most 9 x 9 and the stride-2 block that takes a 9 x 9
 * map to 5 x 5 (blocks 9-17, 18 and 19-24 of B3 at 144^2: 16 blocks; block 25 (hid 2304 > 2048) keeps the four-launch plan;
 * patches) run as ONE launch per block: a workgroup owns whole images, the 6x-expanded map goes from the MFMA accumulators
 * straight into the depthwise taps, the depthwise output and the squeeze-and-excite live in LDS, the block reads its input and
 * writes its output (csrc/mbconv_whole.hip).  off = the four-launch plan (expand, depthwise, SE gate, gated project).  Same
 * arithmetic and the same fp16 roundings for every stored value; the squeeze adds its pixels in a different order (fp32
 * rounding level, which can move a gated value across an fp16 rounding boundary). */
int adaf_effnet_set_fusion(adaf_effnet* net, int on);
/* how many MBConv blocks of a forward at this input size take the one-launch form (0 in fp32 storage or with fusion off) */
int adaf_effnet_whole_blocks(const adaf_effnet* net, int size, int pad_size);
/* how many MBConv blocks of a forward at this input size compute their expand conv inside the depthwise launch (fp16 storage,
 * ADAF_EF_PLAN_FUSED_EXPAND: narrow-input blocks -- cin <= 64 -- that are not whole-image blocks; blocks 2-8 of B3 at 144^2): the
 * expanded map of those blocks never exists in HBM.  Same arithmetic per stored value as the two launches (the two MFMA shapes
 * give the same bits); the tile plan, and with it the order of the squeeze partial sums, is its own. */
int adaf_effnet_fused_expand_blocks(const adaf_effnet* net, int size, int pad_size);
/* ADAF_EF_PLAN_PAIR_CHUNKS: adaf_effnet_forward sends two chunks of patches through the network side by side -- a batch of >= 512 patches that
 * fits one chunk ("effnet_chunk") is cut into two halves -- the second on a stream the library owns (one per caller stream, forked from and
 * joined to it by events inside the call: the call stays asynchronous and stream-ordered for the caller; adaf_effnet_workspace_bytes covers
 * both chunks).  A patch's results do not depend on the chunk it travels in. */
/* ADAF_EF_PLAN_HEAD_POOL (fp16 storage, `features` requested without `featmap`, head maps that fill a 128-row tile to >= 90 %: 3 x 3, 4 x 4,
 * 5 x 5): the head conv 1x1 + BN + swish does not write its fp32 map -- the global average pool runs in the conv launch's epilogue
 * (csrc/conv_gemm.hip adaf_launch_conv_pool16), adding a map's pixels in pixel order and dividing, as the pool launch does: the same bits. */
int adaf_effnet_set_param(adaf_effnet* net, const char* name, const float* dev_ptr, size_t numel);
int adaf_effnet_finalize(adaf_effnet* net, void* stream);
size_t adaf_effnet_workspace_bytes(const adaf_effnet* net, int n, int size, int pad_size);
int adaf_effnet_forward(adaf_effnet* net, const float* frames_nhwc4, int n, int size, int pad_size, int upto_block, void* block_out,
                        float* featmap, float* featvec, int ldvec, void* ws, size_t ws_bytes, void* stream);

/* ---- a11: policy head -------------------------------------------------------------------
 * idx = argmax_a logits[row, a] (first maximum), action = table_yx[idx] -- the eval branch of
 * ActorCritic.act (ACT/models/ppo.py:94: action_probs.max(1)[1]; softmax is monotone) followed by
 * Focuser._get_standard_action (ACT/models/gfv_net.py:345-347).  idx_out (int64) may be NULL; action_out and
 * table_yx may both be NULL when only the index is wanted (the one-step act() of the reference signature). */
int adaf_grid_actions_f32(adaf_handle* h, const float* logits, int rows, int n_actions, const float* table_yx,
                          int64_t* idx_out, float* action_out, void* stream);
/* nn.GRU (batch_first) over a whole sequence: hs[b, t, :] for every step -- the recurrent part of the policy
 * (ACT/models/ppo.py:78-79 applied T times; steps = 1 with h0 = the previous call's state is one call of
 * ActorCritic.act with restart_batch=False, ppo.py:70-79) and of the classifier.  h0 [batch, hidden] or NULL (= zeros,
 * restart_batch=True).  Workspace as adaf_gru_cls_workspace_bytes. */
int adaf_gru_seq_forward_f32(adaf_handle* h, const float* x, int ldx, int batch, int steps, int feat, int hidden,
                             const float* w_ih, const float* w_hh, const float* b_ih, const float* b_hh, const float* h0,
                             float* hs, void* ws, size_t ws_bytes, void* stream);

/* ---- a7: GRU classifier ----------------------------------------------------------------
 * RecurrentClassifier.forward -- ACT/models/gfv_net.py:427-435 (nn.GRU batch_first, gate order
 * r,z,n; h0 = 0; dropout = identity in eval; nn.Linear on every step).
 *   x [B, T, F] with row stride ldx (0 = F); logits_all [B*T, C]; last [B, C]
 * Two launches after the input-projection GEMM's: a memset of the barrier words and ONE persistent kernel that runs the
 * recurrence, the per-step nn.Linear (its rows ride in spare MFMA columns of the recurrent product) and the last-step
 * copy (csrc/gru_scan.hip). */
size_t adaf_gru_cls_workspace_bytes(int batch, int steps, int hidden);
int adaf_gru_cls_forward_f32(adaf_handle* h, const float* x, int ldx, int batch, int steps, int feat, int hidden,
                             int classes, const float* w_ih, const float* w_hh, const float* b_ih,
                             const float* b_hh, const float* fc_w, const float* fc_b, float* logits_all,
                             float* last, void* ws, size_t ws_bytes, void* stream);

/* ---- a8: linear classifier + temporal mean ---------------------------------------------
 * nn.Linear + ConsensusModule('avg') (+ glancer mean logits) -- STH/models/gfv_net.py:164-174,
 * STH/ops/basic_ops.py:17-26.  feat [B*T, F]; global_logit [B, Tg, C] or NULL; out [B, C];
 * ws holds B*T*C floats. */
int adaf_fc_meanpool_forward_f32(adaf_handle* h, const float* feat, int batch, int steps, int feat_dim, int classes,
                                 const float* fc_w, const float* fc_b, const float* global_logit, int global_steps,
                                 float* out, void* ws, size_t ws_bytes, void* stream);

/* ---- glue ------------------------------------------------------------------------------
 * dst[r*ldd + c] = src[r*lds + c]: torch.cat([global_feat, local_feat], dim=1) of
 * ACT/models/gfv_net.py:121 done as a strided copy into the GRU input. */
int adaf_copy2d_f32(adaf_handle* h, const float* src, int lds, float* dst, int ldd, int rows, int cols, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ADAFOCUS_H */
