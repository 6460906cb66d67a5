"""ctypes binding of include/adafocus.h.

PyTorch is plumbing here: it owns device memory and streams; every compute call goes through
the C ABI with raw ``data_ptr()`` values and the current HIP stream.  There is NO fallback: if
``libadafocus_hip.so`` is missing or the device is not a gfx950 part, calls raise.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (ADAF_LIB points at another build of the same library: A/B runs of two kernel versions on one box)
LIB_PATH = os.environ.get("ADAF_LIB") or os.path.join(_HERE, "csrc", "libadafocus_hip.so")

LAYOUT_NCHW, LAYOUT_NHWC, LAYOUT_NHWC4 = 0, 1, 2
ACT_NONE, ACT_RELU, ACT_RELU6, ACT_SIGMOID, ACT_SWISH = 0, 1, 2, 3, 4
MATH_F32, MATH_F32_SPLIT_BF16 = 0, 1
DTYPE_F32, DTYPE_F16 = 0, 1
CONV_TILES = 4

# every symbol include/adafocus.h declares (tests check the library exports all of them)
SYMBOLS = (
    "adaf_version", "adaf_create", "adaf_destroy", "adaf_last_error", "adaf_device_cus", "adaf_set_gru_persistent", "adaf_set_conv_pos_major", "adaf_gru_scan_timeouts", "adaf_set_global_option", "adaf_get_global_option",
    "adaf_crop_gather_f32", "adaf_conv2d_bn_act_f32", "adaf_conv2d_naive_f32", "adaf_pack_conv_weight_f32",
    "adaf_fold_bn_f32", "adaf_maxpool3x3s2_f32", "adaf_global_avgpool_f32", "adaf_temporal_shift_f32",
    "adaf_resnet50_create", "adaf_resnet50_destroy", "adaf_resnet50_set_param", "adaf_resnet50_finalize",
    "adaf_resnet50_workspace_bytes", "adaf_resnet50_forward", "adaf_resnet50_map_size", "adaf_resnet50_forward_map", "adaf_resnet50_launch_count",
    "adaf_resnet50_forward_profiled", "adaf_resnet50_set_tiles", "adaf_resnet50_set_math", "adaf_resnet50_set_fusion", "adaf_resnet50_set_latency_rows", "adaf_resnet50_set_shift_place", "adaf_resnet50_forward_frames", "adaf_gru_cls_workspace_bytes",
    "adaf_gru_cls_forward_f32", "adaf_fc_meanpool_forward_f32", "adaf_copy2d_f32",
    "adaf_pack_dw_weight_f32", "adaf_dwconv3x3_bn_act_f32", "adaf_mobilenetv2_create", "adaf_mobilenetv2_destroy",
    "adaf_mobilenetv2_set_param", "adaf_mobilenetv2_finalize", "adaf_mobilenetv2_workspace_bytes",
    "adaf_mobilenetv2_forward", "adaf_mobilenetv2_set_fusion", "adaf_grid_actions_f32", "adaf_gru_seq_forward_f32",
    "adaf_crop_gather_nhwc4_f32", "adaf_ingest_u8_f32", "adaf_crop_resize_f32", "adaf_resize_nearest_f32",
    "adaf_conv2d_bn_act_f16", "adaf_pack_conv_weight_f16", "adaf_cast_f32_f16", "adaf_dwconv3x3_bn_act_f16",
    "adaf_pack_dw_weight_kxk_f32", "adaf_dwconv_same_workspace_bytes", "adaf_dwconv_same_bn_act", "adaf_se_gate_f32", "adaf_conv1x1_gated_bn",
    "adaf_effnet_create", "adaf_effnet_destroy", "adaf_effnet_feature_dim", "adaf_effnet_block_count", "adaf_effnet_block_info",
    "adaf_effnet_set_dtype", "adaf_effnet_set_fusion", "adaf_effnet_whole_blocks", "adaf_effnet_fused_expand_blocks", "adaf_effnet_set_param", "adaf_effnet_finalize", "adaf_effnet_workspace_bytes", "adaf_effnet_forward",
)


class ConvParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("n", "h", "w", "cin", "cout", "kh", "kw", "stride", "pad", "act",
                                       "tsm_segments", "tsm_div", "ldx", "ldo", "ldr", "tile")]


class AdafError(RuntimeError):
    pass


_lib = None


def load_library():
    """dlopen the in-tree shared library; raises (loudly) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AdafError("HIP extension not built: %s is missing (run `python -c 'import __graft_entry__ as g; "
                        "g.build()'` or `make -C adafocus_amd/csrc`)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, ip, fp = C.c_void_p, C.c_int, C.c_float
    lib.adaf_last_error.restype = C.c_char_p
    lib.adaf_last_error.argtypes = [vp]
    lib.adaf_create.argtypes = [ip, C.POINTER(vp)]
    lib.adaf_destroy.argtypes = [vp]
    lib.adaf_device_cus.argtypes = [vp]
    lib.adaf_set_gru_persistent.argtypes = [vp, ip]
    lib.adaf_set_conv_pos_major.argtypes = [vp, ip]
    lib.adaf_gru_scan_timeouts.argtypes = [vp, C.POINTER(C.c_uint)]
    lib.adaf_set_global_option.argtypes = [C.c_char_p, C.c_double]
    lib.adaf_get_global_option.argtypes = [C.c_char_p]
    lib.adaf_get_global_option.restype = C.c_double
    lib.adaf_crop_gather_f32.argtypes = [vp, vp, ip, ip, ip, ip, vp, ip, ip, ip, vp, ip, vp, vp]
    for name in ("adaf_conv2d_bn_act_f32", "adaf_conv2d_naive_f32"):
        getattr(lib, name).argtypes = [vp, C.POINTER(ConvParams), vp, vp, vp, vp, vp, vp, vp]
    lib.adaf_pack_conv_weight_f32.argtypes = [vp, vp, ip, ip, ip, ip, ip, vp, vp]
    lib.adaf_fold_bn_f32.argtypes = [vp, vp, vp, vp, vp, fp, ip, vp, vp, vp]
    lib.adaf_maxpool3x3s2_f32.argtypes = [vp, vp, ip, ip, ip, ip, vp, vp]
    lib.adaf_global_avgpool_f32.argtypes = [vp, vp, ip, ip, ip, vp, ip, vp]
    lib.adaf_temporal_shift_f32.argtypes = [vp, vp, ip, ip, ip, ip, ip, ip, vp, vp]
    lib.adaf_resnet50_create.argtypes = [vp, C.POINTER(vp)]
    lib.adaf_resnet50_destroy.argtypes = [vp]
    lib.adaf_resnet50_set_param.argtypes = [vp, C.c_char_p, vp, C.c_size_t]
    lib.adaf_resnet50_finalize.argtypes = [vp, vp]
    lib.adaf_resnet50_workspace_bytes.restype = C.c_size_t
    lib.adaf_resnet50_workspace_bytes.argtypes = [vp, ip, ip]
    lib.adaf_resnet50_forward.argtypes = [vp, vp, ip, ip, ip, ip, vp, ip, vp, C.c_size_t, vp]
    lib.adaf_resnet50_forward_frames.argtypes = [vp, vp, ip, ip, ip, ip, vp, ip, ip, ip, ip, ip, vp, ip, vp, C.c_size_t, vp]
    lib.adaf_resnet50_map_size.argtypes = [ip]
    lib.adaf_resnet50_forward_map.argtypes = [vp, vp, ip, ip, ip, ip, vp, vp, ip, vp, C.c_size_t, vp]
    lib.adaf_resnet50_launch_count.argtypes = [vp]
    lib.adaf_resnet50_forward_profiled.argtypes = [vp, vp, ip, ip, ip, ip, vp, ip, vp, C.c_size_t, vp, vp, vp, vp, vp]
    lib.adaf_resnet50_set_tiles.argtypes = [vp, vp, ip]
    lib.adaf_resnet50_set_math.argtypes = [vp, ip]
    lib.adaf_resnet50_set_fusion.argtypes = [vp, ip]
    lib.adaf_resnet50_set_latency_rows.argtypes = [vp, ip]
    lib.adaf_resnet50_set_shift_place.argtypes = [vp, ip]
    lib.adaf_gru_cls_workspace_bytes.restype = C.c_size_t
    lib.adaf_gru_cls_workspace_bytes.argtypes = [ip, ip, ip]
    lib.adaf_gru_cls_forward_f32.argtypes = [vp, vp, ip, ip, ip, ip, ip, ip, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                             C.c_size_t, vp]
    lib.adaf_fc_meanpool_forward_f32.argtypes = [vp, vp, ip, ip, ip, ip, vp, vp, vp, ip, vp, vp, C.c_size_t, vp]
    lib.adaf_copy2d_f32.argtypes = [vp, vp, ip, vp, ip, ip, ip, vp]
    lib.adaf_pack_dw_weight_f32.argtypes = [vp, vp, ip, vp, vp]
    lib.adaf_dwconv3x3_bn_act_f32.argtypes = [vp, vp, ip, ip, ip, ip, ip, vp, vp, vp, ip, vp, vp]
    lib.adaf_mobilenetv2_create.argtypes = [vp, C.POINTER(vp)]
    lib.adaf_mobilenetv2_destroy.argtypes = [vp]
    lib.adaf_mobilenetv2_set_param.argtypes = [vp, C.c_char_p, vp, C.c_size_t]
    lib.adaf_mobilenetv2_finalize.argtypes = [vp, vp]
    lib.adaf_mobilenetv2_workspace_bytes.restype = C.c_size_t
    lib.adaf_mobilenetv2_workspace_bytes.argtypes = [vp, ip, ip, ip]
    lib.adaf_mobilenetv2_forward.argtypes = [vp, vp, ip, ip, ip, ip, vp, vp, ip, vp, C.c_size_t, vp]
    lib.adaf_mobilenetv2_set_fusion.argtypes = [vp, ip]
    lib.adaf_grid_actions_f32.argtypes = [vp, vp, ip, ip, vp, vp, vp, vp]
    lib.adaf_gru_seq_forward_f32.argtypes = [vp, vp, ip, ip, ip, ip, ip, vp, vp, vp, vp, vp, vp, vp, C.c_size_t, vp]
    lib.adaf_crop_gather_nhwc4_f32.argtypes = [vp, vp, ip, ip, ip, vp, ip, ip, ip, vp, vp, vp]
    lib.adaf_ingest_u8_f32.argtypes = [vp, vp, ip, ip, ip, ip, C.POINTER(C.c_float), C.POINTER(C.c_float), vp, vp]
    lib.adaf_crop_resize_f32.argtypes = [vp, vp, ip, ip, ip, ip, ip, vp, ip, ip, vp, ip, ip, vp, ip, vp, vp]
    lib.adaf_resize_nearest_f32.argtypes = [vp, vp, ip, ip, ip, ip, ip, ip, ip, vp, ip, vp]
    lib.adaf_conv2d_bn_act_f16.argtypes = [vp, C.POINTER(ConvParams), vp, ip, vp, vp, vp, vp, vp, ip, vp]
    lib.adaf_pack_conv_weight_f16.argtypes = [vp, vp, ip, ip, ip, ip, ip, vp, vp]
    lib.adaf_cast_f32_f16.argtypes = [vp, vp, C.c_size_t, vp, ip, vp]
    lib.adaf_dwconv3x3_bn_act_f16.argtypes = [vp, vp, ip, ip, ip, ip, ip, vp, vp, vp, ip, vp, vp]
    lib.adaf_pack_dw_weight_kxk_f32.argtypes = [vp, vp, ip, ip, vp, vp]
    lib.adaf_dwconv_same_workspace_bytes.restype = C.c_size_t
    lib.adaf_dwconv_same_workspace_bytes.argtypes = [ip, ip, ip, ip, ip, ip, ip]
    lib.adaf_dwconv_same_bn_act.argtypes = [vp, vp, ip, ip, ip, ip, ip, ip, ip, vp, vp, vp, ip, vp, vp, vp, C.c_size_t, vp]
    lib.adaf_se_gate_f32.argtypes = [vp, vp, ip, ip, vp, vp, ip, vp, vp, vp, vp]
    lib.adaf_conv1x1_gated_bn.argtypes = [vp, vp, ip, ip, ip, ip, vp, vp, ip, vp, vp, vp, vp, vp]
    lib.adaf_effnet_create.argtypes = [vp, fp, fp, C.POINTER(vp)]
    lib.adaf_effnet_destroy.argtypes = [vp]
    lib.adaf_effnet_feature_dim.argtypes = [vp]
    lib.adaf_effnet_block_count.argtypes = [vp]
    lib.adaf_effnet_block_info.argtypes = [vp, ip, C.POINTER(C.c_int)]
    lib.adaf_effnet_set_dtype.argtypes = [vp, ip]
    lib.adaf_effnet_set_fusion.argtypes = [vp, ip]
    lib.adaf_effnet_whole_blocks.argtypes = [vp, ip, ip]
    lib.adaf_effnet_fused_expand_blocks.argtypes = [vp, ip, ip]
    lib.adaf_effnet_set_param.argtypes = [vp, C.c_char_p, vp, C.c_size_t]
    lib.adaf_effnet_finalize.argtypes = [vp, vp]
    lib.adaf_effnet_workspace_bytes.restype = C.c_size_t
    lib.adaf_effnet_workspace_bytes.argtypes = [vp, ip, ip, ip]
    lib.adaf_effnet_forward.argtypes = [vp, vp, ip, ip, ip, ip, vp, vp, vp, ip, vp, C.c_size_t, vp]
    _lib = lib
    return lib


_handles = {}


def handle(device):
    """One library handle per device per process."""
    idx = torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    if idx not in _handles:
        lib = load_library()
        h = C.c_void_p()
        rc = lib.adaf_create(idx, C.byref(h))
        if rc != 0:
            raise AdafError("adaf_create(device=%d) failed with %d: no gfx950 (MI355X) device -- this package "
                            "has no CPU or non-gfx950 path" % (idx, rc))
        _handles[idx] = h
    return _handles[idx]


EF_PLAN_WHOLE_BLOCK, EF_PLAN_TINY_DW, EF_PLAN_STRIP_PROJECT, EF_PLAN_STRIP_EXPAND, EF_PLAN_OWN_STEM, EF_PLAN_FUSED_EXPAND, EF_PLAN_PACKED_STEM, EF_PLAN_HEAD_POOL, EF_PLAN_PAIR_CHUNKS = 1, 2, 4, 8, 16, 32, 64, 128, 256


def get_option(key):
    """Current value of a process-wide tuning / A-B switch of the library (include/adafocus.h: adaf_get_global_option)."""
    v = load_library().adaf_get_global_option(key.encode())
    if v != v:
        raise AdafError("adaf_get_global_option: unknown key %r" % key)
    return v


def set_option(key, value, device=None):
    """Set a process-wide switch (include/adafocus.h: adaf_set_global_option -- global state of the library, no handle); returns the
    previous value."""
    old = get_option(key)
    if load_library().adaf_set_global_option(key.encode(), float(value)) != 0:
        raise AdafError("adaf_set_global_option: %s = %r is out of range" % (key, value))
    return old


class option:
    """with _lib.option("conv_pool", 0): ...  -- a switch flipped for the duration of a block (tests, A/B tools)."""

    def __init__(self, key, value):
        self.key, self.value, self.old = key, value, None

    def __enter__(self):
        self.old = set_option(self.key, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.key, self.old)
        return False


def check(rc, h):
    if rc != 0:
        raise AdafError("adafocus HIP call failed (%d): %s" % (rc, load_library().adaf_last_error(h).decode()))


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def on_current_device(*tensors):
    """Kernels are launched on the CURRENT device's current stream: refuse tensors that live elsewhere (a launch on
    the wrong device would be an invalid-handle error at best, unsynchronised peer access at worst)."""
    cur = torch.cuda.current_device()
    for t in tensors:
        if t is not None and t.is_cuda and t.device.index != cur:
            raise AdafError("adafocus_amd: tensor on cuda:%d but the current device is cuda:%d -- wrap the call in "
                            "torch.cuda.device(%d) (one process per GPU is the supported mode)" % (t.device.index, cur, t.device.index))


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def need_gpu_f32(*tensors):
    """Product-path guard: the HIP path is the only path."""
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise AdafError("adafocus_amd runs on MI355X only: got a %s tensor (no CPU fallback exists)" % t.device)
        if t.dtype != torch.float32:
            raise AdafError("adafocus_amd computes in fp32: got %s" % t.dtype)
    on_current_device(*tensors)
