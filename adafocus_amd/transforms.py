"""Device-side tail of the reference's data pipeline (row f1 of SURVEY.md §8f).

The reference normalises on the host (``Stack`` -> ``ToTorchFormatTensor`` -> ``GroupNormalize``,
ACT/ops/transforms.py:305-336,64-77) and ships fp32 frames over PCIe (617 MB per 64-clip batch at
T = 16).  Here the loader's stacked uint8 clip goes to the GPU as is (4x fewer bytes) and one kernel
emits the normalised pixel-major frames that both the glancer and the patch gather consume.
JPEG decode / resize / centre-crop stay with the loader (out of scope).
"""
from . import hip_ops

__all__ = ["INPUT_MEAN", "INPUT_STD", "ingest_uint8"]

INPUT_MEAN = (0.485, 0.456, 0.406)   # GFV.input_mean / input_std
INPUT_STD = (0.229, 0.224, 0.225)


def ingest_uint8(clips_hwc_u8, num_frames, mean=INPUT_MEAN, std=INPUT_STD):
    """(B, H, W, T*3) uint8 on the GPU -> (B*T, H, W, 4) fp32; bit-exact with
    ``GroupNormalize(mean, std)(ToTorchFormatTensor()(stacked))`` in the first three lanes."""
    return hip_ops.ingest_u8(clips_hwc_u8, num_frames, mean, std)
