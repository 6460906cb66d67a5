"""Host-side mirror of the reference's ``models/utils.py`` for the hot path.

``get_patch`` keeps the reference signature and return layout (ACT/models/utils.py:37-51 =
STH/models/utils.py:44-58) but is one HIP launch with no host synchronisation instead of a
Python loop with four ``.item()`` calls per sample.  ``random_crop`` / ``zero_pad`` /
``prep_a_net`` are training-only and intentionally absent.
"""
import torch

from . import hip_ops
from ._lib import LAYOUT_NCHW, LAYOUT_NHWC4

__all__ = ["get_patch", "get_patch_nhwc4", "nchw_to_nhwc4"]


def get_patch(images, action_sequence, patch_size):
    """images (N,C,H,W) fp32 on the GPU, action_sequence (N,2) fp32 in [0,1] (row fraction, column
    fraction) -> (N,C,P,P).  Window origin = floor(action * (H - P)).int(), bit-exact."""
    return hip_ops.crop_gather(images, action_sequence.to(device=images.device, dtype=torch.float32), patch_size, 1,
                               LAYOUT_NCHW)


def get_patch_nhwc4(frames, action_sequence, patch_size, frames_per_action=1):
    """Same gather, emitted pixel-major with the channel axis padded 3 -> 4: the layout the local
    CNN's stem consumes.  frames (N,3,H,W); one action per `frames_per_action` consecutive frames."""
    return hip_ops.crop_gather(frames, action_sequence.to(device=frames.device, dtype=torch.float32), patch_size,
                               frames_per_action, LAYOUT_NHWC4)


def nchw_to_nhwc4(x):
    """(N,3,S,S) -> (N,S,S,4): the gather kernel with a full-frame window (span H-P = 0)."""
    if x.shape[2] != x.shape[3]:
        raise ValueError("nchw_to_nhwc4: square images expected")
    zeros = torch.zeros((x.shape[0], 2), device=x.device, dtype=torch.float32)
    return hip_ops.crop_gather(x, zeros, x.shape[2], 1, LAYOUT_NHWC4)
