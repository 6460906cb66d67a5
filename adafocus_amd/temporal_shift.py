"""Temporal Shift Module -- host mirror of STH/ops/temporal_shift.py.

``TemporalShift.shift`` (temporal_shift.py:28-46) is a HIP kernel (`adaf_temporal_shift_f32`);
inside the local CNN the shift is never materialised at all with the shipped placement ('blockres'):
``make_temporal_shift`` only marks the ResNet so the trunk fuses it into every Bottleneck conv1's operand load
(DESIGN.md §3.2); place='block' materialises the shifted block input once per block.
``InplaceShift`` / ``TemporalPool`` are dead code in the reference (inplace raises, :36-38) and absent.
"""
from torch import nn

from . import hip_ops

__all__ = ["TemporalShift", "make_temporal_shift"]


class TemporalShift(nn.Module):
    def __init__(self, net, n_segment=3, n_div=8, inplace=False):
        super().__init__()
        if inplace:
            raise NotImplementedError("in-place shift is not implemented (neither is it in the reference)")
        self.net = net
        self.n_segment = n_segment
        self.fold_div = n_div
        self.inplace = False

    def forward(self, x):
        return self.net(self.shift(x, self.n_segment, fold_div=self.fold_div))

    @staticmethod
    def shift(x, n_segment, fold_div=3, inplace=False):
        """x (N*T, C, H, W) on the GPU -> shifted copy; zero rows at clip boundaries."""
        if inplace:
            raise NotImplementedError
        return hip_ops.temporal_shift(x, n_segment, fold_div, hip_ops.LAYOUT_NCHW)


def make_temporal_shift(net, n_segment, n_div=8, place="blockres", temporal_pool=False):
    """temporal_shift.py:99-142 for ResNet-50 without temporal pooling.  place='blockres' (every shipped configuration,
    :123-140): every Bottleneck conv1 sees the shifted block input -- fused into that conv's operand load.  place='block'
    (:104-121): TemporalShift wraps the whole Bottleneck, so conv1, the downsample conv and the identity all see it -- the shifted
    map is materialised once per block (adaf_resnet50_set_shift_place)."""
    if temporal_pool:
        raise NotImplementedError("temporal_pool is unused by the reference drivers")
    if place == "block":
        net.tsm_place = "block"
    elif "blockres" in place:
        net.tsm_place = "blockres"
    else:
        # (the reference silently inserts NO shift for any other string -- its `else: raise` belongs to the isinstance test,
        #  temporal_shift.py:111-142; refused here rather than running a model that is not the one asked for)
        raise NotImplementedError(place)
    net.tsm_segments = int(n_segment)
    net.tsm_div = int(n_div)
    return net
