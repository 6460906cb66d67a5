"""Host mirror of STH/ops/basic_ops.py (:17-26): the temporal-consensus module the STH model applies to its per-frame
logits.  Only two behaviours exist in the reference -- 'avg' (mean over the segment axis, kept as a size-1 axis) and
'identity' (also what 'rnn' maps to); anything else yields None there, and here.  Stand-alone this is a plain reduction;
on the model path the mean is fused with the classifier FC in ``adaf_fc_meanpool_forward_f32``."""
import torch
from torch import nn

__all__ = ["ConsensusModule", "SegmentConsensus", "Identity"]

_REDUCERS = {
    "avg": lambda x, axis: torch.mean(x, axis, keepdim=True),
    "identity": lambda x, axis: x,
}


def _consensus(kind, x, axis):
    fn = _REDUCERS.get(kind)
    return fn(x, axis) if fn is not None else None


class Identity(nn.Module):
    """Pass-through (same name as the reference's helper)."""

    def forward(self, x):
        return x


class SegmentConsensus(nn.Module):
    """Reduction over the segment axis selected by name."""

    def __init__(self, consensus_type, dim=1):
        super().__init__()
        self.consensus_type, self.dim = consensus_type, dim

    def forward(self, x):
        return _consensus(self.consensus_type, x, self.dim)


class ConsensusModule(nn.Module):
    """Constructor-compatible with the reference: ``ConsensusModule('avg')(logits[B, T, C]) -> [B, 1, C]``."""

    def __init__(self, consensus_type, dim=1):
        super().__init__()
        self.consensus_type = "identity" if consensus_type == "rnn" else consensus_type
        self.dim = dim

    def forward(self, x):
        return _consensus(self.consensus_type, x, self.dim)
