"""Host mirror of STH/ops/basic_ops.py: ``ConsensusModule('avg')`` = temporal mean with keepdim
(:17-26).  Stand-alone it is a plain reduction; on the model path it is fused with the classifier FC in
``adaf_fc_meanpool_forward_f32``."""
import torch

__all__ = ["ConsensusModule", "SegmentConsensus", "Identity"]


class Identity(torch.nn.Module):
    def forward(self, input):
        return input


class SegmentConsensus(torch.nn.Module):
    def __init__(self, consensus_type, dim=1):
        super().__init__()
        self.consensus_type = consensus_type
        self.dim = dim

    def forward(self, input_tensor):
        if self.consensus_type == "avg":
            return input_tensor.mean(dim=self.dim, keepdim=True)
        if self.consensus_type == "identity":
            return input_tensor
        return None


class ConsensusModule(torch.nn.Module):
    def __init__(self, consensus_type, dim=1):
        super().__init__()
        self.consensus_type = consensus_type if consensus_type != "rnn" else "identity"
        self.dim = dim

    def forward(self, input):
        return SegmentConsensus(self.consensus_type, self.dim)(input)
