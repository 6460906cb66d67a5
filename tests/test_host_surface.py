"""Small host-side pieces of the reference surface that carry no kernels (CPU only)."""
import torch

from adafocus_amd.basic_ops import ConsensusModule, Identity, SegmentConsensus
from adafocus_amd.synth import grid_table


def test_consensus_module_semantics():
    """STH/ops/basic_ops.py:17-26: 'avg' = mean over the segment axis kept as a size-1 axis, 'rnn' -> identity,
    unknown types give None; used as ``ConsensusModule(args.consensus_type)(logits.view(B, T, C)).squeeze(1)``
    (STH/models/gfv_net.py:164-174)."""
    x = torch.arange(2 * 8 * 5, dtype=torch.float32).view(2, 8, 5)
    out = ConsensusModule("avg")(x)
    assert out.shape == (2, 1, 5) and torch.equal(out, x.mean(dim=1, keepdim=True))
    assert torch.equal(ConsensusModule("avg", dim=2)(x), x.mean(dim=2, keepdim=True))
    assert ConsensusModule("rnn")(x) is x and ConsensusModule("identity")(x) is x
    assert ConsensusModule("max")(x) is None
    assert torch.equal(SegmentConsensus("avg")(x), out) and Identity()(x) is x


def test_grid_tables_are_row_major_fp32_fractions():
    """ACT/models/gfv_net.py:272-307: torch.Tensor([[i/(s-1), j/(s-1)] ...]) -- python doubles rounded to fp32, first
    coordinate = row."""
    for s in (5, 6, 7, 8):
        t = torch.from_numpy(grid_table(s))
        ref = torch.tensor([[i / (s - 1), j / (s - 1)] for i in range(s) for j in range(s)], dtype=torch.float32)
        assert t.dtype == torch.float32 and torch.equal(t, ref)
