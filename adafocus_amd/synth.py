"""Deterministic synthetic weights / inputs (numpy PCG64), shared by tests, the golden
generator and bench.py.

There is no network for checkpoints, so every parity vector and every bench number in this
repo is produced from weights drawn by :func:`synth_state_dict`.  The draw for a tensor depends
only on ``(seed, crc32(key), shape)`` so the reference model (inside ``tools/gen_golden.py``), the
oracle and the HIP path all see bit-identical parameters without a weight blob being stored.

Statistics are chosen so activations stay O(1) through 53 conv layers (SURVEY.md §7 "hard
parts"): He-style conv weights, BatchNorm running stats and affine terms randomised (default
init would make BN an identity and leave the folding untested), last BN of every residual
branch damped.
"""
import zlib

import numpy as np

__all__ = ["synth_tensor", "synth_state_dict", "synth_frames", "synth_actions", "grid_table"]


def _rng(seed, key):
    return np.random.Generator(np.random.PCG64([int(seed), zlib.crc32(key.encode())]))


def synth_tensor(key, shape, seed, kind):
    """One fp32 numpy array for parameter `key`; `kind` selects the distribution."""
    g = _rng(seed, key)
    shape = tuple(int(s) for s in shape)
    if kind == "bn_var":
        a = g.uniform(0.5, 1.5, shape)
    elif kind == "bn_mean":
        a = g.normal(0.0, 0.1, shape)
    elif kind == "bn_gamma":
        a = g.uniform(0.5, 1.5, shape)
        if ".bn3." in key:          # last BN of a ResNet bottleneck's residual branch
            a = a * 0.25
        elif "._bn2." in key:       # project BN of an EfficientNet MBConv block: 26 blocks of swish + SE neither blow up nor die out
            a = a * 0.75
    elif kind == "bn_beta":
        a = g.normal(0.0, 0.1, shape)
    elif kind == "conv":
        fan_in = int(np.prod(shape[1:]))
        a = g.normal(0.0, np.sqrt(2.0 / fan_in), shape)
    elif kind == "linear":
        a = g.normal(0.0, np.sqrt(1.0 / shape[-1]), shape)
    elif kind == "bias":
        a = g.normal(0.0, 0.05, shape)
    elif kind == "count":
        return np.zeros(shape, dtype=np.int64)
    else:
        raise ValueError(kind)
    return a.astype(np.float32)


def classify_key(key, shape, all_keys):
    """Distribution class of a state-dict entry, from its name and neighbours."""
    stem, _, leaf = key.rpartition(".")
    is_bn = (stem + ".running_mean") in all_keys
    if leaf == "num_batches_tracked":
        return "count"
    if leaf == "running_var":
        return "bn_var"
    if leaf == "running_mean":
        return "bn_mean"
    if is_bn:
        return "bn_gamma" if leaf == "weight" else "bn_beta"
    if len(shape) == 4:
        return "conv"
    if len(shape) == 2:
        return "linear"
    if len(shape) == 1:
        return "bias"
    raise ValueError("cannot classify %s %s" % (key, shape))


def synth_state_dict(shapes, seed):
    """shapes: mapping key -> shape.  Returns mapping key -> numpy array."""
    keys = set(shapes)
    return {k: synth_tensor(k, s, seed, classify_key(k, tuple(s), keys)) for k, s in shapes.items()}


def synth_frames(batch, frames, size=224, seed=0):
    """Normalised-looking video clip batch (B, T*3, H, W) ~ N(0,1), as GroupNormalize emits
    (reference ACT/ops/transforms.py:64-77)."""
    g = np.random.Generator(np.random.PCG64([int(seed), 0xF4A3E5]))
    return g.standard_normal((batch, frames * 3, size, size), dtype=np.float32)


def grid_table(side):
    """The discrete action table of the reference focuser (ACT/models/gfv_net.py:272-307):
    row-major [i/(s-1), j/(s-1)] computed in Python doubles, then rounded to fp32."""
    s = int(side)
    return np.array([[i / (s - 1), j / (s - 1)] for i in range(s) for j in range(s)], dtype=np.float64).astype(np.float32)


def synth_actions(n, side=7, seed=2):
    """Forced uniform-random grid sequence so crop windows scatter."""
    g = np.random.Generator(np.random.PCG64([int(seed), 0xAC7105]))
    idx = g.integers(0, side * side, size=n)
    return idx.astype(np.int64), grid_table(side)[idx]
