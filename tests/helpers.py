"""Shared test helpers: golden loading, manifest-driven synthetic state dicts."""
import os

import numpy as np
import torch

from adafocus_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def rnd(shape, seed, scale=1.0):
    """Same generator as tools/gen_golden.py:rnd (inputs are stored as seeds, not arrays)."""
    g = np.random.Generator(np.random.PCG64([seed, 0xBEEF]))
    return torch.from_numpy(g.standard_normal(shape, dtype=np.float32) * np.float32(scale))


def load_tool(name):
    """Import tools/<name>.py as a module (the A/B tools double as test fixtures: tests call their digests())."""
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", name + ".py")
    spec = importlib.util.spec_from_file_location("adaf_tool_" + name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_MANIFEST = None


def manifest():
    """tag -> {key: shape} from tests/golden/state_dict_manifest.txt (written by gen_golden)."""
    global _MANIFEST
    if _MANIFEST is None:
        m = {}
        with open(os.path.join(GOLDEN, "state_dict_manifest.txt")) as f:
            for line in f:
                tag, key, shp = line.split()
                shape = () if shp == "scalar" else tuple(int(d) for d in shp.split("x"))
                m.setdefault(tag, {})[key] = shape
        _MANIFEST = m
    return _MANIFEST


def synth_sd(tag, seed, strip_prefix="", keep_prefix=True):
    """Synthetic torch state dict for the keys of `tag` that start with strip_prefix.
    keep_prefix=False generates under the stripped names (as the golden generator did when it
    instantiated the sub-module on its own)."""
    shapes = {}
    for k, s in manifest()[tag].items():
        if k.startswith(strip_prefix):
            shapes[k if keep_prefix else k[len(strip_prefix):]] = s
    sd = synth.synth_state_dict(shapes, seed)
    return {k: torch.from_numpy(v) for k, v in sd.items()}
