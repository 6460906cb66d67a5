import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` on a box without a GPU: skip (not fail) -- every gpu-marked test needs the device, and the HIP path has no fallback."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no MI355X visible (gpu-marked tests run on the GPU box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", params=["f32", "split_bf16"])
def trunk_math(request):
    """The GPU parity modules run twice: with the local CNN's convolutions on the fp32 matrix pipe (the reported
    configuration) and with the opt-in split-bf16 arithmetic as the default of every ResNet built."""
    from adafocus_amd import resnet
    old = resnet.DEFAULT_MATH
    resnet.DEFAULT_MATH = request.param
    yield request.param
    resnet.DEFAULT_MATH = old
