"""CPU checks of the EfficientNet restatement (oracle/ref_effnet.py) and of the Python mirror's state-dict layout.

PARITY UNPINNED: the reference holds no EfficientNet output (dead import only).  What it does hold --
feature dimension 1536 and the (1.80 GFLOPs, 12 M parameters) prior for "efficientnet-b3" (STH/ops/net_flops_table.py:17,29)
-- is asserted here, together with the published figures of the package it names (EfficientNet paper, Table 2: B0 0.39 B
multiply-adds / 5.3 M parameters; B3 1.8 B / 12 M at 300^2) and known answers of its SAME-padding rule.
"""
import torch
import torch.nn.functional as F

from adafocus_amd import synth
from oracle import ref_effnet as R


def test_reference_held_numbers_for_b3():
    macs, params = R.count_macs_params("efficientnet-b3")            # native 300^2
    assert abs(macs / 1e9 - 1.80) / 1.80 < 0.03, macs                # net_flops_table.py:29 (1.80, 12)
    assert abs(params / 1e6 - 12.0) / 12.0 < 0.03, params
    assert R.head_channels(1.2) == 1536                              # net_flops_table.py:17 feat_dim_dict
    # the reference's own rescaling rule (net_flops_table.py:35-37) at config 5's 144^2 patches
    prior = 1.80 / 224 / 224 * 144 * 144
    actual = R.count_macs_params("efficientnet-b3", 144)[0] / 1e9
    assert abs(prior - 0.744) < 1e-3 and abs(actual - 0.432) < 2e-3  # the table files the 300^2 figure under 224


def test_published_numbers_for_b0():
    macs, params = R.count_macs_params("efficientnet-b0")
    assert abs(macs / 1e9 - 0.39) < 0.01 and abs(params / 1e6 - 5.3) < 0.05


def test_b3_topology():
    bl = R.block_list(1.2, 1.4)
    assert len(bl) == 26 and R.stem_channels(1.2) == 40
    assert [b["cout"] for b in bl if b["stride"] == 2 or b is bl[0]] == [24, 32, 48, 96, 232]
    assert sorted(set(b["cout"] for b in bl)) == [24, 32, 48, 96, 136, 232, 384]
    assert [b["k"] for b in bl] == [3] * 5 + [5] * 3 + [3] * 5 + [5] * 11 + [3] * 2
    assert bl[0]["expand"] == 1 and bl[1]["sq"] == 6 and bl[-1]["hid"] == 2304 and bl[-1]["sq"] == 96


def test_byte_models_of_the_launch_plan_for_b3_at_144():
    """bench.py's three byte counts for config 5 (workload.effnet_*): the block-level figure, the plan that runs (whole-image blocks 9-17 and
    19-24, the expand conv inside the depthwise launch for blocks 2-8) and the four-launch plan; the Python mirrors of the two eligibility
    rules pick the blocks the library picks (the GPU tests assert the library's own counts: 15 and 7)."""
    from adafocus_amd import workload as W
    _, blocks, _ = W.effnet_blocks("efficientnet-b3", 144)
    whole = [i for i, b in enumerate(blocks) if W.effnet_whole_block(b)]
    fused = [i for i, b in enumerate(blocks) if W.effnet_fused_expand_block(b)]
    assert whole == list(range(9, 18)) + list(range(19, 25)) and fused == list(range(2, 9))
    assert [W.effnet_fused_expand_slices(blocks[i]) for i in fused] == [3, 3, 3, 3, 6, 6, 6]          # 144 / 48, 192 / 64, 288 / 48
    assert not any(W.effnet_fused_expand_block(b, 4) or W.effnet_whole_block(b, 4) for b in blocks)  # fp32 storage: four launches
    blk, plan, plain = W.effnet_block_bytes_per_frame(), W.effnet_bytes_per_frame(), W.effnet_bytes_per_frame(fused=False)
    assert abs(blk / 1e6 - 4.06) < 0.01 and abs(plan / 1e6 - 11.87) < 0.01 and abs(plain / 1e6 - 23.15) < 0.01
    assert blk < plan < plain


def test_same_padding_known_answers():
    # Conv2dStaticSamePadding: total = max((ceil(i/s) - 1) s + k - i, 0), before = total // 2
    assert R.same_pad(144, 3, 2) == (0, 1) and R.same_pad(300, 3, 2) == (0, 1)
    assert R.same_pad(36, 5, 2) == (1, 2) and R.same_pad(75, 5, 2) == (2, 2)
    assert R.same_pad(9, 5, 2) == (2, 2) and R.same_pad(18, 3, 2) == (0, 1) and R.same_pad(19, 3, 2) == (1, 1)
    assert R.same_pad(72, 3, 1) == (1, 1) and R.same_pad(5, 5, 1) == (2, 2)
    x = torch.randn(1, 4, 9, 9)
    w = torch.randn(4, 1, 5, 5)
    assert torch.equal(R._conv_same(x, w, 1, 9, groups=4), F.conv2d(x, w, None, 1, 2, 1, 4))     # odd k, stride 1: symmetric
    assert R._conv_same(x, w, 2, 9, groups=4).shape[-1] == 5


def test_oracle_runs_and_static_padding_differs_only_where_it_should():
    shapes = R.state_dict_shapes("efficientnet-b0", 10)
    sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict(shapes, 3).items()}
    x = torch.randn(1, 3, 64, 64)
    with torch.no_grad():
        f = R.extract_features(sd, x, "efficientnet-b0", image_size=None)
        assert f.shape == (1, 1280, 2, 2)
        assert torch.equal(R.extract_features(sd, x, "efficientnet-b0"), R.extract_features(sd, x, "efficientnet-b0", image_size=224))
        assert torch.equal(R.extract_features(sd, x, "efficientnet-b0", image_size=64), f)
        # padding computed for another resolution: 100 -> 50 -> 50 -> 25 -> 13: the 5x5 / stride-2 conv pads (2, 2) there but
        # (1, 2) on the actual 16-pixel map; 224's chain happens to give the same pads as 64's everywhere
        assert torch.equal(R.extract_features(sd, x, "efficientnet-b0", image_size=224), f)
        g = R.extract_features(sd, x, "efficientnet-b0", image_size=100)
        assert g.shape == f.shape and not torch.equal(g, f)
        assert R.features_pooled(sd, x, "efficientnet-b0").shape == (1, 1280)


def test_python_mirror_has_the_package_state_dict_layout():
    from adafocus_amd.efficientnet import EfficientNet
    for name, classes in (("efficientnet-b0", 1000), ("efficientnet-b3", 200)):
        m = EfficientNet.from_name(name, num_classes=classes)
        got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        want = {k: tuple(s) for k, s in R.state_dict_shapes(name, classes).items()}
        assert got == want, set(got) ^ set(want)
        assert m.feature_dim == R.head_channels(R.PARAMS[name][0])
        m.load_state_dict({k: torch.zeros(s) if s else torch.tensor(0) for k, s in want.items()}, strict=True)
    assert EfficientNet.from_name("efficientnet-b3").feature_dim == 1536


def test_local_cnn_wrapper_registers_the_classifier_once():
    from adafocus_amd.mbconv_local import EfficientNetLocalCNN
    e = EfficientNetLocalCNN("efficientnet-b0", num_classes=7)
    assert e.fc is e._fc and sum(k.endswith("fc.weight") for k in e.state_dict()) == 1
    assert e.image_size == 224                    # the package's default is static padding for the native resolution
    assert EfficientNetLocalCNN("efficientnet-b3", image_size=None).image_size is None
    assert EfficientNetLocalCNN("efficientnet-b3", image_size=144).image_size == 144
