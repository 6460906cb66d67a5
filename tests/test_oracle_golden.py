"""Pin the oracle (oracle/ref_model.py) against vectors produced by the real reference
(tools/gen_golden.py).  CPU only.  These run before any HIP parity claim is trusted."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import ref_model as O
from tests.helpers import golden, rnd, synth_sd
from adafocus_amd import synth

SIZES = (96, 128, 144, 160, 176, 192)


def test_g1_tables_and_known_answers():
    g = golden("g1_crop_indices")
    for dim in (25, 36, 49, 64):
        t = O.standard_actions(dim)
        assert np.array_equal(t.numpy(), g["table_%d" % dim])
        assert np.array_equal(synth.grid_table(int(dim ** 0.5)), g["table_%d" % dim])
        for p in SIZES:
            c = O.patch_coords(t, 224, p).numpy()
            assert c.dtype == np.int32
            assert np.array_equal(c, g["coords_%d_%d" % (dim, p)])
    # known answers listed in SURVEY.md §8(a2)
    c = O.patch_coords(O.standard_actions(49), 224, 96).numpy()
    assert sorted(set(c[:, 0].tolist())) == [0, 21, 42, 64, 85, 106, 128]
    c = O.patch_coords(O.standard_actions(49), 224, 128).numpy()
    assert sorted(set(c[:, 1].tolist())) == [0, 16, 32, 48, 64, 80, 96]


def test_g1_continuous_actions_bit_exact():
    g = golden("g1_crop_indices")
    a = torch.from_numpy(g["cont_actions"])
    for p in SIZES:
        assert np.array_equal(O.patch_coords(a, 224, p).numpy(), g["cont_coords_%d" % p])


def _sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), dtype=np.uint8)


def test_g2_crop_payload():
    g = golden("g2_crop_payload")
    fr, fr2 = rnd((4, 3, 224, 224), 21), rnd((2, 24, 224, 224), 22)
    a, a2 = torch.from_numpy(g["a"]), torch.from_numpy(g["a2"])
    for p in (96, 128):
        o = O.get_patch(fr, a, p).numpy()
        o2 = O.get_patch(fr2, a2, p).numpy()
        assert o.shape == (4, 3, p, p) and o2.shape == (2, 24, p, p)
        assert np.array_equal(_sha(o), g["sha_%d" % p])
        assert np.array_equal(_sha(o2), g["sha2_%d" % p])
        assert np.array_equal(o[:, :, :8, :8], g["corner_%d" % p])
        assert np.array_equal(o2[:, :3, -8:, -8:], g["corner2_%d" % p])


def test_g3_temporal_shift():
    g = golden("g3_temporal_shift")
    x = torch.arange(2 * 8 * 16 * 3 * 3, dtype=torch.float32).view(16, 16, 3, 3)
    assert np.array_equal(O.temporal_shift(x, 8, 8).numpy(), g["out_arange"])
    assert np.array_equal(O.temporal_shift(rnd((12, 64, 2, 2), 31), 4, 8).numpy(), g["out_rand"])


def test_g4_resnet_blocks():
    g = golden("g4_resnet_blocks")
    sd = synth_sd("ACT", 404, "focuser.net.", keep_prefix=False)
    with torch.no_grad():
        stem = O.resnet50_stem(sd, "", rnd((2, 3, 32, 32), 41))
        b_ds = O.bottleneck(sd, "layer2.0", rnd((4, 256, 12, 12), 42), 2)
        b_pl = O.bottleneck(sd, "layer2.1", rnd((4, 512, 6, 6), 43), 1)
        b_l1 = O.bottleneck(sd, "layer1.0", rnd((4, 64, 12, 12), 44), 1)
        xt = rnd((2, 3, 64, 64), 45)
        trunk = O.resnet50_trunk(sd, "", xt)
        tmap = O.resnet50_trunk(sd, "", xt, pooled=False)
    for got, key in ((stem, "stem"), (b_ds, "b_ds"), (b_pl, "b_pl"), (b_l1, "b_l1"), (tmap, "trunk_map")):
        np.testing.assert_allclose(got.numpy(), g[key], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(trunk.numpy().reshape(2, -1), g["trunk"], rtol=1e-5, atol=1e-5)
    assert float(np.abs(g["trunk"]).max()) < 50 and float(np.abs(g["trunk"]).mean()) > 1e-3  # O(1) activations


def test_g5_mobilenetv2_act():
    g = golden("g5_mbv2_act")
    sd = synth_sd("ACT", 505, "glancer.net.", keep_prefix=False)
    # the stand-alone mobilenet has a 1000-way classifier; it is not on the features path
    with torch.no_grad():
        fm, fv = O.glancer_act(sd, "", rnd((2, 3, 64, 64), 53))
    np.testing.assert_allclose(fm.numpy(), g["fm"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(fv.numpy(), g["fv"], rtol=1e-5, atol=1e-5)


def test_g6_gru_classifier():
    g = golden("g6_gru_classifier")
    sd = synth_sd("ACT", 606, "classifier.", keep_prefix=False)
    with torch.no_grad():
        logits, last = O.recurrent_classifier(sd, "", rnd((2, 8, 3328), 61, 0.5))
    np.testing.assert_allclose(logits.numpy(), g["logits"], rtol=1e-5, atol=2e-5)
    np.testing.assert_allclose(last.numpy(), g["last"], rtol=1e-5, atol=2e-5)


@pytest.mark.parametrize("per_step", [True, False])
def test_g7_act_end_to_end(per_step):
    g = golden("g7_act_e2e")
    sd = synth_sd("ACT", 1007)
    frames = torch.from_numpy(synth.synth_frames(2, 8, 224, seed=0))
    with torch.no_grad():
        logits, last, idx, feat = O.act_forward(sd, frames, frames, 96, 49, per_step=per_step, return_aux=True)
        forced = torch.from_numpy(g["forced_idx"])
        logits_f, last_f = O.act_forward(sd, frames, frames, 96, 49, forced_action_idx=forced, per_step=per_step)
    assert np.array_equal(idx.numpy(), g["policy_idx"])
    np.testing.assert_allclose(feat[:, :, :1280].numpy(), g["glancer_vec"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(logits.numpy(), g["logits"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(last.numpy(), g["last"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(logits_f.numpy(), g["logits_forced"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(last_f.numpy(), g["last_forced"], rtol=1e-4, atol=2e-5)
    assert len(set(g["forced_idx"].reshape(-1).tolist())) > 8     # crops really vary


def test_g7_sth_end_to_end():
    g = golden("g7_sth_e2e")
    sd = synth_sd("STH", 1007)
    sd.update(synth_sd("STH_POLICY", 1007))
    sd = O.canonical_resnet_keys(sd, "focuser.net.base_model.")
    gl = torch.from_numpy(synth.synth_frames(2, 8, 224, seed=3))
    fo = torch.from_numpy(synth.synth_frames(2, 8, 224, seed=4)).view(2, 8, 3, 224, 224)
    with torch.no_grad():
        logit, patch, action = O.sth_forward(sd, gl, fo, 128, 8, 8)
        forced = torch.from_numpy(g["forced_action"])
        logit_f, patch_f, _ = O.sth_forward(sd, gl, fo, 128, 8, 8, forced_action=forced)
    np.testing.assert_allclose(action.numpy(), g["policy_action"], rtol=1e-5, atol=1e-6)
    assert np.array_equal(patch[:, :, :, :4, :4].numpy(), g["patch_corner"])
    assert np.array_equal(patch_f[:, :, :, :4, :4].numpy(), g["patch_forced_corner"])
    np.testing.assert_allclose(logit.numpy(), g["logits"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(logit_f.numpy(), g["logits_forced"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(logit_f.numpy(), g["logits_stage3_forced"], rtol=1e-4, atol=2e-5)


# ---- the plain-C restatement (oracle/crop_ref.c) against the same reference vectors ----------
def _c_oracle():
    import ctypes
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = os.path.join(root, "oracle", "libcrop_ref.so")
    if not os.path.exists(so):
        subprocess.run(["make", "-C", os.path.join(root, "oracle")], check=True)
    return ctypes.CDLL(so)


def _p(a):
    import ctypes
    return a.ctypes.data_as(ctypes.c_void_p)


def test_c_oracle_coords_and_payload():
    lib = _c_oracle()
    g = golden("g1_crop_indices")
    a = np.ascontiguousarray(g["cont_actions"])
    for p in SIZES:
        out = np.empty((a.shape[0], 2), dtype=np.int32)
        lib.ref_patch_coords(_p(a), a.shape[0], 224, p, _p(out))
        assert np.array_equal(out, g["cont_coords_%d" % p])
    g2 = golden("g2_crop_payload")
    fr = np.ascontiguousarray(rnd((2, 24, 224, 224), 22).numpy())
    act = np.ascontiguousarray(g2["a2"])
    out = np.empty((2, 24, 128, 128), dtype=np.float32)
    lib.ref_get_patch(_p(fr), 2, 24, 224, 224, _p(act), 128, _p(out))
    assert np.array_equal(_sha(out), g2["sha2_128"])


def test_c_oracle_temporal_shift():
    lib = _c_oracle()
    g = golden("g3_temporal_shift")
    x = np.arange(2 * 8 * 16 * 3 * 3, dtype=np.float32)
    out = np.empty_like(x)
    lib.ref_temporal_shift(_p(x), 16, 16, 9, 8, 8, _p(out))
    assert np.array_equal(out.reshape(16, 16, 3, 3), g["out_arange"])


def test_g8_ingest_uint8():
    g = golden("g8_ingest")
    gen8 = np.random.Generator(np.random.PCG64([88, 0xC0]))
    u8 = gen8.integers(0, 256, size=(40, 56, 4 * 3), dtype=np.uint8)
    u8[0, 0, :] = 0
    u8[0, 1, :] = 255
    out = O.ingest_uint8(u8).numpy()
    assert out.shape == (12, 40, 56)
    assert np.array_equal(_sha(out), g["sha"])
    assert np.array_equal(out[:, :4, :4], g["corner"])
