#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the REAL reference (read-only, /root/reference).

Runs only in the build container (the reference cannot travel to the GPU box); the fixtures it
writes are data: seeded inputs (or their seeds) and the reference's outputs.  No reference source
is copied anywhere.  Three harness shims make the reference importable on CPU:
  1. a stub ``torchvision`` (constructor-only transforms; for STH ``models.resnet50`` is bound to
     the reference's own vendored, byte-identical torchvision ResNet source),
  2. ``Tensor.cuda`` / ``Module.cuda`` patched to identity (hard-coded .cuda() calls),
  3. ``pretrained=True`` downloads disabled.

Usage:  python tools/gen_golden.py            (writes every fixture)
"""
import importlib
import os
import sys
import types
import hashlib

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from adafocus_amd import synth  # noqa: E402

REF = "/root/reference"
ACT = os.path.join(REF, "Experiments on ActivityNet, FCVID and Mini-Kinetics")
STH = os.path.join(REF, "Experiments on Something-Something V1&V2")
OUT = os.path.join(REPO, "tests", "golden")


# ------------------------------------------------------------------ shims
def _install_shims():
    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")
    md = types.ModuleType("torchvision.models")

    class _Ctor:
        def __init__(self, *a, **k):
            pass

    tr.Compose = tr.Resize = tr.CenterCrop = _Ctor
    tv.transforms, tv.models = tr, md
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.transforms"] = tr
    sys.modules["torchvision.models"] = md
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    return md


def _enter_tree(root):
    for name in list(sys.modules):
        if name.split(".")[0] in ("models", "ops", "basic_tools"):
            del sys.modules[name]
    for p in (ACT, STH):
        if p in sys.path:
            sys.path.remove(p)
    sys.path.insert(0, root)
    importlib.invalidate_caches()


class Args:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def act_args(**over):
    a = dict(num_segments=8, num_classes=200, reward="random", dataset="actnet", input_size=224, batch_size=2,
             patch_size=96, with_glancer=True, feature_map_channels=1280, glance_size=224, action_dim=49,
             hidden_state_dim=1024, policy_conv=True, gpu=None, continuous=False, gamma=0.7, policy_lr=0.0003,
             random_patch=False, dropout=0.5, consensus="gru", hidden_dim=1024)
    a.update(over)
    return Args(**a)


def sth_args(**over):
    a = dict(num_segments_glancer=8, num_segments_focuser=8, num_classes=174, batch_size=2, patch_size=128,
             with_glancer=True, feature_map_channels=1280, video_div=1, glance_size=224, action_dim=49,
             hidden_state_dim=1024, policy_conv=True, gpu=None, ppo_continuous=True, gamma=0.7, policy_lr=0.0003,
             action_std=0.25, actorcritic_with_bn=True, modality="RGB", base_model="resnet50", partial_bn=False,
             pretrain="imagenet", is_shift=True, shift_div=8, shift_place="blockres", fc_lr5=False,
             temporal_pool=False, non_local=False, random_patch=False, dropout=0.5)
    a.update(over)
    return Args(**a)


def load_synth(module, seed):
    shapes = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    sd = synth.synth_state_dict(shapes, seed)
    module.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=True)
    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(np.ascontiguousarray(sd[k]).tobytes())
    return shapes, h.hexdigest()


def rnd(shape, seed, scale=1.0):
    g = np.random.Generator(np.random.PCG64([seed, 0xBEEF]))
    return (g.standard_normal(shape, dtype=np.float32) * np.float32(scale))


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


# ------------------------------------------------------------------ margins (so the GPU tests need no escape hatch)
ARGMAX_GAP_MIN = 2e-3      # top-1 minus top-2 actor log-probability: 20x the 1e-4 the HIP policy's logits may differ by
PIXEL_MARGIN_MIN = 0.02    # distance of action * (H - P) from an integer, in pixels (1e-3 action error at H - P = 96 is 0.1 px;
                           # the HIP policy's actions differ from torch-CPU's by ~1e-6)


class ActorGap:
    """Forward hook on the REFERENCE policy's `actor` head: records every output while active.  Discrete policies
    (ACT/models/ppo.py:49-53: Linear + Softmax) -> `gaps()` = log p(top-1) - log p(top-2) per (call, clip), the logit gap an
    arg-max flip would have to overcome; continuous (STH/models/ppo_continuous.py:59-61: Linear + Sigmoid) -> `outputs`."""

    def __init__(self, actor):
        self.actor, self.outputs = actor, []

    def __enter__(self):
        self._h = self.actor.register_forward_hook(lambda m, i, o: self.outputs.append(o.detach().clone()))
        return self

    def __exit__(self, *exc):
        self._h.remove()

    def gaps(self):
        out = []
        for probs in self.outputs:
            top = torch.log(probs).topk(2, dim=1)[0]
            out.append((top[:, 0] - top[:, 1]).numpy())
        return np.stack(out, 1).astype(np.float32)          # (B, calls)


def pixel_margin(action, image_size, patch_size):
    """Distance of every crop coordinate `action * (H - P)` (ACT/models/utils.py:42) from the nearest integer, in pixels."""
    v = np.asarray(action, dtype=np.float64) * (image_size - patch_size)
    return np.abs(v - np.round(v)).astype(np.float32)


# ------------------------------------------------------------------ ACT tree
def gen_act():
    _enter_tree(ACT)
    import models.gfv_net as G
    import models.utils as U
    import models.resnet as R
    import models.mobilenet as M
    _rn, _mb = G.resnet50, G.mobilenet_v2
    G.resnet50 = lambda pretrained=False, **k: _rn(pretrained=False, **k)
    G.mobilenet_v2 = lambda pretrained=False, **k: _mb(pretrained=False, **k)

    # ---- G1: crop-index tables for every grid x patch size, plus continuous actions
    g1 = {}
    sizes = [96, 128, 144, 160, 176, 192]
    model = G.GFV(act_args())
    for dim, table in model.focuser.standard_actions_set.items():
        g1["table_%d" % dim] = table.numpy()
        for p in sizes:
            g1["coords_%d_%d" % (dim, p)] = torch.floor(table * (224 - p)).int().numpy()
    gen = np.random.Generator(np.random.PCG64([11, 0xC0]))
    cont = 1.0 / (1.0 + np.exp(-gen.standard_normal((4096, 2)).astype(np.float32)))
    cont = cont.astype(np.float32)
    cont[0] = (0.0, 1.0)
    cont[1] = (1.0, 0.0)
    k = 2
    for s in (5, 6, 7, 8):          # nextafter neighbours of every k/(s-1)
        for i in range(s):
            v = np.float32(i / (s - 1))
            for w in (np.nextafter(v, np.float32(-1)), v, np.nextafter(v, np.float32(2))):
                cont[k] = (min(max(w, 0.0), 1.0), v)
                k += 1
    g1["cont_actions"] = cont
    for p in sizes:
        # the reference expression itself, utils.py:42
        g1["cont_coords_%d" % p] = torch.floor(torch.from_numpy(cont) * (224 - p)).int().numpy()
    save("g1_crop_indices", **g1)

    # ---- G2: crop payload (ACT per-frame coords and STH per-clip coords share get_patch)
    fr = rnd((4, 3, 224, 224), 21)
    a = np.array([[0.0, 1.0], [1.0, 0.0], [0.5, 1 / 6], [0.3337, 0.81]], dtype=np.float32)
    fr2 = rnd((2, 24, 224, 224), 22)
    a2 = np.array([[0.25, 0.75], [0.99, 0.01]], dtype=np.float32)
    g2 = {"seed_frames": np.array([21, 22]), "a": a, "a2": a2}
    for p in (96, 128):
        o = U.get_patch(torch.from_numpy(fr), torch.from_numpy(a), p).numpy()
        o2 = U.get_patch(torch.from_numpy(fr2), torch.from_numpy(a2), p).numpy()
        g2["sha_%d" % p] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(o).tobytes()).digest(), dtype=np.uint8)
        g2["sha2_%d" % p] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(o2).tobytes()).digest(), dtype=np.uint8)
        g2["corner_%d" % p] = o[:, :, :8, :8].copy()
        g2["corner2_%d" % p] = o2[:, :3, -8:, -8:].copy()
    save("g2_crop_payload", **g2)

    # ---- G4: ResNet pieces with randomised BN: stem, bottleneck w/ stride+downsample, plain bottleneck
    net = R.resnet50(pretrained=False)
    net.eval()
    load_synth(net, 404)
    with torch.no_grad():
        xs = torch.from_numpy(rnd((2, 3, 32, 32), 41))
        stem = net.maxpool(net.relu(net.bn1(net.conv1(xs))))
        x1 = torch.from_numpy(rnd((4, 256, 12, 12), 42))
        b_ds = net.layer2[0](x1)            # 256->512, stride 2, downsample
        x2 = torch.from_numpy(rnd((4, 512, 6, 6), 43))
        b_pl = net.layer2[1](x2)            # plain
        x3 = torch.from_numpy(rnd((4, 64, 12, 12), 44))
        b_l1 = net.layer1[0](x3)            # 64->256, stride 1, downsample
        xt = torch.from_numpy(rnd((2, 3, 64, 64), 45))
        trunk = net.get_featmap(xt, pooled=True)
        trunk_map = net.get_featmap(xt, pooled=False)
    save("g4_resnet_blocks", seed_weights=np.array([404]), seeds_in=np.array([41, 42, 43, 44, 45]),
         stem=stem.numpy(), b_ds=b_ds.numpy(), b_pl=b_pl.numpy(), b_l1=b_l1.numpy(),
         trunk=trunk.numpy().reshape(2, -1), trunk_map=trunk_map.numpy())

    # ---- G5: MobileNetV2 inverted residuals (ACT variant): t=1, t=6 s=1 residual, t=6 s=2; + whole net small
    mb = M.mobilenet_v2(pretrained=False)
    mb.eval()
    load_synth(mb, 505)
    with torch.no_grad():
        xa = torch.from_numpy(rnd((2, 32, 16, 16), 51))
        r1 = mb.features[1](xa)              # t=1
        xb = torch.from_numpy(rnd((2, 24, 16, 16), 52))
        r2 = mb.features[3](xb)              # t=6, s=1, residual
        r3 = mb.features[4](xb)              # t=6, s=2
        xc = torch.from_numpy(rnd((2, 3, 64, 64), 53))
        fm, fv = mb.get_featmap(xc)
    save("g5_mbv2_act", seed_weights=np.array([505]), seeds_in=np.array([51, 52, 53]), r1=r1.numpy(),
         r2=r2.numpy(), r3=r3.numpy(), fm=fm.numpy(), fv=fv.numpy())

    # ---- G6: GRU classifier
    cls = G.RecurrentClassifier(seq_len=8, input_dim=3328, batch_size=2, hidden_dim=1024, num_classes=200,
                                dropout=0.5)
    cls.eval()
    load_synth(cls, 606)
    with torch.no_grad():
        feat = torch.from_numpy(rnd((2, 8, 3328), 61, 0.5))
        logits, last = cls(feat)
    save("g6_gru_classifier", seed_weights=np.array([606]), seed_in=np.array([61]), logits=logits.numpy(),
         last=last.numpy())

    # ---- G7a: end-to-end ACT, config 1 (T=8, P=96, B=2), policy actions + forced actions
    model = G.GFV(act_args())
    model.eval()
    shapes, sha = load_synth(model, 1007)
    frames = torch.from_numpy(synth.synth_frames(2, 8, 224, seed=0))
    with torch.no_grad():
        logits, last = model(input=frames, scan=frames, training=False, backbone_pred=False, one_step=True, gpu=None)
        # record the policy's own choices by replaying select_action
        fm, fv = model.glance(frames)
        idxs = []
        with ActorGap(model.focuser.policy.policy_old.actor) as ag:
            for s in range(8):
                idxs.append(model.focuser.policy.select_action(fm[:, s], model.focuser.memory, s == 0, False))
        pol_idx = torch.stack(idxs, 1)
        pol_gap = ag.gaps()
        assert pol_gap.min() >= ARGMAX_GAP_MIN, "G7a: arg-max near-tie (%g): pick another frame seed" % pol_gap.min()
        # forced actions: patch the policy so crops vary
        forced_idx, _ = synth.synth_actions(16, 7, seed=2)
        forced = torch.from_numpy(forced_idx).view(2, 8)
        step = {"s": 0}

        def fake_select(state, memory, restart_batch=False, training=True):
            if restart_batch:
                step["s"] = 0
            r = forced[:, step["s"]]
            step["s"] += 1
            return r
        model.focuser.policy.select_action = fake_select
        logits_f, last_f = model(input=frames, scan=frames, training=False, backbone_pred=False, one_step=True,
                                 gpu=None)
    # ---- G8: loader tail (Stack output -> ToTorchFormatTensor -> GroupNormalize), the reference's own classes
    import ops.transforms as TR
    gen8 = np.random.Generator(np.random.PCG64([88, 0xC0]))
    u8 = gen8.integers(0, 256, size=(40, 56, 4 * 3), dtype=np.uint8)
    u8[0, 0, :] = 0
    u8[0, 1, :] = 255
    norm = TR.GroupNormalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])(TR.ToTorchFormatTensor(div=True)(u8.copy()))
    save("g8_ingest", seed=np.array([88]), sha=np.frombuffer(hashlib.sha256(np.ascontiguousarray(norm.numpy()).tobytes()).digest(), dtype=np.uint8),
         corner=norm[:, :4, :4].numpy().copy())

    # ---- G9: evaluation metrics over logits (the reference's own accuracy / cal_map)
    import ops.utils as UT
    gen9 = np.random.Generator(np.random.PCG64([99, 0xC0]))
    lg = torch.from_numpy(gen9.standard_normal((300, 20)).astype(np.float32) * 2)
    tg = torch.from_numpy(gen9.integers(0, 20, size=(300, 1)).astype(np.int64))
    lg[torch.arange(300), tg[:, 0]] += 1.5                       # make the scores informative
    a1, a5 = UT.accuracy(lg, tg[:, 0], topk=(1, 5))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m_ap, ap = UT.cal_map(lg, tg)
        tg2 = torch.cat([tg, torch.full((300, 1), -1, dtype=torch.int64)], 1)
        tg2[::7, 1] = (tg2[::7, 0] + 3) % 20
        m_ap2, ap2 = UT.cal_map(lg, tg2)
    save("g9_metrics", seed=np.array([99]), acc1=a1.numpy(), acc5=a5.numpy(), mAP=np.array([float(m_ap)]), ap=ap.numpy(),
         mAP_multi=np.array([float(m_ap2)]), ap_multi=ap2.numpy())

    save("g7_act_e2e", seed_weights=np.array([1007]), weights_sha256=np.frombuffer(bytes.fromhex(sha), dtype=np.uint8),
         policy_idx=pol_idx.numpy(), policy_argmax_gap=pol_gap, logits=logits.numpy(), last=last.numpy(), forced_idx=forced.numpy(),
         logits_forced=logits_f.numpy(), last_forced=last_f.numpy(), glancer_vec=fv.numpy(),
         keys=np.array(sorted(shapes)), )
    return sorted(shapes.items())


# ------------------------------------------------------------------ STH tree
def gen_sth(md):
    _enter_tree(STH)
    import models.resnet as R
    md.resnet50 = lambda pretrained=False, **k: R.resnet50(pretrained=False, **k)
    md.ResNet = R.ResNet
    import models.gfv_net as G
    import ops.temporal_shift as TS
    _mb = G.mobilenet_v2
    G.mobilenet_v2 = lambda n_class, pretrained=True: _mb(n_class=n_class, pretrained=False)

    # ---- G3: temporal shift
    x = torch.arange(2 * 8 * 16 * 3 * 3, dtype=torch.float32).view(16, 16, 3, 3)
    o = TS.TemporalShift.shift(x, 8, fold_div=8)
    x2 = torch.from_numpy(rnd((12, 64, 2, 2), 31))
    o2 = TS.TemporalShift.shift(x2, 4, fold_div=8)
    save("g3_temporal_shift", out_arange=o.numpy(), seed_in=np.array([31]), out_rand=o2.numpy())

    # ---- G7b: end-to-end STH (config 4: TSM-R50, Tg=Tf=8, P=128, B=2), continuous policy
    args = sth_args()
    model = G.GFV(args)
    model.focuser.net.base_model = torch.nn.Sequential(*list(model.focuser.net.base_model.children())[:-1])
    model.eval()
    model.focuser.policy.policy.eval()
    model.focuser.policy.policy_old.eval()
    shapes, sha = load_synth(model, 1007)
    pol_shapes = {k: tuple(v.shape) for k, v in model.focuser.policy.policy_old.state_dict().items()}
    pol_sd = synth.synth_state_dict({"policy." + k: s for k, s in pol_shapes.items()}, 1007)
    model.focuser.policy.policy_old.load_state_dict({k[len("policy."):]: torch.from_numpy(v) for k, v in pol_sd.items()})
    model.focuser.policy.policy.load_state_dict(model.focuser.policy.policy_old.state_dict())
    gl = torch.from_numpy(synth.synth_frames(2, 8, 224, seed=3))
    fo = torch.from_numpy(synth.synth_frames(2, 8, 224, seed=4)).view(2, 8, 3, 224, 224)
    torch.manual_seed(0)
    with torch.no_grad():
        fm, glog = model.glance(gl)
        pred, _base, patch = model.action_stage2(fo, fm, glog, 0, args, prev_local_patch=None, training=False)
        # the action the policy produced = where the patch came from; recover by matching is fragile, replay instead
        act = model.focuser.policy.policy_old.act(fm.view(2, -1, 7, 7), model.focuser.memory, True, False)
        forced = torch.tensor([[0.13, 0.92], [0.77, 0.31]])
        model.focuser.policy.select_action = lambda *a, **k: forced
        pred_f, _b2, patch_f = model.action_stage2(fo, fm, glog, 0, args, prev_local_patch=None, training=False)
        pred3, patch3 = model.action_stage3(fo, fm, glog, 0, args, prev_local_patch=None)
    px = pixel_margin(act.numpy(), 224, 128)
    assert px.min() >= PIXEL_MARGIN_MIN, "G7b: crop origin within %g px of a pixel boundary: pick another frame seed" % px.min()
    save("g7_sth_e2e", seed_weights=np.array([1007]),
         weights_sha256=np.frombuffer(bytes.fromhex(sha), dtype=np.uint8), policy_action=act.numpy(), policy_action_px_margin=px,
         logits=pred.numpy(), forced_action=forced.numpy(), logits_forced=pred_f.numpy(),
         logits_stage3_forced=pred3.numpy(), glancer_logit=glog.numpy(),
         patch_corner=patch[:, :, :, :4, :4].numpy(), patch_forced_corner=patch_f[:, :, :, :4, :4].numpy(),
         keys=np.array(sorted(shapes)), policy_keys=np.array(sorted(pol_sd)))
    return sorted(shapes.items()), sorted((k, tuple(v.shape)) for k, v in pol_sd.items())


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    md = _install_shims()
    act_shapes = gen_act()
    sth_shapes, sth_pol = gen_sth(md)
    # key/shape manifests let the build's modules be checked for state-dict compatibility
    with open(os.path.join(OUT, "state_dict_manifest.txt"), "w") as f:
        for tag, items in (("ACT", act_shapes), ("STH", sth_shapes), ("STH_POLICY", sth_pol)):
            for k, s in items:
                f.write("%s %s %s\n" % (tag, k, "x".join(str(int(d)) for d in s) or "scalar"))
    print("done")


if __name__ == "__main__":
    main()
