"""ActivityNet / FCVID / Mini-Kinetics model composition -- host-side mirror of
ACT/models/gfv_net.py for offline inference (`evaluate=true`, stage-3 branch of
ACT/main_dist.py:367-371).

Same class names, constructor arguments (`args` fields of ACT/conf/default.yaml), attributes
(``glancer``, ``focuser``, ``focuser.net``, ``focuser.policy.policy/policy_old``, ``classifier``) and
state-dict keys as the reference, so its checkpoints load unchanged.  What differs is the
execution plan of ``GFV.forward(one_step=True)`` (gfv_net.py:95-133):

  reference: glancer; for t in range(T): policy step -> python-loop crop (4 .item() syncs per sample) ->
             ResNet-50 on B patches -> cat;  then GRU
  here:      glancer on adaf_mobilenetv2 -> policy over all T on the engine (its input never depends
             on local features in eval mode) ->
             ONE batched HIP gather of B*T patches (NHWC4) -> ONE ResNet-50 trunk pass over B*T
             patches whose avgpool writes straight into the GRU input matrix -> HIP GRU + FC

which reproduces the reference logits.  Training branches (stage 0-2 forward
modes, PPO update) are out of scope and raise.  ``GFV.one_step_act(training=False)`` -- the body of the
stage-2 VALIDATION loop (ACT/main_dist.py:346-362), reward baseline included -- keeps the reference's
per-step structure on the same HIP ops (round 6, pinned by G15).
"""
import math

import numpy as np
import torch
from torch import nn

from . import hip_ops
from .mobilenet import mobilenet_v2
from .ppo import PPO, Memory
from .resnet import resnet50
from .synth import grid_table
from .utils import get_patch, get_patch_nhwc4

__all__ = ["GFV", "Glancer", "Focuser", "PatchSampler", "RecurrentClassifier", "LinearCLassifier"]


class GFV(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.num_segments = args.num_segments
        self.num_class = args.num_classes
        self.rew = getattr(args, "reward", "random")
        if getattr(args, "dataset", None) == "fcvid":
            assert args.num_classes == 239
        self.input_size = args.input_size
        self.batch_size = args.batch_size
        self.patch_size = args.patch_size
        self.input_mean = [0.485, 0.456, 0.406]
        self.input_std = [0.229, 0.224, 0.225]
        self.with_glancer = args.with_glancer
        self.glance_size = getattr(args, "glance_size", args.input_size)
        self._pipe = None                      # (front stream, back stream) of offline_forward_pipelined
        self.glancer = Glancer(num_classes=self.num_class)
        cells = math.ceil(args.glance_size / 32)
        policy_params = dict(feature_dim=args.feature_map_channels, state_dim=args.feature_map_channels * cells * cells,
                             action_dim=args.action_dim, hidden_state_dim=args.hidden_state_dim,
                             policy_conv=args.policy_conv, gpu=args.gpu, continuous=getattr(args, "continuous", False),
                             gamma=getattr(args, "gamma", 0.7), policy_lr=getattr(args, "policy_lr", 0.0003))
        # build-specific: local_arch = "efficientnet-b3" (+ local_dtype "f16" | "f32", local_image_size = the resolution the
        # package's static SAME padding is computed for: "native" like EfficientNet.from_name, an int, or None = the input's own
        # size) selects BASELINE config 5's local CNN (no reference implementation, parity unpinned -- adafocus_amd/mbconv_local.py);
        # default = the reference's ResNet-50
        self.focuser = Focuser(args.patch_size, args.random_patch, policy_params, self.num_class,
                               local_arch=getattr(args, "local_arch", "resnet50"), local_dtype=getattr(args, "local_dtype", "f16"),
                               local_image_size=getattr(args, "local_image_size", "native"))
        self.dropout = nn.Dropout(p=args.dropout)
        feat_dim = self.focuser.feature_dim + (self.glancer.feature_dim if self.with_glancer else 0)
        if args.consensus == "gru":
            self.classifier = RecurrentClassifier(seq_len=args.num_segments, input_dim=feat_dim,
                                                  batch_size=self.batch_size, hidden_dim=args.hidden_dim,
                                                  num_classes=args.num_classes, dropout=args.dropout)
        elif args.consensus == "fc":
            self.classifier = LinearCLassifier(seq_len=args.num_segments, input_dim=feat_dim,
                                               batch_size=self.batch_size, hidden_dim=args.hidden_dim,
                                               num_classes=args.num_classes, dropout=args.dropout)
        else:
            raise ValueError("consensus must be 'gru' or 'fc'")

    # ---- reference surface --------------------------------------------------------------
    def forward(self, *argv, **kwargs):
        if kwargs.get("backbone_pred"):
            x = kwargs["input"]
            b, tc, hh, ww = x.shape
            x2 = x.view(b * (tc // 3), 3, hh, ww)
            net = self.glancer if kwargs.get("glancer") else self.focuser
            return net.predict(x2).view(b, tc // 3, -1)
        if kwargs.get("training"):
            raise NotImplementedError("only training=False (offline inference / validation) is implemented")
        if not kwargs.get("one_step"):
            # the stage-1 form (gfv_net.py:135-150) in eval mode, as validate() runs it at train_stage 1 (ACT/main_dist.py:334-340): glancer and
            # focuser over all B*T frames at once -- random crops when the model was built with random_patch (the stage-1 configuration),
            # else ONE policy step over the B*T frames as a batch, exactly as the reference's call does -- then the classifier
            x, scan = kwargs["input"], kwargs["scan"]
            b, tc, hh, ww = x.shape
            t = tc // 3
            with torch.no_grad():
                fmap, fvec = self.glancer(scan.reshape(b * t, 3, scan.shape[2], scan.shape[3]))
                local = self.focuser(input=x.reshape(b * t, 3, hh, ww), state=fmap, restart_batch=True, training=False)[0].view(b * t, -1)
                feature = torch.cat([fvec, local], dim=1) if self.with_glancer else local
                return self.classifier(feature.view(b, t, -1))
        if self.focuser.random:
            return None          # (gfv_net.py:108: the one_step form only has a body for a policy-driven focuser)
        return self.offline_forward(kwargs["input"], kwargs["scan"])[:2]

    @torch.no_grad()
    def offline_forward(self, images, scan, forced_action_idx=None):
        """images, scan: (B, T*3, H, W) fp32 on the GPU.  Returns (logits (B*T,C), last (B,C),
        feature matrix (B,T,F), action indices (B,T))."""
        b, tc, hh, ww = images.shape
        t = tc // 3
        # glancer (adaf_mobilenetv2) -> pixel-major map + 1280-d vectors; policy over all T steps at once
        fmap, fvec = self.glancer.net.features_nhwc(scan.reshape(b * t, 3, scan.shape[2], scan.shape[3]))
        table = self.focuser.action_table(images.device)
        idx, actions = self.focuser.policy.policy_old.act_sequence_nhwc(fmap, b, t, table)
        if forced_action_idx is not None:
            idx = forced_action_idx.to(idx.device)
            actions = table[idx.reshape(-1)]
        return self.hot_path(images.view(b * t, 3, hh, ww), fvec.view(b, t, -1) if self.with_glancer else None, actions,
                             b, t) + (idx,)

    def glancer_input(self, images):
        """`input_prime = F.interpolate(images, (glance_size, glance_size))` of the reference's drivers (nearest mode,
        ACT/main_dist.py:331-332): the identity when glance_size equals the frame size (every shipped config), else one
        bit-exact resize launch.  images (B, T*3, H, W) planar or (B*T, H, W, 4) pixel-major; same layout back."""
        if images.shape[-1] == 4 and images.shape[1] != 3:
            if images.shape[1] == self.glance_size:
                return images
            return hip_ops.resize_nearest(images, self.glance_size, hip_ops.LAYOUT_NHWC4)
        if images.shape[-1] == self.glance_size:
            return images
        b, tc, hh, ww = images.shape
        g = self.glance_size
        return hip_ops.resize_nearest(images.reshape(b * tc, 1, hh, ww), g, hip_ops.LAYOUT_NCHW).view(b, tc, g, g)

    @torch.no_grad()
    def offline_forward_nhwc4(self, frames_nhwc4, b, t, forced_action_idx=None):
        """Same as offline_forward for frames that are already normalised pixel-major (B*T,H,W,4)
        (``transforms.ingest_uint8``): the glancer and the gather read them directly."""
        fmap, fvec = self.glancer.net.features_from_nhwc4(self.glancer_input(frames_nhwc4))
        table = self.focuser.action_table(frames_nhwc4.device)
        idx, actions = self.focuser.policy.policy_old.act_sequence_nhwc(fmap, b, t, table)
        if forced_action_idx is not None:
            idx = forced_action_idx.to(idx.device)
            actions = table[idx.reshape(-1)]
        return self.hot_path(frames_nhwc4, fvec.view(b, t, -1) if self.with_glancer else None, actions, b, t) + (idx,)

    @torch.no_grad()
    def offline_forward_pipelined(self, clips, t, forced_action_idx=None):
        """One batch of the full forward with the two halves on the model's own streams, so consecutive calls overlap BY
        CONSTRUCTION: ingest + glancer + policy of batch i+1 (HBM-bound depthwise work) on the front stream while the hot
        path of batch i (MFMA-bound trunk) runs on the back stream; an event hands the frames / glancer vectors / actions
        over.  clips: the loader's stacked uint8 clips (B, H, W, T*3) or normalised pixel-major frames (B*T, H, W, 4).
        Returns (logits, last, idx, done, handoff) -- HIP events: the outputs belong to the back stream until the
        consumer's stream waits on `done` (``torch.cuda.current_stream().wait_event(done)``) or `pipeline_flush()` is
        called.  `handoff` is the RELEASE event of `clips`: the caller may overwrite the buffer once it has passed.  For
        uint8 clips that is when the front half has consumed them (ingest makes a fresh frame tensor); for normalised
        float frames the back stream's gather reads the caller's buffer itself, so `handoff` is `done` (a slot
        reused at the front-half event raced the crop)."""
        dev = clips.device
        if self._pipe is None or self._pipe[0].device != dev:
            self._pipe = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
        front, back = self._pipe
        cur = torch.cuda.current_stream(dev)
        front.wait_stream(cur)                       # the caller produced `clips` on its stream
        with torch.cuda.stream(front):
            if clips.dtype == torch.uint8:
                from .transforms import ingest_uint8
                b = clips.shape[0]
                frames = ingest_uint8(clips, t)
            else:
                frames = clips
                b = frames.shape[0] // t
            clips.record_stream(front)
            fmap, fvec = self.glancer.net.features_from_nhwc4(self.glancer_input(frames))
            table = self.focuser.action_table(dev)
            idx, actions = self.focuser.policy.policy_old.act_sequence_nhwc(fmap, b, t, table)
            if forced_action_idx is not None:
                idx = forced_action_idx.to(dev)
                actions = table[idx.reshape(-1)]
            handoff = torch.cuda.Event()
            handoff.record(front)
        back.wait_event(handoff)
        with torch.cuda.stream(back):
            for x in (frames, fvec, actions):
                x.record_stream(back)                # allocated on the front stream, read here
            logits, last, _ = self.hot_path(frames, fvec.view(b, t, -1) if self.with_glancer else None, actions, b, t)
            done = torch.cuda.Event()
            done.record(back)
            if clips.dtype != torch.uint8:
                clips.record_stream(back)
        for x in (logits, last, idx):
            x.record_stream(cur)
        return logits, last, idx, done, (handoff if clips.dtype == torch.uint8 else done)

    def pipeline_flush(self):
        """Make the current stream wait for everything offline_forward_pipelined has enqueued."""
        if self._pipe is not None:
            cur = torch.cuda.current_stream(self._pipe[0].device)
            cur.wait_stream(self._pipe[0])
            cur.wait_stream(self._pipe[1])

    def hot_path(self, frames, global_feat, actions, b, t):
        """Batched crop -> local CNN -> concat -> classifier: the benchmarked slice.
        frames (B*T,3,H,W) [reference layout] or (B*T,H,W,4) [pixel-major], global_feat (B,T,1280) or
        None, actions (B*T,2)."""
        gdim = global_feat.shape[2] if global_feat is not None else 0
        feature = torch.empty((b, t, gdim + self.focuser.feature_dim), device=frames.device, dtype=torch.float32)
        flat = feature.view(b * t, -1)
        net = self.focuser.net
        if hasattr(net, "features_from_frames"):
            # the gather rides in the trunk's first launch (adaf_resnet50_forward_frames): no patch tensor between get_patch and the local CNN
            net.features_from_frames(frames, actions, self.patch_size, out=flat[:, gdim:])
        else:
            if frames.shape[-1] == 4 and frames.shape[1] != 3:
                patches = hip_ops.crop_gather_nhwc4(frames, actions, self.patch_size)
            else:
                patches = get_patch_nhwc4(frames, actions, self.patch_size)
            net.features_nhwc4(patches, out=flat[:, gdim:])
        if gdim:
            hip_ops.copy2d(global_feat.reshape(b * t, gdim), flat[:, :gdim])
        logits, last = self.classifier(feature)
        return logits, last, feature

    def capture_hot_path(self, b, t, frame_shape=None, exclusive=False, check_every=16):
        """Latency mode for small batches (BASELINE config 1 is B = 2; the reference's published CPU figure is a bs = 1
        latency): one hot-path step for a FIXED (B, T) captured into a HIP graph, so its ~50 dependent, nearly empty
        launches are replayed back to back by the runtime instead of being issued one by one from Python.  Returns a
        HotPathGraph; call it with (frames, glancer vectors, actions).
        The persistent GRU scan's grid barrier needs its blocks co-resident, which the library guarantees for eager launches by
        throttling scans over `scan_slots` events -- events that cannot be part of a capture.  So by default a capture takes the
        GRU's launch-per-step form (no grid barrier: any number of graphs may replay side by side, beside eager hot paths; same
        arithmetic as hot_path() under hip_ops.set_gru_persistent(0), bit for bit).  `exclusive=True` keeps the persistent scan in the
        graph (bit-identical to the default eager hot_path()): the caller then promises that nothing else runs GRU scans on the device
        while the graph replays, and the graph checks the barrier's time-out counter every `check_every` replays (a device
        synchronisation; 0 = never) and raises AdafError instead of handing out NaN-poisoned logits."""
        return HotPathGraph(self, b, t, frame_shape, exclusive=exclusive, check_every=check_every)

    def glance(self, input_prime):
        """Reference layout: (featmap (B,T,1280,h,w) [a permuted view of the pixel-major map], vec (B,T,1280))."""
        b, tc, hh, ww = input_prime.shape
        t = tc // 3
        fm, fv = self.glancer(input_prime.reshape(b * t, 3, hh, ww))
        return fm.unflatten(0, (b, t)), fv.view(b, t, -1)

    @torch.no_grad()
    def one_step_act(self, img, global_feat_map, global_feat, restart_batch=False, training=True):
        """One step of the stage-2 loop (gfv_net.py:160-210).  training=False is the VALIDATION branch of ACT/main_dist.py:346-362:
        policy step -> crop -> local CNN -> concat with the glancer vector -> one GRU + FC step, plus the reward baseline's logits from
        the classifier's CURRENT state (`test_single_forward`, gfv_net.py:448-457) -- for reward = 'random' a random crop per clip
        (`Focuser.random_patching`; origins drawn like utils.py:31-32, from numpy's global generator), for 'padding' | 'prev' | 'conf'
        zeros in place of the local feature.  Returns (logits (B,C), last_out (B,C), None, standard action (B,2), baseline logits (B,C)).
        training=True is the PPO roll-out of stage-2 TRAINING (memory of log-probabilities, sampled actions): out of scope."""
        if training:
            raise NotImplementedError("one_step_act(training=True) is the stage-2 (PPO) training loop body: out of scope")
        b = img.shape[0]
        local_feat, pack = self.focuser(input=img, state=global_feat_map, restart_batch=restart_batch, training=False)
        patch_size_list, action_list = pack if pack is not None else (None, None)
        local_feat = local_feat.view(b, -1)
        if self.rew == "random":
            base_local = self.focuser.random_patching(img)[0].view(b, -1)
        elif self.rew in ("padding", "prev", "conf"):
            base_local = torch.zeros_like(local_feat)
        else:
            raise NotImplementedError("reward %r" % (self.rew,))
        if self.with_glancer:
            feature = torch.cat([global_feat, local_feat], dim=1)
            baseline_feature = torch.cat([global_feat, base_local], dim=1)
        else:
            feature, baseline_feature = local_feat, base_local
        baseline_logits, _ = self.classifier.test_single_forward(baseline_feature.unsqueeze(1), reset=restart_batch)
        logits, last_out = self.classifier.single_forward(feature.unsqueeze(1), reset=restart_batch)
        return logits, last_out, patch_size_list, action_list, baseline_logits

    def train_mode(self, args):
        raise NotImplementedError("training modes are out of scope; use .eval()")

    @property
    def scale_size(self):
        return self.input_size * 256 // 224

    @property
    def crop_size(self):
        return self.input_size


class HotPathGraph:
    """A captured hot-path step (GFV.capture_hot_path).  The graph reads the static buffers `frames` (B*T,3,H,W),
    `gvec` (B,T,1280) and `actions` (B*T,2) and writes `logits` (B*T,C) / `last` (B,C); __call__ copies its arguments into
    the static inputs (or write them in place and call replay()).  One graph = one stream: replay it from the stream the
    results are consumed on.  See GFV.capture_hot_path for `exclusive` / `check_every`."""

    def __init__(self, model, b, t, frame_shape=None, exclusive=False, check_every=16):
        from . import _lib, hip_ops
        dev = next(model.parameters()).device
        hh = model.input_size
        self.b, self.t = b, t
        self.device = dev
        self.exclusive, self.check_every = bool(exclusive), int(check_every)
        self._replays = 0
        self.frames = torch.zeros(tuple(frame_shape) if frame_shape else (b * t, 3, hh, hh), device=dev)
        self.gvec = torch.zeros((b, t, model.glancer.feature_dim), device=dev) if model.with_glancer else None
        self.actions = torch.zeros((b * t, 2), device=dev)
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(cur)
        with torch.no_grad(), torch.cuda.stream(side):      # lazy initialisation (weight packing, scratch) stays out of the capture
            for _ in range(2):
                model.hot_path(self.frames, self.gvec, self.actions, b, t)
        cur.wait_stream(side)
        torch.cuda.synchronize(dev)
        self._timeouts0 = hip_ops.gru_scan_timeouts(dev) if self.exclusive else 0
        self.graph = torch.cuda.CUDAGraph()
        # captured on a stream of its OWN: the engines key their scratch (trunk workspace, GRU buffers) by the stream they are called on, and
        # torch's default capture stream is one per process -- two graphs captured on it would share scratch and race when replayed side by side
        self._capture_stream = torch.cuda.Stream(device=dev)
        with _lib.option("gru_graph_persistent", 1 if self.exclusive else 0):
            with torch.no_grad(), torch.cuda.graph(self.graph, stream=self._capture_stream):
                self.logits, self.last, self.feature = model.hot_path(self.frames, self.gvec, self.actions, b, t)

    def check(self):
        """exclusive graphs: synchronise and raise if a persistent scan's grid barrier has timed out since the capture (its logits are
        NaN-poisoned: something else ran GRU scans beside this graph)."""
        from . import _lib, hip_ops
        if self.exclusive:
            n = hip_ops.gru_scan_timeouts(self.device)
            if n != self._timeouts0:
                self._timeouts0 = n
                raise _lib.AdafError("HotPathGraph: %d block(s) of a persistent GRU scan timed out at their grid barrier -- an `exclusive` graph "
                                     "was replayed beside other GRU scans; the affected logits are NaN-poisoned" % n)

    def replay(self):
        self.graph.replay()
        if self.exclusive and self.check_every > 0:
            self._replays += 1
            if self._replays % self.check_every == 0:
                self.check()
        return self.logits, self.last

    def __call__(self, frames, gvec, actions):
        self.frames.copy_(frames.view_as(self.frames))
        if self.gvec is not None:
            self.gvec.copy_(gvec)
        self.actions.copy_(actions)
        return self.replay()


class Glancer(nn.Module):
    def __init__(self, skip=False, num_classes=200):
        super().__init__()
        self.net = mobilenet_v2(pretrained=False)
        self.net.classifier = nn.Sequential(nn.Dropout(0.2), nn.Linear(self.net.last_channel, num_classes))
        self.skip = skip

    def forward(self, input):
        return self.net.get_featmap(input)

    def predict(self, input):
        return self.net(input)

    @property
    def feature_dim(self):
        return self.net.feature_dim


class Focuser(nn.Module):
    def __init__(self, size=96, random=True, policy_params=None, num_classes=200, local_arch="resnet50", local_dtype="f16",
                 local_image_size="native"):
        super().__init__()
        if local_arch == "resnet50":
            self.net = resnet50(pretrained=False)
            self.net.fc = nn.Linear(self.net.fc.in_features, num_classes)
        elif local_arch.startswith("efficientnet-b"):
            from .mbconv_local import EfficientNetLocalCNN
            self.net = EfficientNetLocalCNN(local_arch, num_classes=num_classes, dtype=local_dtype, image_size=local_image_size)
        else:
            raise ValueError("local_arch must be 'resnet50' or 'efficientnet-b<N>'")
        self.patch_size = size
        self.random = random
        self.patch_sampler = PatchSampler(self.patch_size, self.random)
        self.policy = None
        self.memory = Memory()
        if not self.random:
            assert policy_params is not None
            # s x s grids of [row/(s-1), col/(s-1)], python doubles -> fp32 (gfv_net.py:272-307)
            self._tables = {s * s: torch.from_numpy(grid_table(s)) for s in (5, 6, 7, 8)}
            self.policy_feature_dim = policy_params["feature_dim"]
            self.policy_state_dim = policy_params["state_dim"]
            self.policy_action_dim = policy_params["action_dim"]
            self.policy_hidden_state_dim = policy_params["hidden_state_dim"]
            self.policy_conv = policy_params["policy_conv"]
            self.gpu = policy_params["gpu"]
            self.policy = PPO(self.policy_feature_dim, self.policy_state_dim, self.policy_action_dim,
                              self.policy_hidden_state_dim, self.policy_conv, self.gpu, gamma=policy_params["gamma"],
                              lr=policy_params["policy_lr"])

    @property
    def standard_actions_set(self):
        return self._tables

    def action_table(self, device):
        table = self._tables[self.policy_action_dim]
        if table.device != torch.device(device):
            table = self._tables[self.policy_action_dim] = table.to(device)
        return table

    def _get_standard_action(self, action):
        return self.action_table(action.device)[action], None

    def forward(self, *argv, **kwargs):
        """One focuser step with the reference's contract (gfv_net.py:316-331): returns
        (local feature (B,2048,1,1), (None, standard_action))."""
        if self.random:           # gfv_net.py:317-327: random crops, no policy, no action pack
            return self.random_patching(kwargs["input"])
        action = self.policy.select_action(kwargs["state"], self.memory, kwargs["restart_batch"], kwargs["training"])
        standard_action, _ = self._get_standard_action(action)
        imgs = kwargs["input"]
        feat = self.net.features_nhwc4(get_patch_nhwc4(imgs, standard_action, self.patch_size))
        return feat.view(imgs.shape[0], -1, 1, 1), (None, standard_action)

    def random_patching(self, imgs):
        """gfv_net.py:334-336: the local CNN's feature of one random crop per image -- the reward baseline of the stage-2 loop and the focuser of a
        random_patch model.  Origins from `PatchSampler.random_actions` (the reference's draw order); the crop itself is the batched HIP gather."""
        action = self.patch_sampler.random_actions(imgs)
        feat = self.net.features_nhwc4(get_patch_nhwc4(imgs, action, self.patch_size))
        return feat.view(imgs.shape[0], -1, 1, 1), None

    def predict(self, input):
        return self.net(input)

    def update(self):
        raise NotImplementedError("policy update is training code")

    @property
    def feature_dim(self):
        return self.net.feature_dim


class PatchSampler(nn.Module):
    def __init__(self, size=96, random=True):
        super().__init__()
        self.random = random
        self.size = size

    def sample(self, imgs, action=None):
        if self.random:
            return self.random_sample(imgs)
        assert action is not None
        return get_patch(imgs, action, self.size)

    def random_actions(self, imgs):
        """One crop origin per image drawn exactly like utils.py:24-35 (`np.random.randint(0, H - P)` for y, then for x, image by image; no draw
        at H == P), returned as gather actions (origin + 0.5) / (H - P): floor(a * (H - P)) (utils.py:42) lands on the drawn integer whatever
        the rounding.  A seeded numpy generator therefore reproduces the reference's crops."""
        n, hh, ww = imgs.shape[0], imgs.shape[2], imgs.shape[3]
        act = np.zeros((n, 2), dtype=np.float64)
        if hh != self.size:
            if hh != ww:
                raise ValueError("random crops: square frames expected (the gather scales both axes by H - P, utils.py:40-42)")
            for i in range(n):
                act[i, 0] = np.random.randint(0, hh - self.size)
                act[i, 1] = np.random.randint(0, ww - self.size)
            act = (act + 0.5) / (hh - self.size)
        return torch.from_numpy(act.astype(np.float32)).to(imgs.device)

    def random_sample(self, imgs):
        """gfv_net.py:376-381: a crop at a random position per image."""
        return get_patch(imgs, self.random_actions(imgs), self.size)

    def forward(self, *argv, **kwargs):
        raise NotImplementedError


class LinearCLassifier(nn.Module):
    """softmax-mean classifier (gfv_net.py:388-407); FC on the HIP engine, softmax/mean in torch."""

    def __init__(self, seq_len, input_dim, batch_size, hidden_dim, num_classes, dropout):
        super().__init__()
        self.seq_len, self.input_dim, self.hidden_dim, self.num_classes, self.batch_size = \
            seq_len, input_dim, hidden_dim, num_classes, batch_size
        self.fc = nn.Linear(input_dim, num_classes)
        self.dropout = nn.Dropout(dropout)

    def forward(self, feature):
        b, t, _ = feature.shape
        logits = hip_ops.linear(feature.reshape(b * t, -1), self.fc.weight.detach(), self.fc.bias.detach())
        avg = torch.softmax(logits, dim=1).reshape(b, t, -1).mean(dim=1)
        return torch.log(avg), avg


class RecurrentClassifier(nn.Module):
    """GRU + FC classifier (gfv_net.py:409-435).  ``self.gru`` / ``self.fc`` hold the parameters under
    the reference's names; the arithmetic is adaf_gru_cls_forward_f32."""

    def __init__(self, seq_len, input_dim, batch_size, hidden_dim, num_classes, dropout, bias=True):
        super().__init__()
        self.seq_len, self.input_dim, self.hidden_dim, self.num_classes, self.batch_size = \
            seq_len, input_dim, hidden_dim, num_classes, batch_size
        self.gru = nn.GRU(input_size=input_dim, hidden_size=hidden_dim, bias=bias, batch_first=True)
        self.fc = nn.Linear(hidden_dim, num_classes)
        self.dropout = nn.Dropout(dropout)

    def forward(self, feature):
        if self.training:
            raise RuntimeError("RecurrentClassifier: eval mode only (dropout must be the identity)")
        g = self.gru
        return hip_ops.gru_cls_forward(feature, g.weight_ih_l0.detach(), g.weight_hh_l0.detach(), g.bias_ih_l0.detach(),
                                       g.bias_hh_l0.detach(), self.fc.weight.detach(), self.fc.bias.detach())

    def _steps_from(self, feature, hx):
        """GRU over feature (B,t,F) from the hidden state hx ((1,B,H) or None = zeros) + FC on every step: (logits (B*t,C), last (B,C),
        final hidden (1,B,H))."""
        if self.training:
            raise RuntimeError("RecurrentClassifier: eval mode only (dropout must be the identity)")
        g = self.gru
        b, t, _ = feature.shape
        hs = hip_ops.gru_seq_forward(feature, g.weight_ih_l0.detach(), g.weight_hh_l0.detach(), g.bias_ih_l0.detach(), g.bias_hh_l0.detach(),
                                     h0=None if hx is None else hx[0])
        logits = hip_ops.linear(hs.reshape(b * t, -1), self.fc.weight.detach(), self.fc.bias.detach())
        return logits, logits.view(b, t, -1)[:, -1, :].reshape(b, -1), hs[:, -1].unsqueeze(0).contiguous()

    def _state(self, reset, b):
        if reset:
            self.hx = None                       # (the reference stores explicit zeros, gfv_net.py:439-440; None = the scan's zero state)
        elif not hasattr(self, "hx"):
            raise RuntimeError("RecurrentClassifier: no hidden state yet -- the first step needs reset=True")
        elif self.hx is not None and self.hx.shape[1] != b:
            raise RuntimeError("RecurrentClassifier: batch %d after a state of batch %d without reset" % (b, self.hx.shape[1]))
        return self.hx

    def single_forward(self, feature, reset=False, gpu=0):
        """gfv_net.py:437-446: advance the stored hidden state `hx` over feature (B,t,F)."""
        logits, last, self.hx = self._steps_from(feature, self._state(reset, feature.shape[0]))
        return logits, last

    def test_single_forward(self, feature, reset=False, gpu=0):
        """gfv_net.py:448-457: the same step(s) from the stored state WITHOUT storing the result (the reward baseline)."""
        hx = self._state(reset, feature.shape[0])
        logits, last, _ = self._steps_from(feature, hx)
        return logits, last
