"""TSN wrapper of the local CNN -- host mirror of STH/models/tsn.py for the one configuration the
Something-Something drivers use (ResNet-50, RGB, TSM 'blockres', no temporal pooling).

``forward(input, no_reshape=True)`` (tsn.py:215-241) = TSM-ResNet-50 trunk -> (N, 2048), executed by
``adaf_resnet50_forward`` with the shift fused into conv1.  State-dict compatibility: the reference
wraps every Bottleneck conv1 in ``TemporalShift`` (key ``...conv1.net.weight``; with shift_place='block' the
whole Bottleneck: ``layer1.0.net.conv1.weight``) and
STH/evaluate.py:83 re-wraps the children in a Sequential to drop fc (keys ``base_model.4.0.conv1.net.
weight``); both spellings load and save here.
"""
import re

import torch
from torch import nn

from .resnet import ResNet
from .temporal_shift import make_temporal_shift

__all__ = ["TSN"]

_SEQ = {"conv1": "0", "bn1": "1", "layer1": "4", "layer2": "5", "layer3": "6", "layer4": "7"}
_SEQ_INV = {v: k for k, v in _SEQ.items()}


class TSN(nn.Module):
    def __init__(self, num_segments, modality="RGB", base_model="resnet50", new_length=None, crop_num=1,
                 partial_bn=True, print_spec=False, pretrain="imagenet", is_shift=False, shift_div=8,
                 shift_place="blockres", fc_lr5=False, temporal_pool=False, non_local=False):
        super().__init__()
        if modality != "RGB" or "resnet50" not in base_model or non_local or temporal_pool:
            raise NotImplementedError("adafocus_amd.TSN: RGB ResNet-50 without non-local / temporal pooling only")
        self.modality, self.num_segments, self.reshape = modality, num_segments, False
        self.is_shift, self.shift_div, self.shift_place = is_shift, shift_div, shift_place
        self.new_length = 1
        self._stripped = False
        object.__setattr__(self, "_in_init", True)
        self.base_model = ResNet(num_classes=1000)
        object.__setattr__(self, "_in_init", False)
        if is_shift:
            make_temporal_shift(self.base_model, num_segments, n_div=shift_div, place=shift_place)
        self._register_load_state_dict_pre_hook(self._canonicalise_keys)
        self._register_state_dict_hook(self._reference_keys)

    # ---- STH/evaluate.py:83 compatibility ------------------------------------------------
    def strip_fc(self):
        """Equivalent of `base_model = Sequential(*children[:-1])`: fc disappears from the state dict and
        the remaining keys take Sequential indices; the HIP trunk is untouched (it never used fc)."""
        self._stripped = True
        return self

    def __setattr__(self, name, value):
        if name == "base_model" and isinstance(value, nn.Sequential) and not getattr(self, "_in_init", True) \
                and "base_model" in self._modules:
            self.strip_fc()          # the unmodified driver line lands here
            return
        super().__setattr__(name, value)

    # ---- key translation -----------------------------------------------------------------
    def _canonicalise_keys(self, state_dict, prefix, *args):
        p = prefix + "base_model."
        for k in [k for k in state_dict if k.startswith(p)]:
            rest = k[len(p):].replace(".conv1.net.", ".conv1.")                      # shift_place = 'blockres': TemporalShift(conv1)
            rest = re.sub(r"^((?:layer)?\d\.\d+)\.net\.", r"\1.", rest)                 # shift_place = 'block': TemporalShift(Bottleneck)
            head, _, tail = rest.partition(".")
            if head in _SEQ_INV:
                rest = _SEQ_INV[head] + "." + tail
            if rest != k[len(p):]:
                state_dict[p + rest] = state_dict.pop(k)
        if self._stripped:           # a stripped checkpoint has no fc; keep ours
            for leaf in ("fc.weight", "fc.bias"):
                state_dict.setdefault(p + leaf, getattr(self.base_model.fc, leaf.split(".")[1]).detach())

    def _reference_keys(self, module, state_dict, prefix, local_metadata):
        p = prefix + "base_model."
        for k in [k for k in state_dict if k.startswith(p)]:
            rest = k[len(p):]
            if self.is_shift and self.shift_place == "block":
                rest = re.sub(r"^(layer\d\.\d+)\.", r"\1.net.", rest)
            elif self.is_shift:
                rest = re.sub(r"^(layer\d\.\d+\.conv1)\.", r"\1.net.", rest)
            if self._stripped:
                head, _, tail = rest.partition(".")
                if head == "fc":
                    del state_dict[k]
                    continue
                rest = _SEQ[head] + "." + tail
            if rest != k[len(p):]:
                state_dict[p + rest] = state_dict.pop(k)
        return state_dict

    # ---- reference surface ---------------------------------------------------------------
    def forward(self, input, no_reshape=False):
        if not no_reshape:
            input = input.view((-1, 3) + input.size()[-2:])
        return self.base_model.get_featvec(input).squeeze()

    def features_nhwc4(self, patches_nhwc4, out=None):
        return self.base_model.features_nhwc4(patches_nhwc4, out=out)

    def features_from_frames(self, frames, actions, patch_size, frames_per_action=1, out=None):
        return self.base_model.features_from_frames(frames, actions, patch_size, frames_per_action, out=out)

    def partialBN(self, enable):
        self._enable_pbn = enable

    @property
    def feature_dim(self):
        return 2048
