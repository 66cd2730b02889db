"""EfficientNet-b5 trunk for the `--backbone eff_b5` configuration (BASELINE.json configs[3]).

The reference obtains it from torch.hub — `hub.load('rwightman/gen-efficientnet-pytorch', 'tf_efficientnet_b5_ap')`
(reference networks/base_encoder.py:90-94) — a third-party package that is not part of the reference tree.  This module
restates that architecture from its public definition (EfficientNet-B0 stage table scaled by width 1.6 / depth 2.2, TensorFlow
"SAME" padding, BatchNorm eps 1e-3, swish, squeeze-and-excite with a quarter of the block's INPUT channels) under
gen-efficientnet's module names, so its state dict lines up key by key with a hub checkpoint:
    conv_stem, bn1, blocks.{stage}.{i}.{conv_pw,bn1,conv_dw,bn2,se.conv_reduce,se.conv_expand,conv_pwl,bn3}
    (stage 0: conv_dw, bn1, se.*, conv_pw, bn2), conv_head, bn2
Parity of the trunk arithmetic against the hub package is UNPINNED (it cannot be imported here); the oracle holds the same
restatement in plain torch (oracle/torch_ref.py EfficientNetB5), and the device kernels are tested against it.
Every operator runs on libsqd kernels except the 3-channel 3x3/2 stem convolution (ATen)."""
import math

import torch.nn as nn

from sqd import nnops as X

BN_EPS = 1e-3
#        kind  kernel stride expand  out  repeats     (EfficientNet-B0 table x width 1.6 (multiples of 8) / depth 2.2 (ceil))
B5_STAGES = (("ds", 3, 1, 1, 24, 3), ("ir", 3, 2, 6, 40, 5), ("ir", 5, 2, 6, 64, 5), ("ir", 3, 2, 6, 128, 7),
             ("ir", 5, 1, 6, 176, 7), ("ir", 5, 2, 6, 304, 9), ("ir", 3, 1, 6, 512, 3))
B5_STEM, B5_HEAD = 48, 2048


def _se_channels(in_chs, ratio=0.25):
    return max(1, int(in_chs * ratio + 0.5))


class SqueezeExcite(nn.Module):
    def __init__(self, chs, reduced):
        super().__init__()
        self.conv_reduce = nn.Conv2d(chs, reduced, 1, bias=True)
        self.conv_expand = nn.Conv2d(reduced, chs, 1, bias=True)

    def forward(self, x):
        return X.squeeze_excite(x, self.conv_reduce, self.conv_expand)


class DepthwiseSeparableConv(nn.Module):
    """stage 0: depthwise k x k -> BN + swish -> SE -> 1x1 -> BN (+ input when the shapes match)"""

    def __init__(self, in_chs, out_chs, k, stride):
        super().__init__()
        self.has_residual = stride == 1 and in_chs == out_chs
        self.stride = stride
        self.conv_dw = nn.Conv2d(in_chs, in_chs, k, stride, 0, groups=in_chs, bias=False)
        self.bn1 = nn.BatchNorm2d(in_chs, eps=BN_EPS)
        self.se = SqueezeExcite(in_chs, _se_channels(in_chs))
        self.conv_pw = nn.Conv2d(in_chs, out_chs, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(out_chs, eps=BN_EPS)

    def forward(self, x):
        y = X.dw_conv_bn_act(x, self.conv_dw, self.bn1, "swish", self.stride, pool=True)      # (pooled sums for the gate on the way)
        y = self.se(y)
        return X.conv_bn_act(y, self.conv_pw, self.bn2, None, residual=x if self.has_residual else None)


class InvertedResidual(nn.Module):
    """MBConv: 1x1 expansion -> BN + swish -> depthwise k x k -> BN + swish -> SE -> 1x1 projection -> BN (+ input)"""

    def __init__(self, in_chs, out_chs, k, stride, expand):
        super().__init__()
        mid = in_chs * expand
        self.has_residual = stride == 1 and in_chs == out_chs
        self.stride = stride
        self.conv_pw = nn.Conv2d(in_chs, mid, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(mid, eps=BN_EPS)
        self.conv_dw = nn.Conv2d(mid, mid, k, stride, 0, groups=mid, bias=False)
        self.bn2 = nn.BatchNorm2d(mid, eps=BN_EPS)
        self.se = SqueezeExcite(mid, _se_channels(in_chs))
        self.conv_pwl = nn.Conv2d(mid, out_chs, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(out_chs, eps=BN_EPS)

    def forward(self, x):
        if self.has_residual:
            # x has two consumers: the expansion hands it through, so that the residual branch's gradient is added in the expansion's
            # data-gradient epilogue (no accumulation pass over [B,C,H,W]) and — the sum being the complete gradient of the previous
            # block's bn3 output — that BatchNorm's backward sums come from the same epilogue (nnops.conv_bn_act, skip=True)
            y, x = X.conv_bn_act(x, self.conv_pw, self.bn1, "swish", skip=True)
        else:
            y = X.conv_bn_act(x, self.conv_pw, self.bn1, "swish")
        y = X.dw_conv_bn_act(y, self.conv_dw, self.bn2, "swish", self.stride, pool=True)      # (pooled sums for the gate on the way)
        y = self.se(y)
        return X.conv_bn_act(y, self.conv_pwl, self.bn3, None, residual=x if self.has_residual else None)


class GenEfficientNet(nn.Module):
    """tf_efficientnet_b5_ap with global_pool and classifier removed (reference base_encoder.py:99-100); forward returns the
    feature list reference Encoder.forward builds (base_encoder.py:63-73) — entries the decoder never reads are None."""

    def __init__(self, stages=B5_STAGES, stem=B5_STEM, head=B5_HEAD):
        """head=None: no conv_head / bn2 (the features_only trunk of `EfficientNetFeatures`)"""
        super().__init__()
        self.conv_stem = nn.Conv2d(3, stem, 3, 2, 0, bias=False)
        self.bn1 = nn.BatchNorm2d(stem, eps=BN_EPS)
        blocks, cin = [], stem
        for kind, k, stride, expand, cout, repeats in stages:
            stage = []
            for i in range(repeats):
                s = stride if i == 0 else 1
                stage.append(DepthwiseSeparableConv(cin, cout, k, s) if kind == "ds" else InvertedResidual(cin, cout, k, s, expand))
                cin = cout
            blocks.append(nn.Sequential(*stage))
        self.blocks = nn.Sequential(*blocks)
        if head is not None:
            self.conv_head = nn.Conv2d(cin, head, 1, bias=False)
            self.bn2 = nn.BatchNorm2d(head, eps=BN_EPS)
            self.global_pool = nn.Identity()
            self.classifier = nn.Identity()
        for m in self.modules():                     # (gen-efficientnet's initialisation: fan-out normal for the convolutions)
            if isinstance(m, nn.Conv2d):
                fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
                nn.init.normal_(m.weight, 0.0, math.sqrt(2.0 / fan_out))
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def forward(self, x):
        feats = [x, None, None, X.stem_same_conv_bn_act(x, self.conv_stem, self.bn1, "swish")]      # conv_stem, bn1, act1
        for stage in self.blocks:
            feats.append(stage(feats[-1]))
        if not hasattr(self, "conv_head"):
            return feats
        feats.append(X.conv2d(feats[-1], self.conv_head))                                           # features[11]
        feats += [None, None, None, None]                    # bn2, act2, global_pool, classifier: never read by the decoder
        return feats


class EfficientNetFeatures(GenEfficientNet):
    """`timm.create_model('tf_efficientnet_b5_ap', features_only=True)` as the reference's `Unet` builds it for
    `--backbone tf_efficientnet_b5_ap` (reference trainer.py:64, networks/Unet.py:114-118; args_files/hisfog/kitti/effb5_320x1024.txt):
    the trunk without conv_head, returning the last feature map of every stride — stages 0, 1, 2, 4, 6 = 24 @ /2, 40 @ /4, 64 @ /8,
    176 @ /16, 512 @ /32 (timm's default out_indices for this family).  timm is not part of the reference tree: restated, parity of
    the trunk arithmetic UNPINNED like the hub trunk above; state-dict keys conv_stem, bn1, blocks.*"""
    TAPS = (0, 1, 2, 4, 6)

    def __init__(self, in_channels=3, stages=B5_STAGES, stem=B5_STEM):
        assert in_channels == 3
        super().__init__(stages, stem, head=None)
        self.num_chs = [stages[i][4] for i in self.TAPS]

    def forward(self, x):
        feats = super().forward(x)                   # [x, None, None, stem, stage 0 .. 6]
        return [feats[4 + i] for i in self.TAPS]
