"""ResNet encoder + DecoderBN of SQLdepth for the MI355X build.

Surface kept from the reference (constructor arguments, forward signatures, state-dict keys —
SURVEY.md App. C): `ResnetEncoder`, `UpSampleBN`, `DecoderBN`, `ResnetEncoderDecoder`,
`Resnet50EncoderDecoder` (reference networks/resnet_encoder.py:64-168) and the ResNet-18 variant
`LiteResnetEncoderDecoder` (reference networks/lite_res_encoder.py:148-157).

The trunk is a from-scratch ResNet v1.5 with torchvision's parameter names (the reference takes it
from torchvision, which is neither in /root/reference nor in this image; ImageNet weights cannot be
downloaded here, so `pretrained=True` of the reference (resnet_encoder.py:153) becomes local init
unless a checkpoint is loaded through --load_pretrained_model).

All tensor arithmetic goes through sqd.nnops — the dispatch point where the hand-written gfx950
kernels replace ATen one operator at a time."""
import torch
import torch.nn as nn

from sqd import nnops as X

_BLOCKS = {18: ("basic", (2, 2, 2, 2)), 34: ("basic", (3, 4, 6, 3)), 50: ("bottleneck", (3, 4, 6, 3)),
           101: ("bottleneck", (3, 4, 23, 3)), 152: ("bottleneck", (3, 8, 36, 3))}


def _conv(cin, cout, k, stride=1, pad=0, bias=False):
    return nn.Conv2d(cin, cout, k, stride, pad, bias=bias)


class _Residual(nn.Module):
    """One residual unit; `kind` selects the 2-conv (ResNet-18/34) or 3-conv (50+) body.
    Attribute names conv1/bn1/.../downsample follow torchvision so checkpoints load unchanged."""

    def __init__(self, kind, cin, planes, stride):
        super().__init__()
        self.kind = kind
        cout = planes * (4 if kind == "bottleneck" else 1)
        if kind == "bottleneck":
            self.conv1, self.bn1 = _conv(cin, planes, 1), nn.BatchNorm2d(planes)
            self.conv2, self.bn2 = _conv(planes, planes, 3, stride, 1), nn.BatchNorm2d(planes)   # v1.5: stride here
            self.conv3, self.bn3 = _conv(planes, cout, 1), nn.BatchNorm2d(cout)
        else:
            self.conv1, self.bn1 = _conv(cin, planes, 3, stride, 1), nn.BatchNorm2d(planes)
            self.conv2, self.bn2 = _conv(planes, planes, 3, 1, 1), nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(_conv(cin, cout, 1, stride), nn.BatchNorm2d(cout))

    def forward(self, x, tap=False):
        """tap=True (units with a down-sample branch): -> (out, x') where x' is the unit's input handed through both of its
        consumers' nodes — the encoder exports it as the previous stage's feature tap, so the decoder's skip gradient is
        added inside the down-sample convolution's data-gradient kernel too."""
        # x has two consumers (conv1 and the shortcut): the second one reads the copy handed through conv1's node, so both
        # gradients meet in conv1's data-gradient kernel (nnops.conv_bn_act, skip=True)
        y, x = X.conv_bn_act(x, self.conv1, self.bn1, "relu", skip=True)
        if self.downsample is None:
            shortcut = x
        elif tap:
            shortcut, x = X.conv_bn_act(x, self.downsample[0], self.downsample[1], None, skip=True)
        else:
            shortcut = X.conv_bn_act(x, self.downsample[0], self.downsample[1], None)
        if self.kind == "bottleneck":
            y = X.conv_bn_act(y, self.conv2, self.bn2, "relu")
            y = X.conv_bn_act(y, self.conv3, self.bn3, "relu", residual=shortcut)
        else:
            y = X.conv_bn_act(y, self.conv2, self.bn2, "relu", residual=shortcut)
        return (y, x) if tap else y


class ResNetTrunk(nn.Module):
    def __init__(self, num_layers):
        super().__init__()
        kind, counts = _BLOCKS[num_layers]
        self.conv1, self.bn1 = _conv(3, 64, 7, 2, 3), nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=False)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        cin = 64
        for i, (planes, n) in enumerate(zip((64, 128, 256, 512), counts)):
            units = []
            for j in range(n):
                units.append(_Residual(kind, cin, planes, (1 if i == 0 else 2) if j == 0 else 1))
                cin = planes * (4 if kind == "bottleneck" else 1)
            setattr(self, "layer%d" % (i + 1), nn.Sequential(*units))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(cin, 1000)   # kept for checkpoint compatibility; never on the path
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")


class ResnetEncoder(nn.Module):
    """Input normalisation + the five feature taps (reference networks/resnet_encoder.py:89-100)."""

    def __init__(self, num_layers, pretrained=False, num_input_images=1):
        super().__init__()
        if num_layers not in _BLOCKS:
            raise ValueError("{} is not a valid number of resnet layers".format(num_layers))
        if num_input_images != 1:
            raise NotImplementedError("multi-image ResNet input is not on the SQLdepth training path")
        self.num_ch_enc = [64, 64, 128, 256, 512] if num_layers <= 34 else [64, 256, 512, 1024, 2048]
        self.encoder = ResNetTrunk(num_layers)

    planar_input = True      # the stem takes the frame as it comes (dense NCHW): no channels-last copy before the call

    def forward(self, input_image):
        e = self.encoder
        f0 = X.conv_bn_act(input_image, e.conv1, e.bn1, "relu", input_affine=(0.45, 0.225))   # (x-0.45)/0.225
        # every tap but the last has two consumers (the next stage and the decoder's skip connection); the decoder reads the
        # copy handed through the next stage's first nodes, so the two gradients meet in a kernel epilogue, not in an add pass
        pooled, f0 = X.maxpool3x3s2(f0, skip=True)
        f1 = e.layer1(pooled)
        f2, f1 = self._stage(e.layer2, f1)
        f3, f2 = self._stage(e.layer3, f2)
        f4, f3 = self._stage(e.layer4, f3)
        self.features = [f0, f1, f2, f3, f4]
        return self.features

    @staticmethod
    def _stage(layer, x):
        """-> (stage output, the stage input as handed through its first unit)"""
        units = list(layer)
        if units[0].downsample is None:
            y, tap = units[0](x), x
        else:
            y, tap = units[0](x, tap=True)
        for u in units[1:]:
            y = u(y)
        return y, tap


class UpSampleBN(nn.Module):
    """bilinear x-up (align_corners=True) to the skip size -> concat -> 2 x (conv3x3 + BN + LeakyReLU)."""

    def __init__(self, skip_input, output_features):
        super().__init__()
        self._net = nn.Sequential(_conv(skip_input, output_features, 3, 1, 1, bias=True), nn.BatchNorm2d(output_features),
                                  nn.LeakyReLU(),
                                  _conv(output_features, output_features, 3, 1, 1, bias=True),
                                  nn.BatchNorm2d(output_features), nn.LeakyReLU())

    def forward(self, x, concat_with):
        f = X.upsample_concat(x, concat_with)
        f = X.conv_bn_act(f, self._net[0], self._net[1], "leaky_relu")
        return X.conv_bn_act(f, self._net[3], self._net[4], "leaky_relu")


class DecoderBN(nn.Module):
    """reference networks/resnet_encoder.py:120-147; `skips` = encoder tap widths (1024,512,256,64 for
    ResNet-50+, 256,128,64,64 for ResNet-18: reference lite_res_encoder.py:127-130).
    conv2 is a 1x1 convolution with padding=1 (resnet_encoder.py:125) — kept."""

    def __init__(self, num_features=2048, num_classes=1, bottleneck_features=512, skips=(1024, 512, 256, 64)):
        super().__init__()
        f = int(num_features)
        self.conv2 = _conv(bottleneck_features, f, 1, 1, 1, bias=True)
        self.up1 = UpSampleBN(f // 1 + skips[0], f // 2)
        self.up2 = UpSampleBN(f // 2 + skips[1], f // 4)
        self.up3 = UpSampleBN(f // 4 + skips[2], f // 8)
        self.up4 = UpSampleBN(f // 8 + skips[3], f // 16)
        self.conv3 = _conv(f // 16, num_classes, 3, 1, 1, bias=True)

    def forward(self, features):
        x = X.conv2d(features[4], self.conv2)
        for up, skip in zip((self.up1, self.up2, self.up3, self.up4), (features[3], features[2], features[1], features[0])):
            x = up(x, skip)
        return X.conv2d(x, self.conv3)


class ResnetEncoderDecoder(nn.Module):
    planar_input = True

    def __init__(self, num_layers=50, num_features=512, model_dim=32):
        super().__init__()
        if num_layers < 50:
            raise ValueError("ResnetEncoderDecoder needs num_layers >= 50 (skip widths 1024/512/256/64); "
                             "use LiteResnetEncoderDecoder for ResNet-18")
        self.encoder = ResnetEncoder(num_layers=num_layers, pretrained=True, num_input_images=1)
        self.decoder = DecoderBN(num_features=num_features, num_classes=model_dim, bottleneck_features=2048)

    def forward(self, x, **kwargs):
        return self.decoder(self.encoder(x), **kwargs)


class Resnet50EncoderDecoder(ResnetEncoderDecoder):
    def __init__(self, model_dim=128):
        super().__init__(num_layers=50, num_features=512, model_dim=model_dim)


class LiteResnetEncoderDecoder(nn.Module):
    planar_input = True

    def __init__(self, model_dim=128):
        super().__init__()
        self.encoder = ResnetEncoder(num_layers=18, pretrained=True, num_input_images=1)
        self.decoder = DecoderBN(num_features=256, num_classes=model_dim, bottleneck_features=512,
                                 skips=(256, 128, 64, 64))

    def forward(self, x, **kwargs):
        return self.decoder(self.encoder(x), **kwargs)
