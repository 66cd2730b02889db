"""`Unet` of the reference (networks/Unet.py:9-312): a timm backbone as the encoder of a U-Net whose decoder blocks are
[bilinear up (to the skip's size, align_corners=True; x2 without a skip, Unet.py:244-251) -> concat -> (conv3x3 -> BN -> ReLU) x 2]
(Unet.py:211-256) and a final 1x1 convolution (Unet.py:293).  With decoder_channels (1024, 512, 256, 128) on ConvNeXt-L the fourth
block has no skip and brings stride 4 to stride 2 — the resolution the SQLdepth head works at.  Two backbones are built, the ones the
reference's KITTI args files name: `convnext_large` (`networks.convnext.ConvNeXtFeatures`, config E) and `tf_efficientnet_b5_ap`
(`networks.efficientnet.EfficientNetFeatures`: five features, decoder_channels (512, 256, 128, 64, 32) — the fifth block has no skip
and brings stride 2 to the full resolution; args_files/hisfog/kitti/effb5_320x1024.txt).
State-dict keys as in the reference: encoder.*, decoder.blocks.{i}.conv{1,2}.{conv,bn}.*, decoder.final_conv.*."""
import torch
import torch.nn as nn

from sqd import nnops as X

from .convnext import LARGE, ConvNeXtFeatures
from .efficientnet import EfficientNetFeatures


class Conv2dBnAct(nn.Module):
    """Unet.py:211-226 (activation ReLU)"""

    def __init__(self, in_channels, out_channels, kernel_size, padding=0):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=1, padding=padding, bias=False)
        self.bn = nn.BatchNorm2d(out_channels)
        self.act = nn.ReLU(inplace=True)

    def forward(self, x):
        return X.conv_bn_act(x, self.conv, self.bn, "relu")


class DecoderBlock(nn.Module):
    """Unet.py:229-256"""

    def __init__(self, in_channels, out_channels, scale_factor=2.0):
        super().__init__()
        self.scale_factor = scale_factor
        self.conv1 = Conv2dBnAct(in_channels, out_channels, 3, padding=1)
        self.conv2 = Conv2dBnAct(out_channels, out_channels, 3, padding=1)

    def forward(self, x, skip=None):
        if skip is not None:
            x = X.upsample_concat(x, skip) if self.scale_factor != 1.0 else torch.cat([x, skip], 1)
        elif self.scale_factor != 1.0:
            x = X.upsample2x(x)
        return self.conv2(self.conv1(x))


class UnetDecoder(nn.Module):
    """Unet.py:258-312 (center=False: nn.Identity)"""

    def __init__(self, encoder_channels, decoder_channels=(256, 128, 64, 32, 16), final_channels=1):
        super().__init__()
        self.center = nn.Identity()
        in_channels = [i + s for i, s in zip([encoder_channels[0]] + list(decoder_channels[:-1]), list(encoder_channels[1:]) + [0])]
        out_channels = decoder_channels
        if len(in_channels) != len(out_channels):
            in_channels.append(in_channels[-1] // 2)
        self.blocks = nn.ModuleList([DecoderBlock(i, o) for i, o in zip(in_channels, out_channels)])
        self.final_conv = nn.Conv2d(out_channels[-1], final_channels, kernel_size=(1, 1))
        for m in self.modules():                                  # Unet.py:296-302
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def forward(self, feats):
        x, skips = self.center(feats[0]), feats[1:]
        for i, b in enumerate(self.blocks):
            x = b(x, skips[i] if i < len(skips) else None)
        return X.conv2d(x, self.final_conv)


class Unet(nn.Module):
    def __init__(self, backbone="convnext_large", pretrained=True, in_channels=3, num_classes=5, decoder_channels=(1024, 512, 256, 128),
                 depths=None, dims=None, **_ignored):
        super().__init__()
        if backbone not in ("convnext_large", "tf_efficientnet_b5_ap"):
            raise NotImplementedError("Unet: the backbones of the reference's args files are built — convnext_large, tf_efficientnet_b5_ap (got %r)" % backbone)
        # (pretrained=True would download ImageNet weights through timm in the reference; there is no network here: random init)
        if backbone == "tf_efficientnet_b5_ap":
            self.encoder = EfficientNetFeatures(in_channels, **({"stages": _ignored["stages"]} if "stages" in _ignored else {}))
        else:
            self.encoder = ConvNeXtFeatures(in_channels, depths or LARGE["depths"], dims or LARGE["dims"])
        self.decoder = UnetDecoder(self.encoder.num_chs[::-1], tuple(decoder_channels), num_classes)

    def forward(self, x):
        feats = self.encoder(x)
        feats.reverse()                                           # Unet.py:145
        return self.decoder(feats)
