"""EfficientNet-b5 encoder + DecoderBN for `--backbone eff_b5` (reference networks/base_encoder.py:24-107).

`BaseEncoder.build(model_dim, num_features)` keeps the reference's constructor surface; where the reference downloads the
trunk from torch.hub (base_encoder.py:94, pretrained=False), this build instantiates its own restatement of the same
architecture (networks/efficientnet.py).  State-dict keys: encoder.original_model.* (hub names) and decoder.* (the
DecoderBN of networks/resnet_encoder.py with the b5 skip widths: +176, +64, +40, +24 — base_encoder.py:31-34)."""
import torch.nn as nn

from .efficientnet import GenEfficientNet
from .resnet_encoder import DecoderBN


class Encoder(nn.Module):
    """reference base_encoder.py:58-73: the feature list of the trunk, stage by stage"""

    def __init__(self, backend):
        super().__init__()
        self.original_model = backend

    def forward(self, x):
        return self.original_model(x)


class BaseEncoder(nn.Module):
    def __init__(self, backend, model_dim=32, num_features=2048):
        super().__init__()
        self.encoder = Encoder(backend)
        # taps features[4, 5, 6, 8, 11] (base_encoder.py:41): 24 @ /2, 40 @ /4, 64 @ /8, 176 @ /16, conv_head 2048 @ /32
        self.decoder = DecoderBN(num_features=num_features, num_classes=model_dim, bottleneck_features=2048, skips=(176, 64, 40, 24))

    def forward(self, x, **kwargs):
        f = self.encoder(x)
        return self.decoder((f[4], f[5], f[6], f[8], f[11]), **kwargs)

    @classmethod
    def build(cls, model_dim, **kwargs):
        return cls(GenEfficientNet(), model_dim=model_dim, **kwargs)
