"""networks package of the MI355X build — same public names as the reference's networks/__init__.py
for the models on the KITTI self-supervised path, including the EfficientNet-b5 `BaseEncoder` (§8f-2).
The timm ConvNeXt `Unet`, PoseDecoder and RectifyNet are not built yet."""
from .base_encoder import BaseEncoder
from .depth_decoder_QTR import Depth_Decoder_QueryTr, Lite_Depth_Decoder_QueryTr
from .efficientnet import GenEfficientNet
from .layers import FullQueryLayer
from .pose_cnn import PoseCNN
from .resnet_encoder import (DecoderBN, LiteResnetEncoderDecoder, Resnet50EncoderDecoder, ResnetEncoder,
                             ResnetEncoderDecoder, UpSampleBN)


def _not_in_scope(name, why):
    class _Missing:
        def __init__(self, *a, **k):
            raise NotImplementedError("%s is not part of the MI355X hot-path build yet: %s" % (name, why))

        @staticmethod
        def build(*a, **k):
            raise NotImplementedError("%s is not part of the MI355X hot-path build yet: %s" % (name, why))
    _Missing.__name__ = name
    return _Missing


Unet = _not_in_scope("Unet", "ConvNeXt-L trunk comes from timm (SURVEY.md §8f-4)")
