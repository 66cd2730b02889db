"""networks package of the MI355X build — same public names as the reference's networks/__init__.py
for the models on the KITTI self-supervised path, including the EfficientNet-b5 `BaseEncoder` (§8f-2).
and the ConvNeXt-L `Unet` (§8f-4; the timm trunk restated).  PoseDecoder and RectifyNet are not on the SQLdepth training path."""
from .base_encoder import BaseEncoder
from .depth_decoder_QTR import Depth_Decoder_QueryTr, Lite_Depth_Decoder_QueryTr
from .efficientnet import GenEfficientNet
from .layers import FullQueryLayer
from .pose_cnn import PoseCNN
from .resnet_encoder import (DecoderBN, LiteResnetEncoderDecoder, Resnet50EncoderDecoder, ResnetEncoder,
                             ResnetEncoderDecoder, UpSampleBN)


from .unet import Unet, UnetDecoder
from .convnext import ConvNeXtFeatures
