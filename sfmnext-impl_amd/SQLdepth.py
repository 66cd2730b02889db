"""`SQLdepth` of the reference (SQLdepth.py:9-50): encoder + Self-Query depth head as one module, `forward(x) -> depth [B,1,H/2,W/2]`;
what the metric-depth finetune loop trains.  Built for the configurations of this build: `model_type == "cvnxt_L"` / backbone
`convnext_large` (config E), resnet / resnet_lite, eff_b5."""
import os

import torch
import torch.nn as nn

import networks


class SQLdepth(nn.Module):
    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        model_type = getattr(opt, "model_type", "")
        if model_type == "cvnxt_L" or opt.backbone == "convnext_large":
            self.encoder = networks.Unet(pretrained=(not opt.load_pretrained_model), backbone="convnext_large", in_channels=3,
                                         num_classes=opt.model_dim, decoder_channels=opt.dec_channels,
                                         depths=getattr(opt, "sqd_convnext_depths", None), dims=getattr(opt, "sqd_convnext_dims", None))
        elif opt.backbone in ("resnet", "resnet_lite"):
            self.encoder = networks.ResnetEncoderDecoder(num_layers=opt.num_layers, num_features=opt.num_features, model_dim=opt.model_dim)
        elif model_type in ("nyu_pth_model", "eff_b5") or opt.backbone in ("eff_b5", "tf_efficientnet_b5_ap"):
            self.encoder = networks.BaseEncoder.build(num_features=opt.num_features, model_dim=opt.model_dim)
        else:
            raise NotImplementedError("SQLdepth: backbone %r is not built" % opt.backbone)
        cls = networks.Lite_Depth_Decoder_QueryTr if opt.backbone.endswith("_lite") else networks.Depth_Decoder_QueryTr
        self.depth_decoder = cls(in_channels=opt.model_dim, patch_size=opt.patch_size, dim_out=opt.dim_out, embedding_dim=opt.model_dim,
                                 query_nums=opt.query_nums, num_heads=4, min_val=opt.min_depth, max_val=opt.max_depth)
        if opt.load_pretrained_model:
            self.load_pretrained_model()

    def load_pretrained_model(self):
        """SQLdepth.py:33-46"""
        dev = next(self.parameters()).device
        enc = torch.load(os.path.join(self.opt.load_pt_folder, "encoder.pth"), map_location=dev)
        own = self.encoder.state_dict()
        self.encoder.load_state_dict({k: v for k, v in enc.items() if k in own})
        self.depth_decoder.load_state_dict(torch.load(os.path.join(self.opt.load_pt_folder, "depth.pth"), map_location=dev))

    def get_1x_lr_params(self):          # the pre-trained trunk at lr / 10 (train_ft_SQLdepth.py:180-182)
        return self.encoder.parameters()

    def get_10x_lr_params(self):
        return self.depth_decoder.parameters()

    def forward(self, x):
        return self.depth_decoder(self.encoder(x))["disp", 0]
