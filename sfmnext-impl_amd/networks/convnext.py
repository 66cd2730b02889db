"""ConvNeXt trunk as a feature extractor — what `timm.create_model('convnext_large', features_only=True)` gives the reference's `Unet`
(reference networks/Unet.py:113-117; timm is third-party and not part of /root/reference: restated from the public ConvNeXt definition,
timm 0.6.x naming).  Stem: 4x4/4 convolution + LayerNorm2d; four stages of (3, 3, 27, 3) blocks at widths (192, 384, 768, 1536), stages
1-3 opened by LayerNorm2d + 2x2/2 convolution; block: 7x7 depthwise conv -> LayerNorm (eps 1e-6) -> Linear(C, 4C) -> GELU -> Linear(4C, C) ->
layer scale (init 1e-6) -> + shortcut.  forward(x [N,3,H,W]) -> [f4, f8, f16, f32] (strides 4, 8, 16, 32).
State-dict keys follow timm's FeatureListNet flattening: stem_0, stem_1, stages_{i}.downsample.{0,1}, stages_{i}.blocks.{j}.{conv_dw,
norm, mlp.fc1, mlp.fc2, gamma}.  Every operator runs through sqd.nnops (HIP kernels on the device)."""
import torch
import torch.nn as nn

from sqd import nnops as X

LARGE = dict(depths=(3, 3, 27, 3), dims=(192, 384, 768, 1536))


class _Mlp(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.fc1 = nn.Linear(dim, 4 * dim)
        self.fc2 = nn.Linear(4 * dim, dim)


class ConvNeXtBlock(nn.Module):
    def __init__(self, dim, ls_init_value=1e-6):
        super().__init__()
        self.conv_dw = nn.Conv2d(dim, dim, kernel_size=7, padding=3, groups=dim)
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.mlp = _Mlp(dim)
        self.gamma = nn.Parameter(ls_init_value * torch.ones(dim))

    def forward(self, x):
        # x has two consumers (the depthwise convolution and the shortcut): the convolution hands it through, the shortcut's gradient is
        # added inside the depthwise data-gradient kernel instead of by an accumulation pass over [N,C,H,W]
        z, x = X.dw_conv(x, self.conv_dw, skip=True)                             # (its bias is added inside the LayerNorm kernel)
        z = X.layer_norm_channels(z, self.norm, pre_bias=self.conv_dw.bias)
        z = X.gelu(X.linear_channels(z, self.mlp.fc1))
        z = X.linear_channels(z, self.mlp.fc2)
        return X.scale_residual(x, z, self.gamma)


class _Downsample(nn.Sequential):
    def __init__(self, cin, cout):
        super().__init__(nn.LayerNorm(cin, eps=1e-6), nn.Conv2d(cin, cout, kernel_size=2, stride=2))

    def forward(self, x):
        # (the 2x2 / 2 convolution runs on the implicit-GEMM kernels as it is — taps (r, s) of the strided input — with the LayerNorm's
        #  operand-scale record; the space-to-depth form costs a layout copy each way)
        return X.conv2d(X.layer_norm_channels(x, self[0]), self[1])


class _Stage(nn.Module):
    def __init__(self, cin, cout, depth, first):
        super().__init__()
        self.downsample = nn.Identity() if first else _Downsample(cin, cout)
        self.blocks = nn.Sequential(*[ConvNeXtBlock(cout) for _ in range(depth)])

    def forward(self, x):
        return self.blocks(self.downsample(x))


class ConvNeXtFeatures(nn.Module):
    def __init__(self, in_chans=3, depths=LARGE["depths"], dims=LARGE["dims"]):
        super().__init__()
        self.stem_0 = nn.Conv2d(in_chans, dims[0], kernel_size=4, stride=4)
        self.stem_1 = nn.LayerNorm(dims[0], eps=1e-6)
        for i in range(4):
            setattr(self, "stages_%d" % i, _Stage(dims[i - 1] if i else dims[0], dims[i], depths[i], first=(i == 0)))
        self.num_chs = list(dims)
        for m in self.modules():                                      # timm's _init_weights: trunc_normal_(std=.02), zero biases
            if isinstance(m, (nn.Conv2d, nn.Linear)):
                nn.init.trunc_normal_(m.weight, std=0.02)
                nn.init.zeros_(m.bias)

    def forward(self, x):
        x = X.layer_norm_channels(X.patchify_conv(x, self.stem_0, 4), self.stem_1)
        feats = []
        for i in range(4):
            x = getattr(self, "stages_%d" % i)(x)
            feats.append(x)
        return feats
