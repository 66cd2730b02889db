"""Host (CPU) implementations of the operator dispatch points of sqd.nnops, for the CPU wiring tests only: the product refuses host
tensors (no CPU fallback), so tests that check module wiring / state-dict layout on the CPU patch these into sqd.nnops
(`with host_ops.patched():`).  Plain torch restatements of the operators' definitions."""
import contextlib

import torch
import torch.nn.functional as F


def _act(y, act):
    return F.relu(y) if act == "relu" else F.leaky_relu(y, 0.01) if act == "leaky_relu" else y


def _conv(x, conv, act=None, skip=False, bn_stats=None):
    y = _act(F.conv2d(x, conv.weight, conv.bias, conv.stride, conv.padding), act)
    return (y, x) if skip else y


def conv2d(x, conv, act=None, skip=False):
    y = _conv(x, conv, act)
    return (y, x) if skip else y


def conv_bn_act(x, conv, bn, act, residual=None, input_affine=None, skip=False):
    if input_affine is not None:
        x = (x - input_affine[0]) / input_affine[1]
    y = bn(F.conv2d(x, conv.weight, conv.bias, conv.stride, conv.padding))
    if residual is not None:
        y = y + residual
    y = _act(y, act)
    return (y, x) if skip else y


def pose_head(x, conv, scale, split=False):
    out = scale * F.conv2d(x, conv.weight, conv.bias).mean(3).mean(2)
    if split:
        out = out.view(out.shape[0], -1, 1, 6)
        return out[..., :3], out[..., 3:]
    return out


def maxpool3x3s2(x, skip=False):
    y = F.max_pool2d(x, 3, 2, 1)
    return (y, x) if skip else y


def upsample_concat(x, skip):
    up = F.interpolate(x, size=[skip.size(2), skip.size(3)], mode="bilinear", align_corners=True)
    return torch.cat([up, skip], dim=1)


def linear(x, lin, act=None):
    y = F.linear(x, lin.weight, lin.bias)
    return F.leaky_relu(y, 0.01) if act == "leaky_relu" else y


def transformer_encoder(tokens, encoder):
    return encoder(tokens)


def full_query_layer(x, queries):
    n, c, h, w = x.shape
    xt = x.reshape(n, c, h * w)
    y = torch.matmul(queries, xt)
    summary = torch.matmul(torch.softmax(y, dim=2), xt.transpose(1, 2))
    return y.view(n, queries.shape[1], h, w), summary


def bins_head(energy_maps, conv1x1, y, min_val, max_val, raw_linear=False):
    if raw_linear:
        y = torch.relu(y) + 0.1
        y = y / y.sum(dim=1, keepdim=True)
    widths = F.pad((max_val - min_val) * y, (1, 0), mode="constant", value=min_val)
    edges = torch.cumsum(widths, dim=1)
    centers = 0.5 * (edges[:, :-1] + edges[:, 1:])
    out = torch.softmax(F.conv2d(energy_maps, conv1x1.weight, conv1x1.bias), dim=1)
    return torch.sum(out * centers.view(centers.shape[0], -1, 1, 1), dim=1, keepdim=True)


def layer_norm_channels(x, norm, pre_bias=None):
    if pre_bias is not None:
        x = x + pre_bias.view(1, -1, 1, 1)
    return F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), norm.weight, norm.bias, norm.eps).permute(0, 3, 1, 2)


def gelu(x):
    return F.gelu(x)


def scale_residual(shortcut, z, gamma):
    return shortcut + z * gamma.view(1, -1, 1, 1)


def upsample2x(x):
    return F.interpolate(x, scale_factor=2.0, mode="bilinear")


def dw_conv(x, conv, skip=False):
    y = F.conv2d(x, conv.weight, None, conv.stride, conv.padding, 1, conv.groups)
    return (y, x) if skip else y


def linear_channels(x, lin, act=None):
    return F.linear(x.permute(0, 2, 3, 1), lin.weight, lin.bias).permute(0, 3, 1, 2)


def patchify_conv(x, conv, s):
    return F.conv2d(x, conv.weight, conv.bias, s)


def tokens_with_positions(emb, pos):
    tokens = emb.flatten(2)                                     # reference networks/depth_decoder_QTR.py:49-51
    return (tokens + pos[:tokens.shape[2], :].T.unsqueeze(0)).permute(2, 0, 1)


def first_queries(tokens, Q):
    return tokens[:Q, ...].permute(1, 0, 2).contiguous()


NAMES = ("_conv", "conv2d", "conv_bn_act", "pose_head", "maxpool3x3s2", "upsample_concat", "linear", "transformer_encoder", "tokens_with_positions",
         "first_queries",
         "full_query_layer", "bins_head", "layer_norm_channels", "gelu", "scale_residual", "upsample2x", "dw_conv", "linear_channels",
         "patchify_conv")


@contextlib.contextmanager
def patched():
    from sqd import nnops
    saved = {n: getattr(nnops, n) for n in NAMES}
    try:
        for n in NAMES:
            setattr(nnops, n, globals()[n])
        yield
    finally:
        for n, f in saved.items():
            setattr(nnops, n, f)
