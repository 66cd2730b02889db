"""CPU restatement of the build's --sqd_bf16 arithmetic contract (test infrastructure: only tests/ import this).

The reference has no bf16 mode of its own (BASELINE.json configs[3] names "EfficientNet-b5, bf16" as a target); the build's
contract for it is DESIGN.md §3.4: every convolution that runs on the implicit-GEMM kernels (sqd_conv_fwd / sqd_conv_dgrad — all
dense convolutions and the bins regressor's Linear layers, which run as 1x1 convolutions over rows) rounds BOTH operands of its
forward product (x, w) and of its data-gradient product (dy, w) to ONE bf16 term, round-to-nearest-even (csrc/conv.hip rne_bf16),
accumulates in fp32 and adds the bias in fp32; weight and bias gradients are fp32 products of the unrounded x and dy; everything
else (depthwise convolutions, squeeze-excite, BatchNorm, the transformer encoder, Self Query Layer, bins head, PoseCNN's 1x1 head,
the photometric chain, Adam) is fp32.  `bf16_operands(modules)` makes exactly those modules of the fp32 oracle (oracle/torch_ref.py)
compute that way, so that a device step under --sqd_bf16 can be held to the oracle at fp32-accumulation-order tolerance instead of
a percent-level band against fp32 arithmetic."""
import contextlib

import torch
import torch.nn as nn
import torch.nn.functional as F


def rne_bf16(t: torch.Tensor) -> torch.Tensor:
    """fp32 -> nearest bf16 (ties to even) -> fp32: what csrc/conv.hip's rne_bf16 does to a finite operand"""
    return t.to(torch.bfloat16).to(torch.float32)


class _ConvBf16Operands(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, padding):
        ctx.save_for_backward(x, w)
        ctx.conf = (stride, padding, b is not None)
        return F.conv2d(rne_bf16(x), rne_bf16(w), b, stride, padding)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, padding, has_bias = ctx.conf
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = torch.nn.grad.conv2d_input(x.shape, rne_bf16(w), rne_bf16(dy), stride, padding)
        if ctx.needs_input_grad[1]:
            dw = torch.nn.grad.conv2d_weight(x, w.shape, dy, stride, padding)          # fp32 operands
        if has_bias and ctx.needs_input_grad[2]:
            db = dy.sum((0, 2, 3))
        return dx, dw, db, None, None


def _conv_forward(self, x):
    assert self.groups == 1 and self.dilation == (1, 1) and self.padding_mode == "zeros"
    return _ConvBf16Operands.apply(x, self.weight, self.bias, self.stride, self.padding)


def _linear_forward(self, x):
    rows = x.shape[0]
    y = _ConvBf16Operands.apply(x.reshape(rows, self.in_features, 1, 1), self.weight.view(self.out_features, self.in_features, 1, 1),
                                self.bias, (1, 1), (0, 0))
    return y.reshape(rows, self.out_features)


def gemm_modules(encoder, depth, pose):
    """the modules of an oracle (encoder, QueryTrDecoder, PoseCNN) triple whose products run on the implicit-GEMM kernels in the build"""
    mods = []
    for net in (encoder, depth, pose):
        for name, m in net.named_modules():
            if isinstance(m, nn.Conv2d) and m.groups == 1:
                if ".se." in "." + name + "." or name.endswith("convert_to_prob.0") or name == "pose_conv":
                    continue            # squeeze-excite gates, the bins head's 1x1 and PoseCNN's head run in their own fp32 kernels
                mods.append(m)
            elif isinstance(m, nn.Linear) and "bins_regressor" in name:
                mods.append(m)
    return mods


@contextlib.contextmanager
def bf16_operands(modules):
    """inside the block the given nn.Conv2d / nn.Linear modules round their operands as the build's --sqd_bf16 mode does"""
    saved = []
    try:
        for m in modules:
            saved.append((m, m.__dict__.get("forward")))
            fn = _conv_forward if isinstance(m, nn.Conv2d) else _linear_forward
            m.forward = fn.__get__(m, type(m))
        yield
    finally:
        for m, old in saved:
            if old is None:
                m.__dict__.pop("forward", None)
            else:
                m.forward = old
