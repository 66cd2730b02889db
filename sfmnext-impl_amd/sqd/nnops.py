"""Operator entry point of the networks (sqd.nnops).

Every tensor operation of the SQLdepth networks goes through this module, so that the module definitions keep the
reference's structure (state-dict keys stay those of the reference) while the arithmetic runs in the hand-written gfx950
kernels of libsqd.so (csrc/*.hip).  There is ONE implementation per operator: a shape the kernels do not take (channel /
feature counts that are not multiples of 4) is an error that names the operator and the shape, not a detour through another
library; host tensors are refused (the CPU restatement of these operators is test infrastructure: oracle/, tests/host_ops.py).
That includes the patch-token transformer encoder: it takes the embedding widths of the reference's args files — 32 (KITTI), 64
(args_res50_kitti_192x640_train.txt) and 56 (args_cityscapes_train.txt:9, args_cityscapes_eval.txt:7: 4 heads of 14) — and 16; another
width, or more than 512 tokens (256 at width 56 / 64), raises: nothing in the step runs on ATen's nn.TransformerEncoder."""
import torch
import torch.nn.functional as F

BACKEND = {
    "conv2d": "hip (implicit GEMM / input-patch kernels; 7x7 and 3x3 stride-2 stems via space-to-depth)",
    "conv_bn_act": "hip conv + hip bn/act/residual", "maxpool3x3s2": "hip", "upsample_concat": "hip",
    "pose_head": "hip", "depthwise_conv": "hip", "squeeze_excite": "hip", "linear": "hip (1x1 implicit GEMM over rows)",
    "transformer_encoder": "hip (fused attention up to 512 tokens — 256 at embedding width 56 / 64 —, feed-forward, add+dropout+layernorm)",
    "full_query_layer": "hip", "bins_head": "hip",
}

ATEN_CALLS = {}               # operator -> calls that ran on ATen: nothing writes to it any more (kept so that tests can assert it stays empty)


def backend_report():
    """BACKEND: one implementation per operator — there is nothing to report per process."""
    return dict(BACKEND)


def _device_only(x, what):
    if not x.is_cuda:
        raise RuntimeError("sqd: %s needs a tensor on the MI355X device — the hot path has no CPU fallback" % what)


def configure(opt, device):
    """What every entry point sets before building its networks (Trainer, evaluate_depth.build_models,
    finetune.FinetuneTrainer): the convolutions' operand precision (--sqd_bf16), a pinned plan set (--sqd_conv_plans) or plan
    timing on the first call unless --sqd_no_conv_tune.  The kernels are NHWC / KRSC only, so channels_last is forced."""
    from . import nnkernels
    nnkernels.set_conv_precision(2 if opt.sqd_bf16 else 0)
    # two-term fp16 operands (f16x2) are plans of the fp32-equivalent arithmetic; --sqd_no_f16x2 keeps the round-4 plan space
    nnkernels.amax_enable(not opt.sqd_bf16 and not getattr(opt, "sqd_no_f16x2", False))
    nnkernels.TUNE_SPACE["f16x2"] = not getattr(opt, "sqd_no_f16x2", False)
    nnkernels.TUNE_CONV = not opt.sqd_no_conv_tune and torch.device(device).type == "cuda"     # first step: ~2 s of plan timing
    nnkernels._WGRAD_SETTLED.clear()     # (a run decides from settled weight-gradient plans from ITS second step on, whatever ran in the process before)
    if getattr(opt, "sqd_conv_plans", None):
        import json
        with open(opt.sqd_conv_plans) as f:
            nnkernels.load_plans(json.load(f))
    opt.sqd_channels_last = True


def _conv(x, conv, act=None, skip=False, bn_stats=None, input_affine=None):
    """skip=True: -> (y, x') with x' the input handed through the convolution node (see nnkernels.Conv2d).
    bn_stats: a list that receives (partials, rows) when the convolution's epilogue produced the statistics partials of
    the BatchNorm that follows (training, plan without split-K)."""
    _device_only(x, "conv2d")
    from . import nnkernels
    native = nnkernels.conv_module_supported(conv)
    s2d = not native and nnkernels.stem_s2d_supported(conv, x)
    if not (native or s2d):
        raise RuntimeError("sqd: conv2d %d -> %d channels, kernel %s, stride %s, padding %s: the implicit-GEMM kernels take channel "
                           "counts that are multiples of 4 (3- / 6-channel frames: the 7x7 stride-2 stems only)"
                           % (conv.in_channels, conv.out_channels, tuple(conv.kernel_size), tuple(conv.stride), tuple(conv.padding)))
    # a dense NCHW frame into a 7x7 stem: layout conversion and (x - a) / b happen inside the space-to-depth pass
    planar = s2d and not skip and nnkernels.stem_s2d_planar_supported(conv, x)
    if input_affine is not None and not planar:
        x = (x - input_affine[0]) / input_affine[1]
    stats = geom = None
    if bn_stats is not None:
        geom = nnkernels.conv_out_geom(x, conv, s2d)
        M, K = geom[0] * geom[9] * geom[10], geom[4]
        # room for the plan with the most rows of partials: 64-row tiles, or the input-patch kernel's 4 x 16 pixel patches
        # (partial patches at the right / lower border make that more than M / 64)
        rows_max = max((M + 63) // 64, geom[0] * ((geom[9] + 3) // 4) * ((geom[10] + 15) // 16))
        stats = torch.empty(rows_max * K * 2, device=x.device, dtype=torch.float32)
    if native:
        out = nnkernels.conv2d_native(x, conv, act, skip, stats)
    elif planar:
        out = nnkernels.conv2d_stem_s2d_planar([(x, None)], conv, act, stats, input_affine or (0.0, 1.0))
    else:
        y = nnkernels.conv2d_stem_s2d(x, conv, act, stats)
        out = (y, x) if skip else y
    if stats is not None:
        rows = nnkernels.conv_stats_rows(geom)      # after the call: the first call may have (re)tuned the plan
        if rows > 0:
            bn_stats.append((stats, rows))
    return out


def conv2d(x, conv, act=None, skip=False):
    """conv (nn.Conv2d holding weight/bias/stride/padding) applied to x, optional activation.  skip=True: -> (y, x') where x' must replace
    x for x's other consumer: that consumer's gradient is then added in this convolution's data-gradient epilogue instead of by a
    separate accumulation pass over the whole tensor."""
    return _conv(x, conv, act, skip)


def stem_pairs(pairs, conv, act=None):
    """conv applied to the channel concatenation of frame pairs: pairs = [(x0, x1), ...] -> [B * len(pairs), K, H', W'] with row
    b * len(pairs) + i = pair i of sample b (the pose network's input assembly, reference trainer.py:319-326, without the copies)."""
    x0 = pairs[0][0]
    _device_only(x0, "stem_pairs")
    from . import nnkernels
    probe = torch.empty((1, conv.in_channels, x0.shape[2], x0.shape[3]), device="meta")
    if nnkernels.stem_s2d_supported(conv, probe) and all(a.is_contiguous() and b.is_contiguous() and a.dtype == torch.float32 for a, b in pairs):
        return nnkernels.conv2d_stem_s2d_planar(pairs, conv, act)
    B, S = x0.shape[0], len(pairs)
    x = torch.stack([torch.cat(p, 1) for p in pairs], 1).reshape((B * S, -1) + tuple(x0.shape[2:]))
    return conv2d(x.contiguous(memory_format=torch.channels_last), conv, act)


def conv_bn_act(x, conv, bn, act, residual=None, input_affine=None, skip=False):
    """[input (x-a)/b] -> conv -> BatchNorm2d (batch stats in training, running stats in eval)
    -> [+ residual] -> activation.  skip=True: -> (y, x') where x' must replace x for every further consumer of x (the
    residual branch, the down-sample convolution): their gradient then reaches x through this convolution's data-gradient
    epilogue instead of a separate accumulation pass."""
    training = bn.training or bn.running_mean is None
    pre = [] if training else None      # statistics partials from the convolution's epilogue
    if skip:
        y, x_skip = _conv(x, conv, None, True, pre, input_affine)
        return _bn_act(y, bn, act, residual, pre), x_skip
    y = _conv(x, conv, None, False, pre, input_affine)
    return _bn_act(y, bn, act, residual, pre)


def _bn_act(y, bn, act, residual, pre=None):
    _device_only(y, "BatchNorm")
    from . import nnkernels
    if not nnkernels.bn_supported(y.shape[1]):
        raise RuntimeError("sqd: BatchNorm kernel needs a channel count that is a multiple of 4 (C=%d)" % y.shape[1])
    if pre:
        return nnkernels.batch_norm_act(y, bn, act, residual, pre[0][0], pre[0][1])
    return nnkernels.batch_norm_act(y, bn, act, residual)


def pose_head(x, conv, scale, split=False):
    """scale * conv(x).mean(3).mean(2) for PoseCNN's 1x1 head (reference networks/pose_cnn.py:40-45) -> [B, J]; split=True:
    -> (axisangle, translation), each [B, J/6, 1, 3] — out.view(-1, F, 1, 6)[..., :3] and [..., 3:] as dense tensors."""
    _device_only(x, "pose_head")
    if not (conv.kernel_size == (1, 1) and conv.out_channels <= 16 and conv.bias is not None):
        raise RuntimeError("sqd: pose_head is a 1x1 convolution with bias onto <= 16 channels; got kernel %s, %d channels, bias %s"
                           % (tuple(conv.kernel_size), conv.out_channels, conv.bias is not None))
    from . import nnkernels
    return nnkernels.PoseHead.apply(x, conv.weight, conv.bias, scale, split)


def dw_conv_bn_act(x, conv, bn, act, stride, pool=False):
    """depthwise k x k convolution with TensorFlow "SAME" padding -> BatchNorm2d -> activation (EfficientNet blocks).
    pool=True: a squeeze-and-excite gate reads the result next — the BatchNorm's element-wise pass takes its pooled sums on the way."""
    _device_only(x, "depthwise convolution")
    from . import nnkernels
    y = nnkernels.DepthwiseConv.apply(x, conv.weight, stride, "same")
    return nnkernels.batch_norm_act(y, bn, act, pool=pool)


def layer_norm_channels(x, norm, pre_bias=None):
    """LayerNorm over the channel axis of a [N,C,H,W] map (timm's LayerNorm2d, or a ConvNeXt block's nn.LayerNorm applied between
    its permutes); `pre_bias` [C] is added first (the bias of the depthwise convolution in front)."""
    _device_only(x, "layer_norm_channels")
    from . import nnkernels
    return nnkernels.LayerNormRows.apply(x, pre_bias, norm.weight, norm.bias, norm.eps)


def gelu(x):
    _device_only(x, "gelu")
    from . import nnkernels
    return nnkernels.Gelu.apply(x)


def scale_residual(shortcut, z, gamma):
    """shortcut + gamma[c] * z (layer scale + residual of a ConvNeXt block)"""
    _device_only(z, "scale_residual")
    from . import nnkernels
    return nnkernels.ScaleResidual.apply(shortcut, z, gamma)


def upsample2x(x):
    """F.interpolate(x, scale_factor=2, mode='bilinear')"""
    _device_only(x, "upsample2x")
    from . import nnkernels
    return nnkernels.Upsample2x.apply(x)


def linear_channels(x, lin, act=None):
    """nn.Linear applied to the channel axis of a [N,C,H,W] map (what a ConvNeXt block does between its permutes) = a 1x1 convolution"""
    _device_only(x, "linear_channels")
    from . import nnkernels
    K, C = lin.weight.shape
    w4 = lin.weight.view(K, C, 1, 1)
    w4._sqd_w_src = lin.weight                   # (the same values: the filter's max |.| is the parameter's, see nnkernels.amax_of_weight)
    return nnkernels.Conv2d.apply(x, w4, lin.bias, 1, 0, act, False, None, None)


def dw_conv(x, conv, skip=False):
    """depthwise convolution with symmetric padding, WITHOUT its bias (the caller folds conv.bias into the LayerNorm that follows).
    skip=True (stride 1): -> (y, x') where x' must replace x for x's other consumer (the block's shortcut): that consumer's gradient is
    then added inside the depthwise data-gradient kernel instead of by a separate accumulation pass over [N,C,H,W]."""
    _device_only(x, "depthwise convolution")
    from . import nnkernels
    return nnkernels.DepthwiseConv.apply(x, conv.weight, conv.stride[0], conv.padding[0], skip)


def patchify_conv(x, conv, s):
    """a k = stride = s convolution (ConvNeXt's 4x4/4 stem on 3 channels, its 2x2/2 down-sampling) = a 1x1 convolution on the
    space-to-depth(s) image: [N,C,H,W] -> [N, s*s*C, H/s, W/s] rows ordered (dy, dx, c), filter regrouped the same way.  The 3-channel
    stem's 48 input features are a multiple of 4, so it runs on the implicit-GEMM kernels too."""
    _device_only(x, "patchify_conv")
    N, C, H, W = x.shape
    K = conv.out_channels
    xs = x.reshape(N, C, H // s, s, W // s, s).permute(0, 3, 5, 1, 2, 4).reshape(N, s * s * C, H // s, W // s)
    w = conv.weight.permute(0, 2, 3, 1).reshape(K, s * s * C, 1, 1).contiguous(memory_format=torch.channels_last)
    w._sqd_w_src = conv.weight                   # (a permutation of the parameter's values)
    from . import nnkernels
    return nnkernels.Conv2d.apply(xs.contiguous(memory_format=torch.channels_last), w, conv.bias, 1, 0, None, False, None, None)


def squeeze_excite(x, conv_reduce, conv_expand):
    """x * sigmoid(expand(swish(reduce(mean_hw(x))))) — the squeeze-and-excite gate of the EfficientNet blocks."""
    _device_only(x, "squeeze-and-excite")
    from . import nnkernels
    return nnkernels.SqueezeExcite.apply(x, conv_reduce.weight, conv_reduce.bias, conv_expand.weight, conv_expand.bias)


def stem_same_conv_bn_act(x, conv, bn, act):
    """EfficientNet stem: 3x3 stride-2 convolution on the 3-channel image with TensorFlow "SAME" padding -> BatchNorm -> activation
    (the convolution runs as a 3x3 / stride 1 one on the space-to-depth image: nnkernels.conv2d_stem3_same_s2d)."""
    _device_only(x, "stem convolution")
    from . import nnkernels
    return nnkernels.batch_norm_act(nnkernels.conv2d_stem3_same_s2d(x, conv), bn, act)


def maxpool3x3s2(x, skip=False):
    """skip=True: -> (y, x') with x' to be read by x's other consumer (see nnkernels.MaxPool3x3s2)."""
    _device_only(x, "max-pool")
    from . import nnkernels
    if x.shape[1] % 4:
        raise RuntimeError("sqd: max-pool kernel needs a channel count that is a multiple of 4")
    return nnkernels.MaxPool3x3s2.apply(x, skip)


def upsample_concat(x, skip):
    """bilinear resize of x to skip's size (align_corners=True) and channel concat [up(x), skip]."""
    _device_only(x, "upsample+concat")
    from . import nnkernels
    if x.shape[1] % 4 or skip.shape[1] % 4:
        raise RuntimeError("sqd: upsample+concat kernel needs channel counts that are multiples of 4")
    return nnkernels.UpsampleConcat.apply(x, skip)


def linear(x, lin, act=None):
    """nn.Linear (+ LeakyReLU(0.01)) of the bins regressor."""
    _device_only(x, "linear")
    from . import nnkernels
    if not nnkernels.linear_supported(lin, x):
        raise RuntimeError("sqd: linear %d -> %d features on input %s: the kernel takes feature counts that are multiples of 4"
                           % (lin.in_features, lin.out_features, tuple(x.shape)))
    return nnkernels.linear_native(x, lin, act)


def tokens_with_positions(emb, pos):
    """the patch embedding [B,E,h,w] plus the first h*w rows of the positional table [Tmax,E], as the [T,B,E] sequence the encoder reads
    (reference networks/depth_decoder_QTR.py:49-51: flatten(2) + positional_encodings[:T].T, permute(2, 0, 1))"""
    _device_only(emb, "tokens_with_positions")
    from . import nnkernels
    if emb.shape[2] * emb.shape[3] > pos.shape[0] or pos.shape[1] != emb.shape[1] or not pos.is_contiguous():
        raise RuntimeError("sqd: %d patch tokens of width %d against a positional table %s" % (emb.shape[2] * emb.shape[3], emb.shape[1], tuple(pos.shape)))
    return nnkernels.TokensWithPos.apply(emb, pos)


def first_queries(tokens, Q):
    """tokens [T,B,E] -> the first Q of them as [B,Q,E] (reference networks/depth_decoder_QTR.py:52)"""
    _device_only(tokens, "first_queries")
    from . import nnkernels
    if Q > tokens.shape[0]:
        raise RuntimeError("sqd: %d queries out of %d tokens" % (Q, tokens.shape[0]))
    return nnkernels.FirstQueries.apply(tokens, Q)


def transformer_encoder(tokens, encoder):
    """tokens [T,B,E] through nn.TransformerEncoder (4 post-norm layers, ReLU feed-forward)."""
    _device_only(tokens, "transformer_encoder")
    from . import nnkernels
    if not nnkernels.encoder_supported(encoder):
        l0 = encoder.layers[0]
        raise RuntimeError("sqd: transformer_encoder with embedding width %d, feed-forward %d, norm_first=%s: the patch-token encoder kernels "
                           "take post-norm ReLU layers whose width is a multiple of 4 up to 64, with attention heads of 4 | 8 (width 16 / 32), "
                           "14 (width 56) or 16 (width 64) features (reference networks/depth_decoder_QTR.py:14-16 with the model_dim of "
                           "every args file: 32, 56, 64); there is no ATen fallback"
                           % (l0.linear1.weight.shape[1], l0.linear1.weight.shape[0], l0.norm_first))
    return nnkernels.transformer_encoder_native(tokens, encoder)


def full_query_layer(x, queries):
    """Self Query Layer (reference networks/layers.py:7-21): x [B,E,h,w], queries [B,Q,E] ->
    energy maps [B,Q,h,w] (raw dot products) and summaries [B,Q,E] (softmax over the h*w pixels)."""
    _device_only(x, "Self Query Layer")
    from . import ops
    E, Q = x.shape[1], queries.shape[1]
    if not ops.sql_supported(E, Q):
        Ep = (E + 15) // 16 * 16
        if E % 4 == 0 and ops.sql_supported(Ep, Q):
            # an embedding width between the kernel's 16-channel steps (model_dim 56 of the reference's Cityscapes args files): zero channels
            # add nothing to the x . query products and come back as zero columns of the summaries, so the width is rounded up with zeros
            # — two small copies (ATen pads) around the same kernels; the data gradients of the zero channels are dropped by the slices
            xp = F.pad(x, (0, 0, 0, 0, 0, Ep - E))
            if x.is_contiguous(memory_format=torch.channels_last):
                xp = xp.contiguous(memory_format=torch.channels_last)
            y, summ = ops.SelfQueryLayer.apply(xp, F.pad(queries, (0, Ep - E)))
            return y, summ[..., :E]
        raise RuntimeError("sqd: Self Query Layer kernel supports embedding widths that are multiples of 4 up to 64, Q <= 128; got E=%d Q=%d" % (E, Q))
    return ops.SelfQueryLayer.apply(x, queries)


def bins_head(energy_maps, conv1x1, y, min_val, max_val, raw_linear=False):
    """1x1 conv + channel softmax over the energy maps, expected value over the adaptive bin centres
    (reference networks/depth_decoder_QTR.py:61-70).  y [B,dim_out] = normalised bin widths — or, with raw_linear=True, the
    regressor's raw outputs of the norm == "linear" head (:56-60: relu + 0.1, divide by the row sum), normalised here."""
    _device_only(energy_maps, "bins head")
    if raw_linear and y.shape[1] <= 128:
        from . import ops
        centers = ops.BinCenters.apply(y, min_val, max_val)
    else:
        if raw_linear:
            y = torch.relu(y) + 0.1
            y = y / y.sum(dim=1, keepdim=True)
        widths = (max_val - min_val) * y
        widths = F.pad(widths, (1, 0), mode="constant", value=min_val)
        edges = torch.cumsum(widths, dim=1)
        centers = 0.5 * (edges[:, :-1] + edges[:, 1:])
    from . import ops
    Q, D = energy_maps.shape[1], conv1x1.weight.shape[0]
    if not ops.bins_supported(Q, D):
        raise RuntimeError("sqd: bins head kernel supports Q, dim_out <= 128; got Q=%d dim_out=%d" % (Q, D))
    return ops.BinsHead.apply(energy_maps, conv1x1.weight, conv1x1.bias, centers)
