"""GPU parity of the fp32-MFMA Self Query Layer (csrc/sql.hip through the C ABI) against the oracle
(oracle/torch_ref.py::full_query_layer, pinned to the reference by golden group G10)."""
import numpy as np
import pytest
import torch

from conftest import tt

pytestmark = pytest.mark.gpu


def _run(x, K, wy, ws, channels_last=False):
    """channels_last: x in the layout the producing convolution writes ([B,h,w,E] memory), read and differentiated in place"""
    from sqd import ops
    xg = x.cuda()
    if channels_last:
        xg = xg.contiguous(memory_format=torch.channels_last)
    xg, Kg = xg.requires_grad_(True), K.cuda().requires_grad_(True)
    y, s = ops.SelfQueryLayer.apply(xg, Kg)
    ((y * wy.cuda()).sum() + (s * ws.cuda()).sum()).backward()
    if channels_last and x.shape[1] > 1 and x.shape[2] * x.shape[3] > 1:
        assert xg.grad.is_contiguous(memory_format=torch.channels_last)      # the gradient comes back in x's layout
    return y.detach().cpu(), s.detach().cpu(), xg.grad.cpu(), Kg.grad.cpu()


def _ref(x, K, wy, ws):
    from oracle import torch_ref as O
    xr, Kr = x.clone().requires_grad_(True), K.clone().requires_grad_(True)
    y, s = O.full_query_layer(xr, Kr)
    ((y * wy).sum() + (s * ws).sum()).backward()
    return y.detach(), s.detach(), xr.grad, Kr.grad


def _check(got, want):
    names = ("energy maps", "summary", "grad_x", "grad_K")
    for n, a, b in zip(names, got, want):
        scale = float(b.abs().max())
        err = float((a - b).abs().max())
        assert err <= 1e-4 * scale + 1e-6, (n, err, scale)


def test_golden_g10(golden):
    g = golden("g10_full_query_layer")
    x, K, wy, ws = tt(g["x"]), tt(g["K"]), tt(g["wy"]), tt(g["ws"])
    y, s, gx, gK = _run(x, K, wy, ws)
    _check((y, s, gx, gK), (tt(g["y"]), tt(g["summary"]), tt(g["grad_x"]), tt(g["grad_K"])))


@pytest.mark.parametrize("B,E,Q,h,w", [(2, 16, 12, 12, 20), (2, 32, 64, 48, 160), (1, 32, 120, 24, 80), (2, 32, 128, 20, 64),
                                       (1, 16, 5, 13, 17), (3, 32, 33, 7, 9), (2, 16, 24, 96, 320),
                                       # E = 48 / 64: config B' of the reference (args_res50_kitti_192x640_train.txt: model_dim 64, Q 120)
                                       (2, 64, 120, 24, 80), (1, 48, 33, 13, 17), (2, 64, 64, 48, 160), (1, 64, 12, 7, 9), (2, 48, 128, 20, 64)])
def test_vs_oracle(B, E, Q, h, w):
    torch.manual_seed(B * 1000 + Q)
    x = torch.randn(B, E, h, w)
    K = 0.5 * torch.randn(B, Q, E)
    K[0, 0] *= 8.0                       # one sharply peaked query: exercises the online-softmax rescale path
    wy, ws = torch.randn(B, Q, h, w), torch.randn(B, Q, E)
    want = _ref(x, K, wy, ws)
    _check(_run(x, K, wy, ws), want)
    _check(_run(x, K, wy, ws, channels_last=True), want)


def test_config_b_full_size():
    B, E, Q, h, w = 12, 32, 64, 96, 320
    torch.manual_seed(5)
    x, K = torch.randn(B, E, h, w), 0.3 * torch.randn(B, Q, E)
    wy, ws = torch.randn(B, Q, h, w) * 1e-3, torch.randn(B, Q, E)
    want = _ref(x, K, wy, ws)
    _check(_run(x, K, wy, ws), want)
    _check(_run(x, K, wy, ws, channels_last=True), want)


@pytest.mark.parametrize("B,E,Q,h,w,cl", [(2, 32, 64, 24, 40, True), (2, 32, 24, 12, 20, False), (1, 64, 128, 16, 24, True), (2, 48, 64, 9, 13, True)])
def test_backward_records_max_abs_of_the_feature_gradient(B, E, Q, h, w, cl):
    """sqd_sql_bwd_amax: the backward kernels (both formulations, both layouts, ragged tiles) leave the bit pattern of max |g_x| in the record the
    feature gradient is tagged with — the operand scale of the convolution backward that reads it"""
    from sqd import nnkernels, ops
    nnkernels.amax_enable(True)
    nnkernels.begin_step()
    g = torch.Generator().manual_seed(B * E + Q + h)
    x = torch.randn(B, E, h, w, generator=g).cuda()
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
    x.requires_grad_(True)
    K = torch.randn(B, Q, E, generator=g).cuda().requires_grad_(True)
    y, s = ops.SelfQueryLayer.apply(x, K)
    gx, = torch.autograd.grad((y * torch.randn(y.shape, generator=g).cuda()).sum() + (s * torch.randn(s.shape, generator=g).cuda()).sum(), x)
    rec = nnkernels._amax_get(gx)
    assert rec is not None
    assert float(rec.view(torch.int32).max().view(torch.float32)) == float(gx.abs().max())
