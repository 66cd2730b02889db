"""Pins oracle/torch_ref.py (the CPU restatement) against the golden vectors frozen from the imported
reference (tests/golden/make_goldens.py, groups G1..G16 of SURVEY.md §8c).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import tt
from oracle import torch_ref as O
from param_fill import chain_inputs, decoder_feats, fill_params, pose_input_case, smooth_images, sparse_gt

RTOL, ATOL = 1e-4, 1e-6           # BASELINE.json north_star: "within 1e-4 rel fp32"


def close(a, b, rtol=RTOL, atol=ATOL):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


def test_g01_pose(golden):
    g = golden("g01_pose")
    aa, tr = tt(g["axisangle"]), tt(g["translation"])
    close(O.rot_from_axisangle(aa), g["R"])
    close(O.transformation_from_parameters(aa, tr, False), g["M"])
    close(O.transformation_from_parameters(aa, tr, True), g["M_inv"])


def test_g02_g03_geometry(golden):
    g2, g3 = golden("g02_backproject"), golden("g03_project3d")
    B, H, W = int(g2["B"]), int(g2["H"]), int(g2["W"])
    d = chain_inputs(int(g2["seed"]), B, H, W)
    depth = O.upsample_disp(tt(d["disp"]), H, W)
    assert np.array_equal(depth.numpy(), g2["depth"])
    cam = O.backproject_depth(depth, tt(d["inv_K"]))
    assert np.array_equal(cam.numpy(), g2["cam_points"])          # same ATen ops -> bit-exact
    grid = O.project_3d(cam, tt(d["K"]), tt(g3["T"]), H, W)
    assert np.array_equal(grid.numpy(), g3["grid"])
    x0, y0 = O.grid_sample_indices(grid, H, W)
    assert np.array_equal(x0.numpy(), g3["x0"]) and np.array_equal(y0.numpy(), g3["y0"])


def test_g04_grid_sample(golden):
    g = golden("g04_grid_sample")
    out = torch.nn.functional.grid_sample(tt(g["img"]), tt(g["grid"]), padding_mode="border", align_corners=True)
    close(out, g["out"])
    x0, y0 = O.grid_sample_indices(tt(g["grid"]), 12, 20)
    assert np.array_equal(x0.numpy(), g["x0"]) and np.array_equal(y0.numpy(), g["y0"])


def test_g05_g06_ssim(golden):
    g = golden("g05_ssim")
    x = tt(g["x"]).requires_grad_(True)
    s = O.ssim(x, tt(g["y"]))
    close(s, g["ssim"])
    (s * tt(g["w"])).sum().backward()
    close(x.grad, g["grad_x"], atol=1e-5)
    g = golden("g06_reprojection")
    x = tt(g["x"]).requires_grad_(True)
    r = O.reprojection_loss(x, tt(g["y"]))
    close(r, g["loss"])
    (r * tt(g["w"])).sum().backward()
    close(x.grad, g["grad_pred"], atol=1e-5)


def _chain(d, B, H, W, grad=True, **loss_options):
    disp = tt(d["disp"]).requires_grad_(grad)
    poses = {f: (tt(d["axisangle_s%d" % i]).requires_grad_(grad), tt(d["translation_s%d" % i]).requires_grad_(grad))
             for i, f in enumerate((-1, 1))}
    colors = {0: tt(d["color0"]), -1: tt(d["color_s0"]), 1: tt(d["color_s1"])}
    noise = tt(d["noise"])
    if loss_options.get("avg_reprojection"):
        noise = noise[:, :1].contiguous()
    out = O.photometric_chain(disp, poses, tt(d["K"]), tt(d["inv_K"]), colors, [0, -1, 1], noise, H, W, **loss_options)
    return out, disp, poses


@pytest.mark.parametrize("tag", ["no_ssim", "avg", "no_automask", "avg_no_automask", "no_ssim_avg"])
def test_g23_loss_options(golden, tag):
    """the oracle's restatement of trainer.py:447-451,480-524 against the reference's own compute_losses under each option set"""
    g = golden("g23_loss_options_" + tag)
    B, H, W = int(g["B"]), int(g["H"]), int(g["W"])
    no_ssim, avg, no_mask = (bool(v) for v in g["flags"])
    out, disp, poses = _chain(chain_inputs(int(g["seed"]), B, H, W), B, H, W, no_ssim=no_ssim, avg_reprojection=avg, disable_automasking=no_mask)
    close(out["loss"], g["loss"])
    if not no_mask:
        assert np.array_equal(out["identity_selection/0"].numpy(), g["identity_selection"])
    else:
        assert "identity_selection/0" not in out
    out["loss"].backward()
    close(disp.grad, g["grad_disp"], atol=1e-9)
    for f, n in ((-1, "m1"), (1, "p1")):
        close(poses[f][0].grad, g["grad_axisangle_" + n], atol=1e-7)
        close(poses[f][1].grad, g["grad_translation_" + n], atol=1e-7)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_g07_g08_chain(golden, tag):
    g7, g8 = golden("g07_generate_images_pred_" + tag), golden("g08_compute_losses_" + tag)
    B, H, W = int(g7["B"]), int(g7["H"]), int(g7["W"])
    out, disp, poses = _chain(chain_inputs(int(g7["seed"]), B, H, W), B, H, W)
    close(out[("depth", 0, 0)], g7["depth"])
    for f, n in ((-1, "m1"), (1, "p1")):
        close(out[("sample", f, 0)], g7["sample_" + n], atol=1e-6)
        close(out[("color", f, 0)], g7["color_" + n], atol=1e-5)
        x0, y0 = O.grid_sample_indices(out[("sample", f, 0)].detach(), H, W)
        ok = ~g7["fragile_" + n]
        assert np.array_equal(x0.numpy()[ok], g7["x0_" + n][ok]) and np.array_equal(y0.numpy()[ok], g7["y0_" + n][ok])
    close(out["loss"], g8["loss"])
    assert np.array_equal(out["identity_selection/0"].numpy(), g8["identity_selection"])
    out["loss"].backward()
    close(disp.grad, g8["grad_disp"], atol=1e-9)
    for f, n in ((-1, "m1"), (1, "p1")):
        close(poses[f][0].grad, g8["grad_axisangle_" + n], atol=1e-7)
        close(poses[f][1].grad, g8["grad_translation_" + n], atol=1e-7)


def test_g09_smooth(golden):
    g = golden("g09_smooth")
    d = tt(g["disp"]).requires_grad_(True)
    loss = O.smooth_loss(d, tt(g["img"]))
    close(loss, g["loss"])
    loss.backward()
    close(d.grad, g["grad_disp"], atol=1e-9)


def test_g10_full_query_layer(golden):
    g = golden("g10_full_query_layer")
    x, K = tt(g["x"]).requires_grad_(True), tt(g["K"]).requires_grad_(True)
    y, s = O.full_query_layer(x, K)
    close(y, g["y"], atol=1e-5)
    close(s, g["summary"], atol=1e-5)
    ((y * tt(g["wy"])).sum() + (s * tt(g["ws"])).sum()).backward()
    close(x.grad, g["grad_x"], atol=1e-5)
    close(K.grad, g["grad_K"], atol=1e-4)


@pytest.mark.parametrize("tag,ff", [("full", 1024), ("lite", 512)])
def test_g11_qtr(golden, tag, ff):
    g = golden("g11_qtr_" + tag)
    kw = {k: (float(v) if "." in v else int(v)) for k, v in g["kw"]}
    m = O.QueryTrDecoder(dim_feedforward=ff, **kw)
    fill_params(m, int(g["seed"]))
    m.eval()
    x = tt(np.random.RandomState(int(g["x_seed"])).standard_normal((2, 16, 32, 48)).astype(np.float32)).requires_grad_(True)
    out = m(x)[("disp", 0)]
    close(out, g["disp"], atol=1e-5)
    (out * tt(g["w"])).sum().backward()
    close(x.grad, g["grad_x"], rtol=1e-3, atol=1e-4)
    P = dict(m.named_parameters())
    for k in g:
        if k.startswith("grad__"):
            close(P[k[6:].replace("__", ".")].grad, g[k], rtol=1e-3, atol=1e-4)


def test_g12_posecnn(golden):
    g = golden("g12_posecnn")
    m = fill_params(O.PoseCNN(2), int(g["seed"]))
    x = tt(smooth_images(np.random.RandomState(int(g["x_seed"])), 2, 64, 96, C=6)).requires_grad_(True)
    aa, tr = m(x)
    close(aa, g["axisangle"]); close(tr, g["translation"])
    (aa.sum() * 3 + tr.sum()).backward()
    close(x.grad, g["grad_x"], atol=1e-7)
    close(m.net[0].weight.grad, g["grad_w0"], atol=1e-6)
    close(m.pose_conv.weight.grad, g["grad_pose_conv"], atol=1e-6)


@pytest.mark.parametrize("tag,skips,chans", [("res50", (1024, 512, 256, 64), (64, 256, 512, 1024, 2048)),
                                              ("lite", (256, 128, 64, 64), (64, 64, 128, 256, 512))])
def test_g13_decoderbn(golden, tag, skips, chans):
    g = golden("g13_decoderbn_" + tag)
    dec = fill_params(O.DecoderBN(int(g["nf"]), 8, int(g["bott"]), skips), int(g["seed"]))
    feats = [tt(f).requires_grad_(True) for f in decoder_feats(int(g["feat_seed"]), chans, 32, 48)]
    dec.train()
    out = dec(feats)
    close(out, g["out_train"], atol=1e-5)
    out.square().mean().backward()
    close(dec.up1._net[1].running_mean, g["up1_running_mean_after"])
    close(feats[0].grad, g["grad_feat0"], rtol=1e-3, atol=1e-7)
    close(feats[4].grad, g["grad_feat4"], rtol=1e-3, atol=1e-7)
    close(dec.conv2.weight.grad, g["grad_conv2_w"], rtol=1e-3, atol=1e-7)
    close(dec.up4._net[1].weight.grad, g["grad_up4_bn_w"], rtol=1e-3, atol=1e-7)
    dec.eval()
    close(dec([f.detach() for f in feats]), g["out_eval"], atol=1e-5)


def test_g14_depth_errors(golden):
    g = golden("g14_depth_errors")
    gt = tt(sparse_gt(int(g["gt_seed"]), 2))
    m = O.compute_depth_losses(tt(g["pred"]), gt)
    close(torch.stack([x.double() for x in m]), g["metrics"], rtol=1e-5)


def _models(kind):
    if kind == "res18":
        enc = O.LiteResnetEncoderDecoder(model_dim=16)
        dep = O.QueryTrDecoder(16, 16, 8, 4, 12, 24, min_val=0.001, max_val=80.0, dim_feedforward=512, dropout=0.0)
    else:
        enc = O.ResnetEncoderDecoder(50, 64, 16)
        dep = O.QueryTrDecoder(16, 16, 8, 4, 12, 24, min_val=0.001, max_val=80.0, dim_feedforward=1024, dropout=0.0)
    pose = O.PoseCNN(2)
    fill_params(enc, 1501); fill_params(dep, 1502); fill_params(pose, 1503)
    for m in (enc, dep, pose):
        m.train()
    return enc, dep, pose


def batch_inputs(seed, B, H, W):
    d = chain_inputs(seed, B, H, W)
    rs = np.random.RandomState(seed + 1)
    aug = {k: np.clip(d[k] * rs.uniform(0.9, 1.1) + rs.uniform(-0.03, 0.03), 0, 1).astype(np.float32)
           for k in ("color0", "color_s0", "color_s1")}
    return {("color", 0, 0): tt(d["color0"]), ("color", -1, 0): tt(d["color_s0"]), ("color", 1, 0): tt(d["color_s1"]),
            ("color_aug", 0, 0): tt(aug["color0"]), ("color_aug", -1, 0): tt(aug["color_s0"]),
            ("color_aug", 1, 0): tt(aug["color_s1"]), ("K", 0): tt(d["K"]), ("inv_K", 0): tt(d["inv_K"])}, tt(d["noise"])


@pytest.mark.parametrize("kind", ["res18", "res50"])
def test_g15_g16_train_steps(golden, kind):
    g15, g16 = golden("g15_process_batch_" + kind), golden("g16_adam_steps_" + kind)
    B, H, W = int(g15["B"]), int(g15["H"]), int(g15["W"])
    enc, dep, pose = _models(kind)
    step = O.RefTrainStep(enc, dep, pose, (0, -1, 1), H, W)
    traj = []
    for it in range(3):
        inputs, noise = batch_inputs(1600 + it, B, H, W)
        outputs, losses = step.process_batch(inputs, noise)
        step.optim.zero_grad()
        losses["loss"].backward()
        if it == 0:
            close(outputs[("disp", 0)], g15["disp"], atol=1e-4)
            close(outputs[("depth", 0, 0)], g15["depth"], atol=1e-4)
            close(outputs[("axisangle", 0, -1)], g15["axisangle_m1"], atol=1e-7)
            close(outputs[("translation", 0, 1)], g15["translation_p1"], atol=1e-7)
            close(outputs[("cam_T_cam", 0, -1)], g15["cam_T_cam_m1"], atol=1e-6)
            close(outputs[("color", -1, 0)], g15["color_m1"], atol=1e-4)
            assert (outputs["identity_selection/0"].numpy() != g15["identity_selection"]).mean() < 1e-3
            close(enc.encoder.encoder.conv1.weight.grad, g15["grad_enc_conv1"], rtol=2e-3, atol=1e-7)
            close(enc.decoder.conv3.weight.grad, g15["grad_dec_conv3"], rtol=2e-3, atol=1e-7)
            close(dep.conv3x3.weight.grad, g15["grad_depth_conv3x3"], rtol=2e-3, atol=1e-7)
            close(pose.pose_conv.weight.grad, g15["grad_pose_conv"], rtol=2e-3, atol=1e-7)
            assert enc.encoder.encoder.fc.weight.grad is None and bool(g15["fc_grad_is_none"])
        step.optim.step()
        traj.append(float(losses["loss"]))
    np.testing.assert_allclose(traj, g16["losses"], rtol=1e-4)
    close(enc.encoder.encoder.conv1.weight, g16["enc_conv1_after"], atol=2e-5)
    close(pose.pose_conv.weight, g16["pose_conv_after"], atol=2e-5)
    close(dep.conv3x3.weight, g16["depth_conv3x3_after"], atol=2e-5)


def test_state_dict_keys(golden):
    g = golden("g00_state_dict_keys")
    mods = {"encoder_res50": O.ResnetEncoderDecoder(50, 256, 32), "encoder_res18": O.LiteResnetEncoderDecoder(32),
            "depth": O.QueryTrDecoder(32, 32, 16, 4, 64, 64), "pose": O.PoseCNN(2)}
    for n, m in mods.items():
        mine = ["%s %s" % (k, tuple(v.shape)) for k, v in m.state_dict().items()]
        assert mine == list(g[n]), n


def test_g17_stereo_chain(golden):
    """--use_stereo (reference trainer.py:52-53,405-421): three source frames, "s" through stereo_T, un-scaled pose translations."""
    g = golden("g17_stereo_chain")
    B, H, W = int(g["B"]), int(g["H"]), int(g["W"])
    d = chain_inputs(int(g["seed"]), B, H, W, S=3)
    disp = tt(d["disp"]).requires_grad_(True)
    poses = {f: (tt(d["axisangle_s%d" % i]).requires_grad_(True), tt(d["translation_s%d" % i]).requires_grad_(True))
             for i, f in enumerate((-1, 1))}
    colors = {0: tt(d["color0"]), -1: tt(d["color_s0"]), 1: tt(d["color_s1"]), "s": tt(d["color_s2"])}
    out = O.photometric_chain(disp, poses, tt(d["K"]), tt(d["inv_K"]), colors, [0, -1, 1, "s"], tt(d["noise"]), H, W,
                              stereo_T=tt(g["stereo_T"]), use_stereo=True)
    close(out[("depth", 0, 0)], g["depth"])
    for f, n in ((-1, "m1"), (1, "p1"), ("s", "s")):
        close(out[("sample", f, 0)], g["sample_" + n], atol=1e-6)
        close(out[("color", f, 0)], g["color_" + n], atol=1e-5)
    close(out["loss"], g["loss"])
    assert np.array_equal(out["identity_selection/0"].numpy(), g["identity_selection"])
    out["loss"].backward()
    close(disp.grad, g["grad_disp"], atol=1e-9)
    for f, n in ((-1, "m1"), (1, "p1")):
        close(poses[f][0].grad, g["grad_axisangle_" + n], atol=1e-7)
        close(poses[f][1].grad, g["grad_translation_" + n], atol=1e-7)


def test_g18_decoderbn_b5(golden):
    """DecoderBN with the EfficientNet-b5 skip widths against the reference's own class (networks/base_encoder.py:24-56)"""
    g = golden("g18_decoderbn_b5")
    dec = fill_params(O.DecoderBN(int(g["nf"]), 8, int(g["bott"]), (176, 64, 40, 24)), int(g["seed"]))
    feats = [tt(f).requires_grad_(True) for f in decoder_feats(int(g["feat_seed"]), (24, 40, 64, 176, 2048), 32, 48)]
    dec.train()
    out = dec(feats)
    close(out, g["out_train"], atol=1e-5)
    out.square().mean().backward()
    close(dec.up1._net[1].running_mean, g["up1_running_mean_after"])
    close(feats[0].grad, g["grad_feat0"], rtol=1e-3, atol=1e-7)
    close(feats[4].grad, g["grad_feat4"], rtol=1e-3, atol=1e-7)
    close(dec.conv2.weight.grad, g["grad_conv2_w"], rtol=1e-3, atol=1e-7)
    close(dec.up4._net[1].weight.grad, g["grad_up4_bn_w"], rtol=1e-3, atol=1e-7)
    dec.eval()
    close(dec([f.detach() for f in feats]), g["out_eval"], atol=1e-5)


def test_efficientnet_b5_restatement_shapes_and_keys():
    """the restated trunk: parameter count of EfficientNet-B5 without its classifier (30.39 M - 2.05 M), tap shapes of
    base_encoder.py:41, and the same state-dict keys in the product module and the oracle"""
    import sys
    from conftest import PRODUCT
    sys.path.insert(0, PRODUCT)
    import networks
    ref = O.BaseEncoder(model_dim=32, num_features=512)
    n = sum(p.numel() for p in ref.encoder.original_model.parameters())
    assert n == 28340784, n
    mine = networks.BaseEncoder.build(model_dim=32, num_features=512)
    assert [(k, tuple(v.shape)) for k, v in mine.state_dict().items()] == [(k, tuple(v.shape)) for k, v in ref.state_dict().items()]
    small = O.BaseEncoder(model_dim=16, num_features=64, stages=(("ds", 3, 1, 1, 24, 1), ("ir", 3, 2, 6, 40, 1), ("ir", 5, 2, 6, 64, 1),
                                                                ("ir", 3, 2, 6, 128, 1), ("ir", 5, 1, 6, 176, 1), ("ir", 5, 2, 6, 304, 1),
                                                                ("ir", 3, 1, 6, 512, 1)))
    small.eval()
    x = torch.rand(1, 3, 64, 96)
    f = small.encoder(x)
    assert [tuple(f[i].shape[1:]) for i in (4, 5, 6, 8, 11)] == [(24, 32, 48), (40, 16, 24), (64, 8, 12), (176, 4, 6), (2048, 2, 3)]
    assert small(x).shape == (1, 16, 32, 48)


def test_g19_eval_functions(golden):
    """oracle/eval_ref.py against the reference's compute_errors / batch_post_process_disparity (evaluate_depth_config.py:30-59)"""
    from oracle import eval_ref as R
    g = golden("g19_eval")
    post = R.batch_post_process_disparity(g["l_disp"], g["r_disp"])
    assert post.dtype == np.float64 and np.array_equal(post, g["post"])
    errs = np.array(R.compute_errors(g["gt"], g["pred"]), dtype=np.float64)
    assert np.array_equal(errs, g["errors"])
    assert (R.MIN_DEPTH, R.MAX_DEPTH, R.STEREO_SCALE_FACTOR) == tuple(g["consts"])
    # resize_linear: identity at equal size, exact at integer up-scaling of a linear ramp's interior, constant stays constant
    a = np.random.RandomState(3).uniform(1, 5, (6, 9))
    assert np.array_equal(R.resize_linear(a, 9, 6), a)
    assert np.allclose(R.resize_linear(np.full((4, 5), 2.5), 13, 11), 2.5, rtol=0, atol=1e-6)


def _g20_inputs(g):
    rs = np.random.RandomState(int(g["seeds"][1]))
    sizes = [(3, 5), (7, 11), (14, 22), (28, 44)]
    feats = [(0.5 * rs.standard_normal((2, int(c), h, w))).astype(np.float32) for c, (h, w) in zip(g["enc_ch"], sizes)]
    w = np.random.RandomState(int(g["seeds"][2])).standard_normal(g["out"].shape).astype(np.float32)
    return feats, w


def test_g20_unet_decoder(golden):
    """oracle UnetDecoder against the reference's own (networks/Unet.py:258-312): same state-dict keys, output and gradients"""
    g = golden("g20_unet_decoder")
    dec = O.UnetDecoder([int(c) for c in g["enc_ch"]], tuple(int(c) for c in g["dec_ch"]), 4)
    assert sorted(dec.state_dict().keys()) == list(g["keys"])
    fill_params(dec, int(g["seeds"][0]))
    dec.train()
    feats, w = _g20_inputs(g)
    fr = [tt(f).clone().requires_grad_(True) for f in feats]
    out = dec(fr)
    (out * tt(w)).sum().backward()
    close(out, g["out"], atol=1e-5)
    close(fr[0].grad, g["grad_feat0"], atol=1e-5)
    close(fr[3].grad, g["grad_feat3"], atol=1e-5)
    close(dec.final_conv.weight.grad, g["grad_final_w"], atol=1e-4)
    close(dec.blocks[0].conv1.conv.weight.grad, g["grad_b0c1"], atol=1e-4)


def test_g21_silog(golden):
    """oracle SILogLoss against the reference's own class (finetune/loss.py:24-42)"""
    from oracle import finetune_ref as FR
    g = golden("g21_silog")
    depth = tt(g["depth"])
    mask = depth > 1e-3
    p = tt(g["pred_lr"]).clone().requires_grad_(True)
    loss = FR.SILogLoss()(p, depth, mask=mask, interpolate=True)
    loss.backward()
    assert abs(float(loss) - float(g["loss_interp"])) <= 1e-6 * abs(float(g["loss_interp"]))
    close(p.grad, g["grad_lr"], atol=1e-7)
    q = tt(g["pred_hr"]).clone().requires_grad_(True)
    loss2 = FR.SILogLoss()(q, depth, mask=mask, interpolate=False)
    loss2.backward()
    assert abs(float(loss2) - float(g["loss_plain"])) <= 1e-6 * abs(float(g["loss_plain"]))
    close(q.grad, g["grad_hr"], atol=1e-7)


def test_g22_metric_errors(golden):
    """oracle compute_errors against the reference's own function (finetune/utils.py:76-96), float32 arrays as in its validation loop"""
    from oracle import finetune_ref as FR
    g = golden("g22_metric_errors")
    e = FR.compute_errors(g["gt"], g["pred"])
    for k, v in e.items():
        assert float(v) == float(g[k]), k


@pytest.mark.parametrize("name", ["all", "all_stereo", "pairs_m2_p1"])
def test_g24_pose_input_variants(golden, name):
    """predict_poses -> generate_images_pred -> compute_losses with --pose_model_input all (with and without --use_stereo) and with pairs on
    frame_ids 0 -2 1 (reference trainer.py:301-361, 404-421), against the reference's own run."""
    g = golden("g24_pose_inputs_" + name)
    B, H, W, seed, stereo, mode = int(g["B"]), int(g["H"]), int(g["W"]), int(g["seed"]), bool(g["stereo"]), str(g["mode"])
    fids = [f if f == "s" else int(f) for f in g["frame_ids"]]
    temporal = [f for f in fids if f != "s"]
    _, np_inputs, np_disp, np_noise = pose_input_case(seed, B, H, W, temporal, stereo)
    inputs = {k: tt(v) for k, v in np_inputs.items()}
    pose = fill_params(O.PoseCNN(2 if mode == "pairs" else len(temporal)), seed + 2)

    class _Feed(torch.nn.Module):                          # encoder / depth stand-ins: the fixture feeds disp directly
        def __init__(self, out):
            super().__init__()
            self.out = out

        def forward(self, x):
            return self.out
    disp = tt(np_disp).requires_grad_(True)
    ref = O.RefTrainStep(_Feed(None), _Feed({("disp", 0): disp}), pose, fids, H, W, use_stereo=stereo, pose_model_input=mode)
    out, losses = ref.process_batch(inputs, tt(np_noise))
    f1, f2 = temporal[1], temporal[2]
    close(out[("axisangle", 0, f1)], g["axisangle_f1"], atol=1e-7)
    close(out[("translation", 0, f2)], g["translation_f2"], atol=1e-7)
    close(out[("cam_T_cam", 0, f1)], g["cam_T_cam_f1"])
    close(out[("cam_T_cam", 0, f2)], g["cam_T_cam_f2"])
    for f, n in ((f1, "f1"), (f2, "f2")):
        close(out[("sample", f, 0)], g["sample_" + n], atol=1e-6)
        close(out[("color", f, 0)], g["color_" + n], atol=1e-5)
    close(losses["loss"], g["loss"])
    assert np.array_equal(out["identity_selection/0"].numpy(), g["identity_selection"])
    losses["loss"].backward()
    close(disp.grad, g["grad_disp"], atol=1e-9)
    close(pose.pose_conv.weight.grad, g["grad_pose_conv"], rtol=1e-3, atol=1e-8)
    close(pose.net[0].weight.grad, g["grad_pose_w0"], rtol=1e-3, atol=1e-8)
