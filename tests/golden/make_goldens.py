#!/usr/bin/env python3
"""Golden-vector generator (groups G1..G16 of SURVEY.md §8c).

Runs ONLY in the build container, where the upstream reference is mounted read-only at
/root/reference.  It imports the reference *unmodified* (with sys.modules stubs for the third-party
packages this image lacks — SURVEY App. D), drives its own functions on small seeded inputs and
freezes inputs-by-seed + expected outputs as .npz under tests/golden/.  No reference source or
bytecode is written anywhere; the fixtures are data.

    python tests/golden/make_goldens.py            # regenerates every fixture

The GPU box never runs this file (there is no /root/reference there); tests only read the .npz.
"""
import importlib
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, HERE)
sys.path.insert(0, REPO)
from param_fill import chain_inputs, decoder_feats, fill_params, kitti_K, pose_input_case, smooth_images, sparse_gt  # noqa: E402

from oracle import torch_ref as O  # noqa: E402  (only for the build's ResNet trunk plugged into G15)

torch.set_num_threads(8)


# ----------------------------------------------------------------------------- reference import
def import_reference():
    sys.path.insert(0, REF)

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    stub("kornia"); stub("kornia.geometry"); stub("kornia.geometry.depth", depth_to_3d=None)
    tw = stub("torch.utils.tensorboard.writer", SummaryWriter=object)
    stub("tensorboard"); stub("torch.utils.tensorboard", writer=tw, SummaryWriter=object)

    class _ResNetBase(nn.Module):
        pass

    res = stub("torchvision.models.resnet", BasicBlock=object, Bottleneck=object, model_urls={})
    tvm = stub("torchvision.models", ResNet=_ResNetBase, resnet=res,
               resnet18=lambda pretrained=False: O.ResNetTrunk(18),
               resnet34=lambda pretrained=False: O.ResNetTrunk(34),
               resnet50=lambda pretrained=False: O.ResNetTrunk(50),
               resnet101=None, resnet152=None)
    tvt = stub("torchvision.transforms", ToTensor=lambda: None, ColorJitter=object, Resize=object,
               ToPILImage=object)
    stub("torchvision", models=tvm, transforms=tvt)
    stub("timm", create_model=None)
    stub("skimage"); stub("skimage.transform")
    stub("cv2", setNumThreads=lambda n: None, ocl=types.SimpleNamespace(setUseOpenCL=lambda b: None))
    import trainer as T                                    # the reference's own trainer module
    torch.Tensor.cuda = lambda self, *a, **k: self         # trainer.py:506,517 call .cuda() unconditionally
    nets = {n: importlib.import_module("networks." + n) for n in
            ("layers", "depth_decoder_QTR", "lite_depth_decoder_QTR", "pose_cnn", "resnet_encoder",
             "lite_res_encoder")}
    return T, nets


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-28s %8.1f KB  %s" % (name, os.path.getsize(path) / 1024, sorted(out)))


def tt(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def make_shim(T, B, H, W, frame_ids=(0, -1, 1)):
    opt = types.SimpleNamespace(scales=[0], frame_ids=list(frame_ids), height=H, width=W,
                                pose_model_type="posecnn", pose_model_input="pairs", use_stereo=False,
                                disable_automasking=False, no_ssim=False, avg_reprojection=False,
                                predictive_mask=False, disparity_smoothness=1e-3, v1_multiscale=False,
                                batch_size=B, min_depth=0.001, max_depth=80.0)
    shim = types.SimpleNamespace(opt=opt, num_scales=1, ssim=T.SSIM(), device=torch.device("cpu"),
                                 backproject_depth={0: T.BackprojectDepth(B, H, W)},
                                 project_3d={0: T.Project3D(B, H, W)}, num_pose_frames=2,
                                 use_pose_net=True,
                                 depth_metric_names=["de/abs_rel", "de/sq_rel", "de/rms", "de/log_rms",
                                                     "da/a1", "da/a2", "da/a3"])
    shim.compute_reprojection_loss = lambda p, t: T.Trainer.compute_reprojection_loss(shim, p, t)
    shim.predict_poses = lambda i, f: T.Trainer.predict_poses(shim, i, f)
    shim.generate_images_pred = lambda i, o: T.Trainer.generate_images_pred(shim, i, o)
    shim.compute_losses = lambda i, o: T.Trainer.compute_losses(shim, i, o)
    return shim


class patched_randn:
    """Replace torch.randn (trainer.py:516) by a recorded tensor for the duration of a call."""

    def __init__(self, noise):
        self.noise = noise

    def __enter__(self):
        self._orig = torch.randn
        torch.randn = lambda *a, **k: self.noise.clone()

    def __exit__(self, *exc):
        torch.randn = self._orig


def zero_dropout(m):
    for mod in m.modules():
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, nn.MultiheadAttention):
            mod.dropout = 0.0
    return m


# ----------------------------------------------------------------------------- groups
def g1_pose(T):
    rs = np.random.RandomState(101)
    aa = (0.3 * rs.standard_normal((6, 1, 3))).astype(np.float32)
    aa[0] = 0.0                       # angle == 0 -> axis = 0/(0+1e-7)
    aa[1] *= 1e-4                     # tiny angle
    tr = rs.standard_normal((6, 1, 3)).astype(np.float32)
    M0 = T.transformation_from_parameters(tt(aa), tt(tr), invert=False)
    M1 = T.transformation_from_parameters(tt(aa), tt(tr), invert=True)
    R = T.rot_from_axisangle(tt(aa))
    save("g01_pose", axisangle=aa, translation=tr, M=M0, M_inv=M1, R=R)


def g2_g3_g4_geometry(T):
    B, H, W = 2, 24, 80
    d = chain_inputs(202, B, H, W)
    depth = torch.nn.functional.interpolate(tt(d["disp"]), [H, W], mode="bilinear", align_corners=False)
    bp, pj = T.BackprojectDepth(B, H, W), T.Project3D(B, H, W)
    cam = bp(depth, tt(d["inv_K"]))
    Tm = T.transformation_from_parameters(tt(d["axisangle_s0"][:, 0]), tt(d["translation_s0"][:, 0]) * 0.09, True)
    grid = pj(cam, tt(d["K"]), Tm)
    x0, y0 = O.grid_sample_indices(grid, H, W)
    ix = ((grid[..., 0] + 1) / 2) * (W - 1)
    iy = ((grid[..., 1] + 1) / 2) * (H - 1)
    fragile = ((ix - ix.round()).abs() < 1e-4 * (1 + ix.abs())) | ((iy - iy.round()).abs() < 1e-4 * (1 + iy.abs()))
    P = torch.matmul(tt(d["K"]), Tm)[:, :3, :]
    save("g02_backproject", seed=202, B=B, H=H, W=W, depth=depth, cam_points=cam)
    save("g03_project3d", seed=202, B=B, H=H, W=W, depth=depth, T=Tm, P=P, grid=grid, x0=x0, y0=y0,
         fragile=fragile)
    # G4: border cases of grid_sample(padding border, align_corners=True)
    rs = np.random.RandomState(404)
    img = smooth_images(rs, 2, 12, 20)
    g = rs.uniform(-1.6, 1.6, (2, 12, 20, 2)).astype(np.float32)
    g[0, 0, :5, 0] = [-1.0, 1.0, -1.0000001, 1.0000001, 0.0]
    g[0, 0, :5, 1] = [-1.0, 1.0, 1.0, -1.0, 0.0]
    g[0, 1, :4, 0] = [-3.0, 3.0, 0.5, -0.5]
    g[0, 1, :4, 1] = [0.25, -0.25, 3.0, -3.0]
    out = torch.nn.functional.grid_sample(tt(img), tt(g), padding_mode="border", align_corners=True)
    gx0, gy0 = O.grid_sample_indices(tt(g), 12, 20)
    save("g04_grid_sample", img=img, grid=g, out=out, x0=gx0, y0=gy0)


def g5_g6_ssim(T):
    rs = np.random.RandomState(505)
    x = smooth_images(rs, 2, 20, 36)
    y = np.clip(x + 0.1 * rs.standard_normal(x.shape), 0, 1).astype(np.float32)
    y[1] = x[1]                                    # identical images -> SSIM loss exactly clamps at 0
    xt = tt(x).requires_grad_(True)
    s = T.SSIM()(xt, tt(y))
    w = tt(rs.uniform(0.5, 1.5, s.shape).astype(np.float32))
    (s * w).sum().backward()
    save("g05_ssim", x=x, y=y, ssim=s, w=w, grad_x=xt.grad)
    shim = make_shim(T, 2, 20, 36)
    xt2 = tt(x).requires_grad_(True)
    r = T.Trainer.compute_reprojection_loss(shim, xt2, tt(y))
    w2 = tt(rs.uniform(0.5, 1.5, r.shape).astype(np.float32))
    (r * w2).sum().backward()
    save("g06_reprojection", x=x, y=y, loss=r, w=w2, grad_pred=xt2.grad)


def run_chain(T, d, B, H, W, with_grad=True, **opt_flags):
    shim = make_shim(T, B, H, W)
    for k, v in opt_flags.items():          # the reference's loss options: no_ssim, avg_reprojection, disable_automasking
        assert hasattr(shim.opt, k)
        setattr(shim.opt, k, v)
    disp = tt(d["disp"]).requires_grad_(with_grad)
    aa = {f: tt(d["axisangle_s%d" % i]).requires_grad_(with_grad) for i, f in enumerate((-1, 1))}
    tr = {f: tt(d["translation_s%d" % i]).requires_grad_(with_grad) for i, f in enumerate((-1, 1))}
    inputs = {("color", 0, 0): tt(d["color0"]), ("color", -1, 0): tt(d["color_s0"]),
              ("color", 1, 0): tt(d["color_s1"]), ("K", 0): tt(d["K"]), ("inv_K", 0): tt(d["inv_K"])}
    outputs = {("disp", 0): disp}
    for f in (-1, 1):
        outputs[("axisangle", 0, f)] = aa[f]
        outputs[("translation", 0, f)] = tr[f]
        outputs[("cam_T_cam", 0, f)] = T.transformation_from_parameters(aa[f][:, 0], tr[f][:, 0], invert=(f < 0))
    T.Trainer.generate_images_pred(shim, inputs, outputs)
    noise = tt(d["noise"])
    if shim.opt.avg_reprojection:
        noise = noise[:, :1].contiguous()     # torch.randn(identity_reprojection_loss.shape): one channel under avg_reprojection (:490,:516)
    with patched_randn(noise):
        losses = T.Trainer.compute_losses(shim, inputs, outputs)
    return shim, inputs, outputs, losses, disp, aa, tr


def g23_loss_options(T):
    """the reference's loss options (trainer.py:447-451, 480-524) through its own compute_losses: loss, identity_selection (with
    automasking) and the gradients w.r.t. disparity and poses, per option set"""
    B, H, W = 2, 24, 80
    d = chain_inputs(723, B, H, W)
    for tag, flags in (("no_ssim", dict(no_ssim=True)), ("avg", dict(avg_reprojection=True)), ("no_automask", dict(disable_automasking=True)),
                       ("avg_no_automask", dict(avg_reprojection=True, disable_automasking=True)),
                       ("no_ssim_avg", dict(no_ssim=True, avg_reprojection=True))):
        shim, inputs, outputs, losses, disp, aa, tr = run_chain(T, d, B, H, W, **flags)
        losses["loss"].backward()
        extra = {} if flags.get("disable_automasking") else {"identity_selection": outputs["identity_selection/0"]}
        save("g23_loss_options_" + tag, seed=723, B=B, H=H, W=W, loss=losses["loss"], grad_disp=disp.grad,
             grad_axisangle_m1=aa[-1].grad, grad_axisangle_p1=aa[1].grad, grad_translation_m1=tr[-1].grad, grad_translation_p1=tr[1].grad,
             flags=np.array([int(flags.get("no_ssim", False)), int(flags.get("avg_reprojection", False)), int(flags.get("disable_automasking", False))]),
             **extra)


def g7_g8_chain(T):
    for tag, seed, B, H, W in (("a", 707, 2, 24, 80), ("b", 708, 2, 48, 160)):
        d = chain_inputs(seed, B, H, W)
        shim, inputs, outputs, losses, disp, aa, tr = run_chain(T, d, B, H, W)
        losses["loss"].backward()
        x0 = {}; frag = {}
        for f in (-1, 1):
            g = outputs[("sample", f, 0)].detach()
            x0[f] = O.grid_sample_indices(g, H, W)
            ix = ((g[..., 0] + 1) / 2) * (W - 1); iy = ((g[..., 1] + 1) / 2) * (H - 1)
            frag[f] = ((ix - ix.round()).abs() < 1e-3) | ((iy - iy.round()).abs() < 1e-3)
        save("g07_generate_images_pred_" + tag, seed=seed, B=B, H=H, W=W,
             depth=outputs[("depth", 0, 0)], sample_m1=outputs[("sample", -1, 0)], sample_p1=outputs[("sample", 1, 0)],
             color_m1=outputs[("color", -1, 0)], color_p1=outputs[("color", 1, 0)],
             x0_m1=x0[-1][0], y0_m1=x0[-1][1], x0_p1=x0[1][0], y0_p1=x0[1][1],
             fragile_m1=frag[-1], fragile_p1=frag[1])
        save("g08_compute_losses_" + tag, seed=seed, B=B, H=H, W=W, loss=losses["loss"],
             identity_selection=outputs["identity_selection/0"], grad_disp=disp.grad,
             grad_axisangle_m1=aa[-1].grad, grad_axisangle_p1=aa[1].grad,
             grad_translation_m1=tr[-1].grad, grad_translation_p1=tr[1].grad)


def g17_stereo_chain(T):
    """--use_stereo: frame_ids [0,-1,1,"s"] (trainer.py:52-53), T of "s" = inputs["stereo_T"] (:406-407), no mean-inverse-depth
    scaling of the pose translations (:412): generate_images_pred + compute_losses with three source frames."""
    seed, B, H, W = 1717, 2, 32, 96
    d = chain_inputs(seed, B, H, W, S=3)
    shim = make_shim(T, B, H, W, frame_ids=(0, -1, 1, "s"))
    shim.opt.use_stereo = True
    disp = tt(d["disp"]).requires_grad_(True)
    aa = {f: tt(d["axisangle_s%d" % i]).requires_grad_(True) for i, f in enumerate((-1, 1))}
    tr = {f: tt(d["translation_s%d" % i]).requires_grad_(True) for i, f in enumerate((-1, 1))}
    stereo_T = torch.eye(4).repeat(B, 1, 1)
    stereo_T[0, 0, 3], stereo_T[1, 0, 3] = -0.1, 0.1          # left / right target camera (datasets/mono_dataset.py:193-199)
    inputs = {("color", 0, 0): tt(d["color0"]), ("color", -1, 0): tt(d["color_s0"]), ("color", 1, 0): tt(d["color_s1"]),
              ("color", "s", 0): tt(d["color_s2"]), ("K", 0): tt(d["K"]), ("inv_K", 0): tt(d["inv_K"]), "stereo_T": stereo_T}
    outputs = {("disp", 0): disp}
    for f in (-1, 1):
        outputs[("axisangle", 0, f)] = aa[f]
        outputs[("translation", 0, f)] = tr[f]
        outputs[("cam_T_cam", 0, f)] = T.transformation_from_parameters(aa[f][:, 0], tr[f][:, 0], invert=(f < 0))
    T.Trainer.generate_images_pred(shim, inputs, outputs)
    with patched_randn(tt(d["noise"])):
        losses = T.Trainer.compute_losses(shim, inputs, outputs)
    losses["loss"].backward()
    save("g17_stereo_chain", seed=seed, B=B, H=H, W=W, stereo_T=stereo_T, depth=outputs[("depth", 0, 0)],
         sample_m1=outputs[("sample", -1, 0)], sample_p1=outputs[("sample", 1, 0)], sample_s=outputs[("sample", "s", 0)],
         color_m1=outputs[("color", -1, 0)], color_p1=outputs[("color", 1, 0)], color_s=outputs[("color", "s", 0)],
         loss=losses["loss"], identity_selection=outputs["identity_selection/0"], grad_disp=disp.grad,
         grad_axisangle_m1=aa[-1].grad, grad_axisangle_p1=aa[1].grad, grad_translation_m1=tr[-1].grad, grad_translation_p1=tr[1].grad)


def g24_pose_inputs(T):
    """predict_poses -> generate_images_pred -> compute_losses of the reference itself for the pose-input variants of trainer.py:301-361:
    "all" (ONE pass of PoseCNN(3) over the concatenated frames; without a stereo frame :414-421 rebuild T of EVERY frame from pose 0 of the
    shared tensors, with --use_stereo T = cam_T_cam of pose i) and pairs on temporal offsets other than +-1 (frame_ids 0 -2 1)."""
    nets = {"pose_cnn": importlib.import_module("networks.pose_cnn")}
    B, H, W = 2, 64, 96
    for name, frame_ids, mode, stereo in (("all", (0, -1, 1), "all", False), ("all_stereo", (0, -1, 1), "all", True),
                                          ("pairs_m2_p1", (0, -2, 1), "pairs", False)):
        seed = 2400 + len(name)
        fids, np_inputs, np_disp, np_noise = pose_input_case(seed, B, H, W, frame_ids, stereo)
        shim = make_shim(T, B, H, W, frame_ids=fids)
        shim.opt.use_stereo, shim.opt.pose_model_input = stereo, mode
        shim.num_pose_frames = 2 if mode == "pairs" else len(frame_ids)
        pose = fill_params(nets["pose_cnn"].PoseCNN(shim.num_pose_frames), seed + 2)
        shim.models = {"pose": pose}
        inputs = {k: tt(v) for k, v in np_inputs.items()}
        stereo_T = tt(np_inputs["stereo_T"]) if stereo else torch.eye(4).repeat(B, 1, 1)
        disp = tt(np_disp).requires_grad_(True)
        outputs = {("disp", 0): disp}
        outputs.update(T.Trainer.predict_poses(shim, inputs, None))
        T.Trainer.generate_images_pred(shim, inputs, outputs)
        with patched_randn(tt(np_noise)):
            losses = T.Trainer.compute_losses(shim, inputs, outputs)
        losses["loss"].backward()
        f1, f2 = frame_ids[1], frame_ids[2]
        save("g24_pose_inputs_" + name, seed=seed, B=B, H=H, W=W, stereo=np.array(stereo), mode=np.array(mode),
             frame_ids=np.array([str(f) for f in fids]), stereo_T=stereo_T,
             axisangle_f1=outputs[("axisangle", 0, f1)], translation_f2=outputs[("translation", 0, f2)],
             cam_T_cam_f1=outputs[("cam_T_cam", 0, f1)], cam_T_cam_f2=outputs[("cam_T_cam", 0, f2)],
             sample_f1=outputs[("sample", f1, 0)], sample_f2=outputs[("sample", f2, 0)],
             color_f1=outputs[("color", f1, 0)], color_f2=outputs[("color", f2, 0)], loss=losses["loss"],
             identity_selection=outputs["identity_selection/0"], grad_disp=disp.grad,
             grad_pose_conv=pose.pose_conv.weight.grad, grad_pose_w0=pose.net[0].weight.grad)


def g9_smooth(T):
    rs = np.random.RandomState(909)
    img = smooth_images(rs, 2, 20, 36)
    disp = rs.uniform(0.5, 2.0, (2, 1, 20, 36)).astype(np.float32)
    dt = tt(disp).requires_grad_(True)
    loss = T.get_smooth_loss(dt, tt(img))
    loss.backward()
    save("g09_smooth", disp=disp, img=img, loss=loss, grad_disp=dt.grad)


def g10_sql(nets):
    rs = np.random.RandomState(1010)
    x = rs.standard_normal((2, 16, 12, 20)).astype(np.float32)
    K = (0.5 * rs.standard_normal((2, 24, 16))).astype(np.float32)
    xt, Kt = tt(x).requires_grad_(True), tt(K).requires_grad_(True)
    y, summ = nets["layers"].FullQueryLayer()(xt, Kt)
    wy = tt(rs.standard_normal(y.shape).astype(np.float32))
    ws = tt(rs.standard_normal(summ.shape).astype(np.float32))
    ((y * wy).sum() + (summ * ws).sum()).backward()
    save("g10_full_query_layer", x=x, K=K, y=y, summary=summ, wy=wy, ws=ws, grad_x=xt.grad, grad_K=Kt.grad)


def g11_qtr(nets):
    for tag, mod, cls in (("full", "depth_decoder_QTR", "Depth_Decoder_QueryTr"),
                          ("lite", "lite_depth_decoder_QTR", "Lite_Depth_Decoder_QueryTr")):
        kw = dict(in_channels=16, embedding_dim=16, patch_size=8, num_heads=4, query_nums=12, dim_out=24,
                  min_val=0.001, max_val=80.0)
        m = getattr(nets[mod], cls)(**kw)
        fill_params(m, 1111)
        m.eval()
        rs = np.random.RandomState(1112)
        x = rs.standard_normal((2, 16, 32, 48)).astype(np.float32)
        xt = tt(x).requires_grad_(True)
        out = m(xt)[("disp", 0)]
        w = tt(rs.standard_normal(out.shape).astype(np.float32))
        (out * w).sum().backward()
        grads = {("grad__" + k.replace(".", "__")): p.grad for k, p in m.named_parameters()
                 if k in ("conv3x3.weight", "bins_regressor.4.bias", "convert_to_prob.0.weight",
                          "embedding_convPxP.weight", "positional_encodings")}
        save("g11_qtr_" + tag, seed=1111, x_seed=1112, disp=out, w=w, grad_x=xt.grad,
             kw=np.array(sorted(kw.items()), dtype=object).astype(str), **grads)


def g12_pose(nets):
    m = fill_params(nets["pose_cnn"].PoseCNN(2), 1212)
    rs = np.random.RandomState(1213)
    x = smooth_images(rs, 2, 64, 96, C=6)
    xt = tt(x).requires_grad_(True)
    aa, tr = m(xt)
    (aa.sum() * 3 + tr.sum()).backward()
    save("g12_posecnn", seed=1212, x_seed=1213, axisangle=aa, translation=tr, grad_x=xt.grad,
         grad_w0=m.net[0].weight.grad, grad_pose_conv=m.pose_conv.weight.grad)


def g13_decoder(nets):
    rs = np.random.RandomState(1313)
    for tag, mod, nf, bott, chans in (("res50", "resnet_encoder", 64, 2048, (64, 256, 512, 1024, 2048)),
                                      ("lite", "lite_res_encoder", 256, 512, (64, 64, 128, 256, 512))):
        dec = nets[mod].DecoderBN(num_features=nf, num_classes=8, bottleneck_features=bott)
        fill_params(dec, 1314)
        feats = decoder_feats(1315, chans, 32, 48)
        fts = [tt(f).requires_grad_(True) for f in feats]
        dec.train()
        out_tr = dec(fts)
        out_tr.square().mean().backward()
        rm = dec.up1._net[1].running_mean.clone()
        dec.eval()
        out_ev = dec([tt(f) for f in feats])
        save("g13_decoderbn_" + tag, seed=1314, feat_seed=1315, nf=nf, bott=bott, out_train=out_tr, out_eval=out_ev,
             up1_running_mean_after=rm, grad_feat0=fts[0].grad, grad_feat4=fts[4].grad,
             grad_conv2_w=dec.conv2.weight.grad, grad_up4_bn_w=dec.up4._net[1].weight.grad)


def g18_decoder_b5(T):
    """DecoderBN of the EfficientNet-b5 encoder (reference networks/base_encoder.py:24-56: skip widths +176, +64, +40, +24, taps
    features[4,5,6,8,11]) — base_encoder.py imports torch only, so the reference's own class runs here."""
    sys.path.insert(0, "/root/reference/networks")
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_base_encoder", "/root/reference/networks/base_encoder.py")
    be = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(be)
    dec = be.DecoderBN(num_features=64, num_classes=8, bottleneck_features=2048)
    fill_params(dec, 1818)
    chans = (24, 40, 64, 176, 2048)
    feats = decoder_feats(1819, chans, 32, 48)
    fts = [tt(f).requires_grad_(True) for f in feats]
    flist = [None] * 12
    for i, f in zip((4, 5, 6, 8, 11), fts):
        flist[i] = f
    dec.train()
    out_tr = dec(flist)
    out_tr.square().mean().backward()
    rm = dec.up1._net[1].running_mean.clone()
    dec.eval()
    out_ev = dec([None if f is None else f.detach() for f in flist])
    save("g18_decoderbn_b5", seed=1818, feat_seed=1819, nf=64, bott=2048, out_train=out_tr, out_eval=out_ev,
         up1_running_mean_after=rm, grad_feat0=fts[0].grad, grad_feat4=fts[4].grad,
         grad_conv2_w=dec.conv2.weight.grad, grad_up4_bn_w=dec.up4._net[1].weight.grad)


def g14_depth_errors(T):
    rs = np.random.RandomState(1414)
    B = 2
    gt = sparse_gt(1415, B)
    pred = rs.uniform(2, 60, (B, 1, 24, 80)).astype(np.float32)
    errs = T.compute_depth_errors(tt(gt[gt > 0]), tt(rs.uniform(1, 80, (gt > 0).sum()).astype(np.float32)))
    shim = make_shim(T, B, 24, 80)
    losses = {}
    T.Trainer.compute_depth_losses(shim, {"depth_gt": tt(gt)}, {("depth", 0, 0): tt(pred)}, losses)
    save("g14_depth_errors", gt_seed=1415, pred=pred, plain_errors=np.array([float(e) for e in errs]), metrics=np.array([float(losses[k]) for k in shim.depth_metric_names], dtype=np.float64))


def g19_eval(T):
    """evaluate_depth_config.py: compute_errors (:30-47) and batch_post_process_disparity (:50-59), the reference's own functions"""
    import evaluate_depth_config as EV
    rs = np.random.RandomState(1919)
    l_disp = rs.uniform(0.5, 60, (3, 24, 80)).astype(np.float32)
    r_disp = rs.uniform(0.5, 60, (3, 24, 80)).astype(np.float32)
    post = EV.batch_post_process_disparity(l_disp, r_disp)
    gt = rs.uniform(1.0, 80.0, 5000).astype(np.float32)
    pred = np.clip(gt.astype(np.float64) * rs.uniform(0.6, 1.6, 5000), 1e-3, 80.0)
    errs = np.array(EV.compute_errors(gt, pred), dtype=np.float64)
    save("g19_eval", l_disp=l_disp, r_disp=r_disp, post=post.astype(np.float64), gt=gt, pred=pred, errors=errs,
         consts=np.array([1e-3, 80.0, EV.STEREO_SCALE_FACTOR], dtype=np.float64))


def g20_unet_decoder(T):
    """the reference's own UnetDecoder (networks/Unet.py:258-312) on seeded feature maps: output + gradients"""
    import importlib
    U = importlib.import_module("networks.Unet")
    torch.manual_seed(2020)
    enc_ch, dec_ch = [48, 24, 16, 8], (32, 24, 16, 8)
    dec = U.UnetDecoder(encoder_channels=enc_ch, decoder_channels=dec_ch, final_channels=4, norm_layer=torch.nn.BatchNorm2d, center=False)
    fill_params(dec, 2021)
    dec.train()
    rs = np.random.RandomState(2022)
    sizes = [(3, 5), (7, 11), (14, 22), (28, 44)]            # head first; skips need not be exact doublings (interpolate to the skip's size)
    feats = [(0.5 * rs.standard_normal((2, c, h, w))).astype(np.float32) for c, (h, w) in zip(enc_ch, sizes)]      # (batch 2: BatchNorm statistics)
    fr = [tt(f).clone().requires_grad_(True) for f in feats]
    out = dec(fr)
    w = tt(np.random.RandomState(2023).standard_normal(tuple(out.shape)).astype(np.float32))
    (out * w).sum().backward()
    save("g20_unet_decoder", out=out.detach().numpy(), enc_ch=np.array(enc_ch), dec_ch=np.array(dec_ch), seeds=np.array([2021, 2022, 2023]),
         grad_feat0=fr[0].grad.numpy(), grad_feat3=fr[3].grad.numpy(),
         grad_final_w=dec.final_conv.weight.grad.numpy(), grad_b0c1=dec.blocks[0].conv1.conv.weight.grad.numpy(),
         keys=np.array(sorted(dec.state_dict().keys())))


def g21_silog(T):
    """the reference's SILogLoss (finetune/loss.py:24-42): loss and input gradient, with and without interpolation"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_ft_loss", os.path.join(REF, "finetune", "loss.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    crit = mod.SILogLoss()
    rs = np.random.RandomState(2121)
    depth = rs.uniform(1.0, 80.0, (2, 1, 24, 40)).astype(np.float32)
    depth[rs.uniform(size=depth.shape) > 0.4] = 0.0
    pred_lr = rs.uniform(0.5, 60.0, (2, 1, 12, 20)).astype(np.float32)
    p = tt(pred_lr).clone().requires_grad_(True)
    mask = tt(depth) > 1e-3
    loss = crit(p, tt(depth), mask=mask, interpolate=True)
    loss.backward()
    pred_hr = rs.uniform(0.5, 60.0, (2, 1, 24, 40)).astype(np.float32)
    q = tt(pred_hr).clone().requires_grad_(True)
    loss2 = crit(q, tt(depth), mask=mask, interpolate=False)
    loss2.backward()
    save("g21_silog", depth=depth, pred_lr=pred_lr, pred_hr=pred_hr, loss_interp=np.float32(loss.item()), grad_lr=p.grad.numpy(),
         loss_plain=np.float32(loss2.item()), grad_hr=q.grad.numpy())


def g22_metric_errors(T):
    """the reference's compute_errors (finetune/utils.py:76-96) on float32 arrays, as its validation loop calls it"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_ft_utils", os.path.join(REF, "finetune", "utils.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    rs = np.random.RandomState(2222)
    gt = rs.uniform(1.0, 80.0, 5000).astype(np.float32)
    pred = (gt * np.exp(rs.normal(0.0, 0.25, 5000))).astype(np.float32)
    pred = np.clip(pred, 1e-3, 80.0).astype(np.float32)
    e = mod.compute_errors(gt, pred)
    save("g22_metric_errors", gt=gt, pred=pred, **{k: np.float64(v) for k, v in e.items()})


def build_reference_models(nets, kind):
    if kind == "res18":
        enc = nets["lite_res_encoder"].LiteResnetEncoderDecoder(model_dim=16)
        dep = nets["lite_depth_decoder_QTR"].Lite_Depth_Decoder_QueryTr(
            in_channels=16, patch_size=8, dim_out=24, embedding_dim=16, query_nums=12, num_heads=4,
            min_val=0.001, max_val=80.0)
    else:
        enc = nets["resnet_encoder"].ResnetEncoderDecoder(num_layers=50, num_features=64, model_dim=16)
        dep = nets["depth_decoder_QTR"].Depth_Decoder_QueryTr(
            in_channels=16, patch_size=8, dim_out=24, embedding_dim=16, query_nums=12, num_heads=4,
            min_val=0.001, max_val=80.0)
    pose = nets["pose_cnn"].PoseCNN(2)
    fill_params(enc, 1501); fill_params(dep, 1502); fill_params(pose, 1503)
    for m in (enc, dep, pose):
        m.train(); zero_dropout(m)
    return enc, dep, pose


def batch_inputs(seed, B, H, W):
    d = chain_inputs(seed, B, H, W)
    rs = np.random.RandomState(seed + 1)
    aug = {k: np.clip(d[k] * rs.uniform(0.9, 1.1) + rs.uniform(-0.03, 0.03), 0, 1).astype(np.float32)
           for k in ("color0", "color_s0", "color_s1")}
    return {("color", 0, 0): tt(d["color0"]), ("color", -1, 0): tt(d["color_s0"]), ("color", 1, 0): tt(d["color_s1"]),
            ("color_aug", 0, 0): tt(aug["color0"]), ("color_aug", -1, 0): tt(aug["color_s0"]),
            ("color_aug", 1, 0): tt(aug["color_s1"]), ("K", 0): tt(d["K"]), ("inv_K", 0): tt(d["inv_K"])}, tt(d["noise"])


def g15_g16_step(T, nets):
    B, H, W = 2, 64, 96
    for kind in ("res18", "res50"):
        enc, dep, pose = build_reference_models(nets, kind)
        shim = make_shim(T, B, H, W)
        shim.models = {"encoder": enc, "depth": dep, "pose": pose}
        params = [p for m in (enc, dep, pose) for p in m.parameters()]
        optim = torch.optim.Adam(params, 1e-4)                    # trainer.py:133
        traj, first = [], None
        for it in range(3):
            inputs, noise = batch_inputs(1600 + it, B, H, W)
            with patched_randn(noise):
                outputs, losses = T.Trainer.process_batch(shim, dict(inputs))
            optim.zero_grad()
            losses["loss"].backward()
            if it == 0:
                first = dict(
                    disp=outputs[("disp", 0)], depth=outputs[("depth", 0, 0)],
                    axisangle_m1=outputs[("axisangle", 0, -1)], translation_p1=outputs[("translation", 0, 1)],
                    cam_T_cam_m1=outputs[("cam_T_cam", 0, -1)], color_m1=outputs[("color", -1, 0)],
                    identity_selection=outputs["identity_selection/0"],
                    out_keys=np.array(sorted(str(k) for k in outputs.keys())),
                    loss_keys=np.array(sorted(losses.keys())),
                    grad_enc_conv1=enc.encoder.encoder.conv1.weight.grad,
                    grad_dec_conv3=enc.decoder.conv3.weight.grad,
                    grad_depth_conv3x3=dep.conv3x3.weight.grad,
                    grad_pose_conv=pose.pose_conv.weight.grad,
                    fc_grad_is_none=np.array(enc.encoder.encoder.fc.weight.grad is None))
            optim.step()
            traj.append(float(losses["loss"]))
        save("g15_process_batch_" + kind, B=B, H=H, W=W, loss=traj[0], **first)
        save("g16_adam_steps_" + kind, B=B, H=H, W=W, losses=np.array(traj, dtype=np.float64),
             enc_conv1_after=enc.encoder.encoder.conv1.weight, pose_conv_after=pose.pose_conv.weight,
             depth_conv3x3_after=dep.conv3x3.weight)


def state_dict_keys(nets):
    enc50 = nets["resnet_encoder"].ResnetEncoderDecoder(num_layers=50, num_features=256, model_dim=32)
    enc18 = nets["lite_res_encoder"].LiteResnetEncoderDecoder(model_dim=32)
    dep = nets["depth_decoder_QTR"].Depth_Decoder_QueryTr(in_channels=32, patch_size=16, dim_out=64,
                                                          embedding_dim=32, query_nums=64, num_heads=4)
    pose = nets["pose_cnn"].PoseCNN(2)
    recs = {}
    for n, m in (("encoder_res50", enc50), ("encoder_res18", enc18), ("depth", dep), ("pose", pose)):
        recs[n] = np.array(["%s %s" % (k, tuple(v.shape)) for k, v in m.state_dict().items()])
    save("g00_state_dict_keys", **recs)


def options_spec():
    import options as ROPT
    p = ROPT.MonodepthOptions().parser
    rows = []
    for a in p._actions:
        if not a.option_strings:
            continue
        rows.append("|".join([a.option_strings[0], type(a).__name__, getattr(a.type, "__name__", str(a.type)),
                              repr(a.default), repr(a.nargs), repr(a.choices)]))
    args_files = {}
    for root, _, files in os.walk(os.path.join(REF, "args_files")):
        for f in files:
            if f.endswith(".txt"):
                rel = os.path.relpath(os.path.join(root, f), REF)
                try:
                    ns = ROPT.MonodepthOptions().parser.parse_args(open(os.path.join(root, f)).read().split())
                    args_files[rel] = repr(sorted(vars(ns).items()))
                except SystemExit:
                    args_files[rel] = "ARGPARSE_ERROR"
    tokens = {}
    for rel in args_files:
        tokens[rel] = " ".join(open(os.path.join(REF, rel)).read().split())     # the flag tokens = parser INPUT
    save("g00_options_spec", rows=np.array(rows), files=np.array(sorted(args_files)),
         parsed=np.array([args_files[k] for k in sorted(args_files)]),
         tokens=np.array([tokens[k] for k in sorted(args_files)]))


def main():
    T, nets = import_reference()
    if len(sys.argv) > 1:                  # regenerate single groups: python make_goldens.py g17_stereo_chain ...
        for name in sys.argv[1:]:
            globals()[name](T)
        return
    g17_stereo_chain(T)
    g18_decoder_b5(T)
    g19_eval(T)
    g20_unet_decoder(T)
    g21_silog(T)
    g22_metric_errors(T)
    g23_loss_options(T)
    g24_pose_inputs(T)
    g1_pose(T)
    g2_g3_g4_geometry(T)
    g5_g6_ssim(T)
    g7_g8_chain(T)
    g9_smooth(T)
    g10_sql(nets)
    g11_qtr(nets)
    g12_pose(nets)
    g13_decoder(nets)
    g14_depth_errors(T)
    g15_g16_step(T, nets)
    state_dict_keys(nets)
    options_spec()


if __name__ == "__main__":
    main()
