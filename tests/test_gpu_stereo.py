"""--use_stereo on the device: the reference's published KITTI args files all carry it (args_files/hisfog/kitti/*.txt), which
appends the other stereo camera "s" to the source frames (reference trainer.py:52-53), takes its pose from inputs["stereo_T"]
(:406-407) and switches the mean-inverse-depth scaling of the predicted translations off (:412).  One training step of the
Trainer against the oracle, and train.py driven by the tokens of the reference's resnet_192x640.txt."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _no_dropout(models):
    for m in models:
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
            if isinstance(mod, torch.nn.MultiheadAttention):
                mod.dropout = 0.0


def test_stereo_step_matches_oracle():
    sys.path.insert(0, REPO)
    from oracle import torch_ref as O
    from options import MonodepthOptions
    from trainer import Trainer
    from datasets.synthetic import synthetic_batch
    H, W, B = 64, 96, 2
    args = ["--backbone", "resnet18_lite", "--model_dim", "16", "--patch_size", "8", "--query_nums", "12", "--dim_out", "24",
            "--height", str(H), "--width", str(W), "--batch_size", str(B), "--num_workers", "0", "--sqd_synthetic", "--use_stereo",
            "--diff_lr", "--log_dir", "/tmp/sqd_stereo_test", "--max_depth", "80.0", "--sqd_no_conv_tune", "--sqd_no_graph"]
    torch.manual_seed(0)
    tr = Trainer(MonodepthOptions().parse(args))
    assert tr.opt.frame_ids == [0, -1, 1, "s"]
    tr.set_train()
    _no_dropout(tr.models.values())
    enc = O.LiteResnetEncoderDecoder(model_dim=16)
    dep = O.QueryTrDecoder(16, 16, 8, 4, 12, 24, min_val=0.001, max_val=80.0, dim_feedforward=512, dropout=0.0)
    pose = O.PoseCNN(2)
    for ref, mine in ((enc, tr.models["encoder"]), (dep, tr.models["depth"]), (pose, tr.models["pose"])):
        ref.load_state_dict({k: v.detach().cpu() for k, v in mine.state_dict().items()})
        ref.train()
    ref = O.RefTrainStep(enc, dep, pose, (0, -1, 1, "s"), H, W, use_stereo=True, diff_lr=True)
    cpu_inputs = synthetic_batch(B, H, W, frame_ids=(0, -1, 1, "s"))
    assert "stereo_T" in cpu_inputs and ("color", "s", 0) in cpu_inputs
    noise = torch.randn(B, 3, H, W)
    for step in range(2):
        ref_out, ref_losses = ref.step(dict(cpu_inputs), noise)
        inputs = {k: v.cuda() for k, v in cpu_inputs.items()}
        inputs[("noise", 0)] = noise.cuda()
        outputs, losses = tr.train_step(inputs)
        got, want = float(losses["loss"]), float(ref_losses["loss"])
        assert abs(got - want) <= 1e-4 * abs(want), (step, got, want)
    assert float((outputs[("color", "s", 0)].cpu() - ref_out[("color", "s", 0)]).abs().max()) < 2e-5
    assert set(k for k in outputs if isinstance(k, tuple) and k[0] == "sample") == {("sample", -1, 0), ("sample", 1, 0), ("sample", "s", 0)}
    # the pose network trains at lr / 10 (--diff_lr): its update after two steps matches the oracle's
    w_ref, w_got = pose.pose_conv.weight.detach(), tr.models["pose"].pose_conv.weight.detach().cpu()
    assert torch.allclose(w_got, w_ref, atol=1e-5), float((w_got - w_ref).abs().max())


@pytest.mark.parametrize("frame_ids,stereo,mode", [
    ((0, -8, 8), False, "pairs"),              # args_files/hisfog/mc/ssl_eff78M_submit.txt
    ((0, -2, -1, 1), True, "pairs"),           # four source frames with the stereo one: SQD_MAX_SOURCES, two pair passes
    ((0, -1, 1), False, "all"),                # --pose_model_input all (reference trainer.py:339-361)
    ((0, -1, 1), True, "all"),
])
def test_frame_id_and_pose_input_variants_step_matches_oracle(frame_ids, stereo, mode):
    """two training steps of the Trainer (the second one of the three through the replayed graph) against the oracle for the frame-id /
    pose-input variants round 5 added; G24 pins the oracle's handling of them to the reference"""
    sys.path.insert(0, REPO)
    from oracle import torch_ref as O
    from options import MonodepthOptions
    from trainer import Trainer
    from datasets.synthetic import synthetic_batch
    H, W, B = 64, 96, 2
    args = ["--backbone", "resnet18_lite", "--model_dim", "16", "--patch_size", "8", "--query_nums", "12", "--dim_out", "24",
            "--height", str(H), "--width", str(W), "--batch_size", str(B), "--num_workers", "0", "--sqd_synthetic",
            "--log_dir", "/tmp/sqd_variants_test", "--max_depth", "80.0", "--sqd_no_conv_tune", "--pose_model_input", mode,
            "--frame_ids"] + [str(f) for f in frame_ids] + (["--use_stereo"] if stereo else [])
    torch.manual_seed(0)
    tr = Trainer(MonodepthOptions().parse(args))
    fids = list(frame_ids) + (["s"] if stereo else [])
    assert tr.opt.frame_ids == fids
    tr.set_train()
    _no_dropout(tr.models.values())
    enc = O.LiteResnetEncoderDecoder(model_dim=16)
    dep = O.QueryTrDecoder(16, 16, 8, 4, 12, 24, min_val=0.001, max_val=80.0, dim_feedforward=512, dropout=0.0)
    pose = O.PoseCNN(2 if mode == "pairs" else len(frame_ids))
    for ref, mine in ((enc, tr.models["encoder"]), (dep, tr.models["depth"]), (pose, tr.models["pose"])):
        ref.load_state_dict({k: v.detach().cpu() for k, v in mine.state_dict().items()})
        ref.train()
    ref = O.RefTrainStep(enc, dep, pose, fids, H, W, use_stereo=stereo, pose_model_input=mode)
    S = len(fids) - 1
    g = torch.Generator().manual_seed(3)
    for step in range(5):                      # three eager steps, then the captured graph (steps 4 and 5 replay it)
        cpu_inputs = synthetic_batch(B, H, W, frame_ids=fids, start=step * B)
        noise = torch.randn(B, S, H, W, generator=g)
        ref_out, ref_losses = ref.step(dict(cpu_inputs), noise)
        inputs = {k: v.cuda() for k, v in cpu_inputs.items()}
        inputs[("noise", 0)] = noise.cuda()
        outputs, losses = tr.train_step(inputs)
        got, want = float(losses["loss"].detach()), float(ref_losses["loss"].detach())
        assert abs(got - want) <= 2e-4 * abs(want), (step, got, want)
        if step == 0:                          # (before the two trainings drift apart by their rounding)
            for f in fids[1:]:
                assert float((outputs[("color", f, 0)].cpu() - ref_out[("color", f, 0)].detach()).abs().max()) < 5e-5, f
    assert tr._graph is not None
    assert set(k[1] for k in outputs if isinstance(k, tuple) and k[0] == "sample") == set(fids[1:])
    w_ref, w_got = pose.pose_conv.weight.detach(), tr.models["pose"].pose_conv.weight.detach().cpu()
    assert torch.allclose(w_got, w_ref, atol=2e-5), float((w_got - w_ref).abs().max())


def test_train_py_runs_the_published_resnet_192x640_args(golden, tmp_path):
    """`python train.py args_files/hisfog/kitti/resnet_192x640.txt` of the reference (its tokens are frozen in the options
    fixture): --backbone resnet_lite, 50 layers, --use_stereo, --diff_lr, ... — minus the checkpoint paths that do not exist
    here, on synthetic frames, one short epoch."""
    import train
    g = golden("g00_options_spec")
    files = [str(f) for f in g["files"]]
    idx = next(i for i, f in enumerate(files) if f.endswith("hisfog/kitti/resnet_192x640.txt"))
    toks, out, skip = str(g["tokens"][idx]).split(), [], 0
    for t in toks:
        if skip:
            skip -= 1
            continue
        if t in ("--data_path", "--log_dir", "--pose_net_path", "--load_weights_folder", "--batch_size", "--num_epochs"):
            skip = 1
            continue
        if t == "--pretrained_pose":
            continue
        out.append(t)
    assert "--use_stereo" in out and "--diff_lr" in out
    out += ["--sqd_synthetic", "--sqd_synthetic_len", "6", "--batch_size", "2", "--num_epochs", "1", "--num_workers", "0",
            "--log_dir", str(tmp_path), "--log_frequency", "2"]
    train.main(out)
    assert os.path.isfile(os.path.join(str(tmp_path), "res_088", "models", "weights_0", "encoder.pth"))
