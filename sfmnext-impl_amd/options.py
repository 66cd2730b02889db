"""Command-line options of the MI355X build — parses the reference's args files unchanged.

Every flag of the reference parser (reference options.py:21-341: name, type, default, nargs,
choices, store_true quirks such as `--png` whose default is the truthy string '.png') is declared in
the FLAGS table below; tests/test_host_logic.py checks the table against the spec frozen from the
reference's own parser (tests/golden/g00_options_spec.npz) and replays every reference args file.
A few build-side flags are appended (prefixed here with "sqd")."""
import argparse
import os

_T = True   # store_true
_SPLITS = ["eigen_zhou", "eigen_full", "odom", "benchmark", "cityscapes_preprocessed", "mc_dataset",
           "mc_mini_dataset", "nyu_raw"]
_DATASETS = ["kitti", "kitti_odom", "kitti_depth", "kitti_test", "cityscapes_preprocessed", "mc_dataset",
             "mc_mini_dataset", "nyu_raw"]

# (flag, type | _T, default, extras)
FLAGS = [
    # paths
    ("intrinsics_file_path", str, "./splits/mc_dataset/KV_intrinsics.txt", {}),
    ("eval_data_path", str, "data/CS_RAW/", {}),
    ("data_path", str, "/home/Process3/KITTI_depth", {}),
    ("log_dir", str, os.path.join(os.path.expanduser("~"), "tmp"), {}),
    # training options
    ("model_name", str, "mdp", {}),
    ("split", str, "eigen_zhou", {"choices": _SPLITS}),
    ("num_features", int, 512, {}),
    ("num_layers", int, 50, {"choices": [18, 34, 50, 101, 152]}),
    ("dec_channels", int, [1024, 512, 256, 128], {"nargs": "+"}),
    ("backbone", str, "convnext_large", {}),
    ("dataset", str, "kitti", {"choices": _DATASETS}),
    ("png", _T, ".png", {}),
    ("dim_out", int, 128, {}),
    ("query_nums", int, 128, {}),
    ("patch_size", int, 20, {}),
    ("model_dim", int, 32, {}),
    ("height", int, 320, {}),
    ("width", int, 1024, {}),
    ("reg_wt", float, 0.01, {}),
    ("feat_wt", float, 0.01, {}),
    ("l1_weight", float, 0.15, {}),
    ("ssim_weight", float, 0.85, {}),
    ("use_mini_reprojection_loss", _T, False, {}),
    ("use_improved_mini_reproj_loss", _T, False, {}),
    ("use_photo_geo_loss", _T, False, {}),
    ("use_flow_pose", _T, False, {}),
    ("loss_geo_weight", float, 1.0, {}),
    ("loss_photo_weight", float, 1.0, {}),
    ("loss_rt_weight", float, 1.0, {}),
    ("loss_rc_weight", float, 1.0, {}),
    ("disparity_smoothness", float, 1e-3, {}),
    ("scales", int, [0], {"nargs": "+"}),
    ("min_depth", float, 0.001, {}),
    ("max_depth", float, 80.0, {}),
    ("use_optical_flow", _T, False, {}),
    ("use_rectify_net", _T, False, {}),
    ("use_stereo", _T, False, {}),
    ("frame_ids", int, [0, -1, 1], {"nargs": "+"}),
    ("pretrained_flow", _T, False, {}),
    ("pretrained_rectify", _T, False, {}),
    ("load_adam", _T, False, {}),
    ("load_pretrained_model", _T, False, {}),
    ("load_pt_folder", str, None, {}),
    ("pose_net_path", str, "/home/Process3/tmp/mdp/models_22_6_27/models/weights_19/", {}),
    ("pretrained_pose", _T, False, {}),
    ("log_attn", _T, False, {}),
    ("multi_gpu", _T, False, {}),
    ("diff_lr", _T, False, {}),
    ("accumulation_steps", int, 1, {}),
    # optimisation options
    ("batch_size", int, 12, {}),
    ("learning_rate", float, 1e-4, {}),
    ("num_epochs", int, 20, {}),
    ("scheduler_step_size", int, 15, {}),
    # ablation options
    ("v1_multiscale", _T, False, {}),
    ("avg_reprojection", _T, False, {}),
    ("disable_automasking", _T, False, {}),
    ("predictive_mask", _T, False, {}),
    ("no_ssim", _T, False, {}),
    ("weights_init", str, "pretrained", {"choices": ["pretrained", "scratch"]}),
    ("pose_model_input", str, "pairs", {"choices": ["pairs", "all"]}),
    ("pose_model_type", str, "posecnn", {"choices": ["posecnn", "pose_flow", "separate_resnet", "shared"]}),
    # system options
    ("no_cuda", _T, False, {}),
    ("num_workers", int, 8, {}),
    # loading / inference options
    ("pred_metric_depth", _T, False, {}),
    ("ext", str, "png", {}),
    ("image_path", str, None, {}),
    ("load_weights_folder", str, None, {}),
    ("models_to_load", str, ["encoder", "depth", "pose_encoder", "pose"], {"nargs": "+"}),
    # logging options
    ("log_frequency", int, 10, {}),
    ("save_frequency", int, 1, {}),
    # evaluation options
    ("eval_stereo", _T, False, {}),
    ("eval_mono", _T, False, {}),
    ("disable_median_scaling", _T, False, {}),
    ("pred_depth_scale_factor", float, 1, {}),
    ("ext_disp_to_eval", str, None, {}),
    ("eval_split", str, "eigen", {"choices": ["eigen", "eigen_benchmark", "benchmark", "odom_9", "odom_10", "cityscapes"]}),
    ("save_pred_disps", _T, False, {}),
    ("no_eval", _T, False, {}),
    ("eval_eigen_to_benchmark", _T, False, {}),
    ("eval_out_dir", str, None, {}),
    ("post_process", _T, False, {}),
]

# build-side additions (not in the reference)
EXTRA_FLAGS = [
    ("sqd_synthetic", _T, False, {"help": "feed synthetic KITTI-shaped batches even if --data_path exists"}),
    ("sqd_synthetic_len", int, 240, {"help": "samples per epoch of the synthetic dataset"}),
    ("sqd_bucket_mb", float, 32.0, {"help": "gradient all-reduce bucket size (MB) for multi-GPU runs"}),
    ("sqd_device_noise", _T, False, {"help": "draw the tie-break noise on the device instead of the CPU RNG"}),
    ("sqd_channels_last", _T, False, {"help": "keep network activations NHWC in memory (always on with the native convolutions)"}),
    ("sqd_no_graph", _T, False, {"help": "never capture the training step into a hipGraph (single-device runs replay one after 3 eager steps)"}),
    ("sqd_graph_ddp", str, "overlap", {"choices": ["overlap", "post"],
                                        "help": "multi-rank hipGraph: 'overlap' (default) = the bucketed all-reduces are graph branches next to backward; "
                                                "'post' = graph of forward+backward, collectives and Adam issued after each replay"}),
    ("sqd_bf16", _T, False, {"help": "bf16 training arithmetic for the convolutions' forward and data gradient (bf16 MFMA, fp32 accumulation; "
                                      "BatchNorm, losses, weight gradients and Adam stay fp32) — BASELINE.json configs[3]"}),
    ("sqd_no_f16x2", _T, False, {"help": "keep the convolutions off the two-term fp16 operand plans (power-of-two scaled operands, three products on the fp16 "
                                          "matrix cores, fp32 accumulation: fp32-level accuracy) — the plan space of round 4: three-term bf16 and fp32 MFMA"}),
    ("sqd_graph_wgrad_batch", int, 0, {"help": "inside the captured step, run the weight gradients of once-used filters on a side branch, one cross-branch "
                                                "dependency per this many convolutions (0: on the capturing stream, with their split sums riding on the next "
                                                "BatchNorm-backward launch)"}),
    ("sqd_no_conv_tune", _T, False, {"help": "keep the cost-model convolution plans instead of timing tile / split-K plans per layer in the first step"}),
    ("sqd_conv_plans", str, None, {"help": "JSON file of convolution plans (written by --sqd_save_conv_plans): pins every listed layer's kernel, "
                                        "tile and split instead of timing them in the first step — last-bit reproducible across boxes"}),
    ("sqd_save_conv_plans", str, None, {"help": "write the convolution plans this run measured (or loaded) to this JSON file after the first steps"}),
    ("sqd_early_identity", _T, False, {"help": "captured step: evaluate the identity-reprojection maps at the start of the step (as the eager steps do "
                                                "on their side stream) instead of right before the fused warp + SSIM kernel"}),
    ("sqd_no_defer_wgrad_reduce", _T, False, {"help": "sum every weight gradient's pixel splits in a launch of its own instead of letting the sum ride "
                                                       "on the next BatchNorm-backward launch (single-rank default: it rides)"}),
]


class MonodepthOptions:
    def __init__(self):
        self.parser = argparse.ArgumentParser(description="SQLdepth options (MI355X build)", fromfile_prefix_chars="@")
        for name, typ, default, extra in FLAGS + EXTRA_FLAGS:
            if typ is _T:
                self.parser.add_argument("--" + name, action="store_true", default=default, help=extra.get("help"))
            else:
                self.parser.add_argument("--" + name, type=typ, default=default, **extra)

    def parse(self, argv=None):
        self.options = self.parser.parse_args(argv)
        return self.options
