#!/usr/bin/env python3
"""bench.py — SQLdepth self-supervised training throughput on MI355X (BASELINE.json metric:
"train images/sec, ResNet-50 640x192").

    python bench.py [--gpus N] [--steps K] [--warmup W]            (N=1 default; N>1 under torchrun)

One step = Trainer.train_step on one synthetic KITTI-shaped batch (configs[1]: ResNet-50 +
Depth_Decoder_QueryTr, 192x640, batch 12 per GPU, fp32, 2 source frames): encoder + depth head +
2 x PoseCNN forward, the fused photometric chain, full backward, gradient all-reduce (N>1), Adam.
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line with the job-wide
images/s, the roofline record of the fused warp+SSIM forward kernel (live HIP-event timing on the
launch stream) and — at N=1 — a host-CPU baseline of the oracle restatement on a bounded sample."""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "sfmnext-impl_amd"))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

CONFIG_B = ["--backbone", "resnet", "--num_layers", "50", "--num_features", "256", "--model_dim", "32",
            "--patch_size", "16", "--query_nums", "64", "--dim_out", "64", "--height", "192", "--width", "640",
            "--batch_size", "12", "--min_depth", "0.001", "--max_depth", "80.0", "--num_workers", "0",
            "--sqd_synthetic", "--sqd_device_noise", "--log_dir", "/tmp/sqd_bench",
            "--model_name", "bench"]
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FUSED_FWD_BYTES_PER_PX = 93      # disp 1 + target 12 + sources 24 + identity/noise 8 | depth 4 + sample 16 + warped 24 + sel 4


def roofline_fused_fwd(trainer, inputs, iters=200):
    """Average duration of the fused warp+SSIM forward launch, HIP events on the launch stream."""
    from sqd import ops
    o = trainer.opt
    B, H, W = o.batch_size, o.height, o.width
    dev = trainer.device
    with torch.no_grad():
        disp = torch.rand(B, 1, H // 2, W // 2, device=dev) * 20 + 1
        depth, part = ops.depth_up_fwd(disp, H, W)
        aa = 0.01 * torch.randn(B, 2, 3, device=dev)
        tr = 0.5 * torch.randn(B, 2, 3, device=dev)
        _, _, P = ops.pose_mats_fwd(aa, tr, [1, 0], inputs[("K", 0)].contiguous(), part, H * W)
        srcs = [inputs[("color", f, 0)].contiguous() for f in (-1, 1)]
        tgt = inputs[("color", 0, 0)].contiguous()
        ident = ops.identity_fwd(tgt, srcs, torch.randn(B, 2, H, W, device=dev))
        inv_K = inputs[("inv_K", 0)].contiguous()
        res = {}
        for name, training in (("infer", False), ("train", True)):
            # buffers allocated once; the loop re-enqueues the same launch through the C ABI on torch's
            # current stream, which is also the stream the events are recorded on
            call, keep = ops.photo_fwd(depth, inv_K, P, tgt, srcs, ident, training=training, prepared_only=True)
            for _ in range(20):
                ops.photo_fwd_relaunch(call)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                ops.photo_fwd_relaunch(call)
            e1.record()
            torch.cuda.synchronize()
            res[name] = e0.elapsed_time(e1) / iters * 1e-3        # seconds per launch, back-to-back launches
    px = B * H * W
    t = res["train"]
    achieved = FUSED_FWD_BYTES_PER_PX * px / t / 1e9
    # HBM-side bytes per launch from the rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected separately and corrected with
    # the dword-access calibration of tools/pmc_calib.py, as the microarchitecture guide prescribes); a committed measurement
    # of this kernel at this size — PMC counters cannot be read from inside this process
    traffic = traffic_bytes = None
    pmc = os.path.join(REPO, "profiles", "r01k_pmc_traffic.json")
    if os.path.exists(pmc) and (B, H, W) == (12, 192, 640):
        traffic_bytes = json.load(open(pmc))["kernels"]["photo_fwd_pk_kernel<1>"]["traffic_bytes"]
        traffic = round(traffic_bytes / t / 1e9, 1)
    return {"bound": "hbm", "kernel": "photo_fwd_pk_kernel<1> (fused warp+SSIM+L1+automask fwd, training mode: also writes the 37 B/px coef+argmin maps)",
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic, "traffic_bytes_per_launch": traffic_bytes,
            "traffic_source": "profiles/r01k_pmc_traffic.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, dword calibration x2.0 / x1.0)" if traffic else None,
            "us_per_launch": round(t * 1e6, 2), "us_per_launch_inference": round(res["infer"] * 1e6, 2),
            "algorithmic_bytes_per_launch": FUSED_FWD_BYTES_PER_PX * px, "pixels_per_launch": px}


def cpu_baseline(max_seconds=30.0):
    """The oracle restatement (oracle/torch_ref.py, pinned to the reference by golden vectors) timed on
    this box's host cores: same config, same batch shape, fwd + bwd + Adam."""
    sys.path.insert(0, REPO)
    from oracle import torch_ref as O
    from datasets.synthetic import synthetic_batch
    # 32 threads: PyTorch's CPU conv/BN kernels scale poorly beyond that on this many-core host (a first
    # run with all 256 threads took 106 s per B=12 step); the sample is a reduced batch so that the whole
    # leg stays within ~30 s
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    B, H, W = 4, 192, 640
    enc = O.ResnetEncoderDecoder(50, 256, 32)
    dep = O.QueryTrDecoder(32, 32, 16, 4, 64, 64, min_val=0.001, max_val=80.0, dim_feedforward=1024)
    pose = O.PoseCNN(2)
    step = O.RefTrainStep(enc, dep, pose, (0, -1, 1), H, W)
    inputs = synthetic_batch(B, H, W)
    noise = torch.randn(B, 2, H, W)
    t0 = time.time()
    step.step(inputs, noise)                                   # warm-up (allocations, oneDNN primitive caches)
    warm = time.time() - t0
    n = max(1, min(3, int(max_seconds / max(warm, 1e-3)) - 1))
    t0 = time.time()
    for _ in range(n):
        step.step(inputs, noise)
    dt = (time.time() - t0) / n
    return {"value": round(B / dt, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "%d full train steps (fwd+bwd+Adam) of the config-B model (ResNet-50, 192x640) at batch %d after 1 "
                      "warm-up, oracle/torch_ref.py on %d host threads, %.2f s/step" % (n, B, cores, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    from options import MonodepthOptions
    from trainer import Trainer
    from datasets.synthetic import synthetic_batch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d must be launched with %d ranks (torchrun); WORLD_SIZE=%d" % (args.gpus, args.gpus, world))
    opts = MonodepthOptions().parse(CONFIG_B + os.environ.get("SQD_BENCH_EXTRA", "").split())
    import contextlib
    with contextlib.redirect_stdout(sys.stderr):          # the Trainer's banner goes to stderr: stdout carries the one JSON line
        trainer = Trainer(opts)
    trainer.set_train()
    rank, dev = trainer.rank, trainer.device
    inputs = synthetic_batch(opts.batch_size, opts.height, opts.width, opts.frame_ids, start=rank * opts.batch_size, device=dev)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.train_step(dict(inputs))
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        _, losses = trainer.train_step(dict(inputs))
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = float(losses["loss"].detach().cpu())

    roof = None
    if rank == 0 and not args.no_roofline:
        roof = roofline_fused_fwd(trainer, inputs)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()
    if rank == 0:
        from sqd import nnops
        ms = elapsed / args.steps * 1e3
        out = {"metric": "train images/sec, ResNet-50 640x192", "value": round(world * opts.batch_size / (ms * 1e-3), 2),
               "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "configs[1]: ResNet-50 + Depth_Decoder_QueryTr, KITTI 192x640, batch 12 per GPU, fp32, "
                                      "2 source frames, fwd+bwd+Adam (num_features 256, model_dim 32, patch 16, Q 64, dim_out 64)",
                          "global_batch": world * opts.batch_size, "parallelism": "dp%d" % world,
                          "operator_backends": nnops.BACKEND},
               "final_loss": round(loss, 6), "roofline": roof, "cpu_baseline": cpu}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
