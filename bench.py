#!/usr/bin/env python3
"""bench.py — SQLdepth self-supervised training throughput on MI355X (BASELINE.json metric:
"train images/sec, ResNet-50 640x192").

    python bench.py [--gpus N] [--steps K] [--warmup W]     (N=1 default; N>1: under torchrun, or plain — it then starts the ranks itself)

One step = Trainer.train_step on one synthetic KITTI-shaped batch (configs[1]: ResNet-50 +
Depth_Decoder_QueryTr, 192x640, batch 12 per GPU, fp32, 2 source frames): encoder + depth head +
2 x PoseCNN forward, the fused photometric chain, full backward, gradient all-reduce (N>1), Adam.
Inputs are resident in HBM before the timed region: NBATCH different batches, fed in rotation, each copied
into the replayed graph's input tensors inside the timed region (a run never sees the same batch twice in a
row).  The convolution plans are the pinned set shipped for configs[1] (plans/): every box runs the same
kernels; the line says whether this box's own plan timing would have chosen differently.  Rank 0 prints ONE
JSON line with the job-wide images/s, the roofline record of the fused warp+SSIM forward kernel (live
HIP-event timing on the launch stream, inside training steps) and — at N=1 — a host-CPU baseline of the
oracle restatement on a bounded sample."""
import argparse
import contextlib
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(REPO, "sfmnext-impl_amd"))

import torch  # noqa: E402

CONFIG_B = ["--backbone", "resnet", "--num_layers", "50", "--num_features", "256", "--model_dim", "32",
            "--patch_size", "16", "--query_nums", "64", "--dim_out", "64", "--height", "192", "--width", "640",
            "--batch_size", "12", "--min_depth", "0.001", "--max_depth", "80.0", "--num_workers", "0",
            "--sqd_synthetic", "--sqd_device_noise", "--log_dir", "/tmp/sqd_bench",
            "--model_name", "bench"]
NBATCH = 3                       # resident batches fed in rotation
PINNED_PLANS = os.path.join(REPO, "sfmnext-impl_amd", "plans", "configB_resnet50_192x640_b12.json")
HBM_PEAK_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FUSED_FWD_BYTES_PER_PX = 93      # disp 1 + target 12 + sources 24 + identity/noise 8 | depth 4 + sample 16 + warped 24 + sel 4


def kernel_source_hash():
    """sha256 over the photometric kernel sources of the loaded library: a PMC traffic record is only valid for the
    kernel revision it was measured on."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(REPO, "sfmnext-impl_amd", "csrc", "photo*.hip")) +
                    glob.glob(os.path.join(REPO, "sfmnext-impl_amd", "csrc", "photo*.h")) +
                    [os.path.join(REPO, "sfmnext-impl_amd", "csrc", "sqd_common.h")]):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


PMC_TRAFFIC_JSON = os.path.join(REPO, "profiles", "r06_pmc_traffic.json")


def roofline_fused_fwd(trainer, batches, iters=200, graph_us=None):
    """Average duration of the fused warp+SSIM forward launch, HIP events on the launch stream."""
    from sqd import ops
    o = trainer.opt
    B, H, W = o.batch_size, o.height, o.width
    dev = trainer.device
    inputs = batches[0]
    with torch.no_grad():
        disp = torch.rand(B, 1, H // 2, W // 2, device=dev) * 20 + 1
        depth, part = ops.depth_up_fwd(disp, H, W)
        aa = 0.01 * torch.randn(B, 2, 3, device=dev)
        tr = 0.5 * torch.randn(B, 2, 3, device=dev)
        _, _, P = ops.pose_mats_fwd(aa, tr, [1, 0], inputs[("K", 0)].contiguous(), part, H * W)
        srcs = [inputs[("color", f, 0)].contiguous() for f in (-1, 1)]
        hwc = bool(getattr(trainer, "_hwc_keys", None))
        if hwc:      # the replayed step keeps its source frames in [B,H,W,3] memory (trainer._capture): time the kernel on what it reads there
            srcs = ops.pack_pixels(srcs)
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
    # the same launch INSIDE training steps (eager, i.e. not the replayed graph, so that HIP events can bracket it): the frames,
    # depth and identity maps were last touched milliseconds earlier and the caches hold the step's other tensors — this is the
    # figure a kernel trace of the step shows (57-61 us), not the back-to-back relaunch of one 137 MB working set above
    in_step = None
    if trainer.reducer is None:
        from options import MonodepthOptions
        from trainer import Trainer
        with contextlib.redirect_stdout(sys.stderr):
            eager = Trainer(MonodepthOptions().parse(bench_args() + ["--sqd_no_graph"]))
        eager.set_train()
        ops.PHOTO_FWD_EVENTS = []
        try:
            ebatches = [dict(b) for b in batches]
            if hwc:
                for b in ebatches:
                    for f in (-1, 1):
                        b[("color", f, 0)] = b[("color", f, 0)].contiguous(memory_format=torch.channels_last)
            for i in range(12):
                eager.train_step(dict(ebatches[i % len(ebatches)]))
            torch.cuda.synchronize()
            ts = sorted(e0.elapsed_time(e1) * 1e-3 for e0, e1 in ops.PHOTO_FWD_EVENTS[2:])
            in_step = ts[len(ts) // 2]
        finally:
            ops.PHOTO_FWD_EVENTS = None
    px = B * H * W
    hot = res["train"]
    # THE figure of the record (frac / achieved / us_per_launch) is the launch inside training steps; the back-to-back relaunch of one
    # Infinity-Cache-resident working set is reported beside it as cache_resident.  (Multi-rank runs have no eager in-step probe:
    # there the cache-resident figure stands in and `timing` says so.)
    # (best: the kernel's duration inside the REPLAYED graph steps — what a kernel trace of the timed loop shows; eager steps leave the
    #  device idle half of the time and its clocks lower: the eager probe reads 10-30 % slow and is only the fallback)
    eager_us = None if in_step is None else round(in_step * 1e6, 2)
    if graph_us:
        in_step = graph_us * 1e-6
    t = in_step if in_step is not None else hot
    achieved = FUSED_FWD_BYTES_PER_PX * px / t / 1e9
    # HBM-side bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (tools/pmc_traffic.py, corrected
    # with the access-width calibration of tools/pmc_calib.py as the microarchitecture guide prescribes).  PMC counters cannot be
    # read from inside this process: the committed record is used only if it was measured on THIS revision of the kernel sources.
    traffic = traffic_bytes = note = None
    if os.path.exists(PMC_TRAFFIC_JSON) and (B, H, W) == (12, 192, 640):
        rec = json.load(open(PMC_TRAFFIC_JSON))
        if rec.get("kernel_source_hash") == kernel_source_hash():
            traffic_bytes = rec["kernels"][rec["judged_kernel"]]["traffic_bytes"]
            traffic = round(traffic_bytes / t / 1e9, 1)
            note = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, access-width calibration)" % os.path.basename(PMC_TRAFFIC_JSON)
        else:
            note = "stale PMC record (kernel sources changed since %s was measured): traffic withheld" % os.path.basename(PMC_TRAFFIC_JSON)
    n_src = len([f for f in trainer.opt.frame_ids if f != 0]) + (1 if trainer.opt.use_stereo and "s" not in trainer.opt.frame_ids else 0)
    caveat = None if n_src == 2 else ("%d source frames: the sources are processed in pairs, %d launches per step (the last one with a single source and the running "
                                      "minimum carried in); us_per_launch averages them and is priced with the two-source launch's 93 B/px — not comparable "
                                      "with configs[1]'s figure" % (n_src, (n_src + 1) // 2))
    return {"bound": "hbm", "kernel": ops.PHOTO_FWD_KERNEL_NAME, **({"caveat": caveat} if caveat else {}),
            "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic, "traffic_bytes_per_launch": traffic_bytes, "traffic_source": note,
            "us_per_launch": round(t * 1e6, 2),
            "us_per_launch_eager_steps": eager_us,
            "clock": ("torch.profiler device activity (roctracer) — the ONE clock of frac / achieved / us_per_launch in this line; the record kept under "
                      "profiles/ is rocprofv3 --kernel-trace of the same command (profiles/r06*_bench_kernel_trace_stats.md), which reads the same launch "
                      "a few per cent apart (round 6: 44.1 against 45.6 us on planar source frames, 42.9 against 42.4 on channels_last ones): compare a line with a line and a trace with a trace" if graph_us else
                      "HIP events on the launch stream (eager steps)"),
            "timing": ("average duration of the launch inside replayed hipGraph training steps (device activity records of 4 steps, child process); "
                       "us_per_launch_eager_steps = median of 10 launches bracketed by HIP events on the launch stream inside EAGER steps" if graph_us else
                       "median of 10 launches, HIP events on the launch stream around the launch inside eager training steps (rotating batches)"
                       if in_step is not None else
                       "no in-step probe in a multi-rank run: HIP events around %d back-to-back launches of one working set" % iters),
            "cache_resident": {
                "us_per_launch": round(hot * 1e6, 2), "us_per_launch_inference": round(res["infer"] * 1e6, 2),
                "achieved": round(FUSED_FWD_BYTES_PER_PX * px / hot / 1e9, 1),
                "frac_cache_resident": round(FUSED_FWD_BYTES_PER_PX * px / hot / 1e9 / HBM_PEAK_GBS, 4),
                "timing": "HIP events on the launch stream around %d back-to-back launches of one 137 MB working set (Infinity-Cache resident: "
                          "an upper bound, not the figure of a training step)" % iters},
            "algorithmic_bytes_per_launch": FUSED_FWD_BYTES_PER_PX * px, "pixels_per_launch": px}


def _time_calls(fn, iters=100, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def roofline_depthwise(trainer):
    """Config D (EfficientNet-b5): the dominant kernel family is the depthwise k x k convolution (HBM-bound: one read and one
    write of the activation, 8 B per element) — timed on the layer of the trunk that moves the most bytes, through the C ABI
    on the launch stream."""
    from sqd import lib as _l, nnkernels
    o = trainer.opt
    dev = trainer.device
    best = None
    H, W = o.height // 2, o.width // 2                       # after the stride-2 stem
    trunk = trainer.models["encoder"].encoder              # BaseEncoder: .original_model is the trunk; Unet: the features_only trunk itself
    for stage in getattr(trunk, "original_model", trunk).blocks:
        for blk in stage:
            conv = blk.conv_dw
            k, st, C = conv.kernel_size[0], conv.stride[0], conv.in_channels
            (Ho, pt), (Wo, pl) = nnkernels.tf_same_pad(H, k, st), nnkernels.tf_same_pad(W, k, st)
            elems = o.batch_size * C * (H * W + Ho * Wo)
            if best is None or elems > best[0]:
                best = (elems, C, k, st, H, W, Ho, Wo, pt, pl)
            H, W = Ho, Wo
    elems, C, k, st, H, W, Ho, Wo, pt, pl = best
    N = o.batch_size
    x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(k * k, C, device=dev)
    y = torch.empty((N, C, Ho, Wo), device=dev).contiguous(memory_format=torch.channels_last)
    L, st_ = _l.lib(), torch.cuda.current_stream().cuda_stream
    t = _time_calls(lambda: _l.check(L.sqd_dw_conv_fwd(x.data_ptr(), wt.data_ptr(), y.data_ptr(), N, H, W, C, k, st, pt, pl, Ho, Wo, st_), "dw_conv_fwd"))
    nbytes = 4 * elems
    return {"bound": "hbm", "kernel": "dw_conv_run_kernel (depthwise %dx%d / stride %d forward, [%d,%d,%d,%d] -> %dx%d: the trunk's largest)" % (k, k, st, N, C, H, W, Ho, Wo),
            "achieved": round(nbytes / t / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(nbytes / t / 1e9 / HBM_PEAK_GBS, 4),
            "traffic": None, "traffic_source": "no PMC record for this kernel", "us_per_launch": round(t * 1e6, 2),
            "algorithmic_bytes_per_launch": nbytes, "note": "8 B per element: one read of the input, one write of the output; filters are %d floats" % (k * k * C)}


def roofline_mlp_gemm(trainer):
    """Config E (ConvNeXt-L): the dominant kernels are the 1x1 convolutions of the block MLPs (Linear C -> 4C -> C) — timed on
    stage 2's expansion (27 of the 36 blocks) under the plan the tuner registered for it, priced against the fp32 MFMA peak
    (the arithmetic is fp32-equivalent whichever plan runs)."""
    from sqd import lib as _l, nnkernels
    o = trainer.opt
    dev = trainer.device
    N, H, W, C, K = o.batch_size, o.height // 16, o.width // 16, 768, 3072
    geom = (N, H, W, C, K, 1, 1, 1, 0, H, W)
    x = torch.randn(N, C, H, W, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn(K, C, 1, 1, device=dev).contiguous(memory_format=torch.channels_last)
    b = torch.zeros(K, device=dev)
    y = torch.empty((N, K, H, W), device=dev).contiguous(memory_format=torch.channels_last)
    ws = nnkernels._conv_ws(0, geom, dev)
    L, st_ = _l.lib(), torch.cuda.current_stream().cuda_stream
    plan = nnkernels.CHOSEN_PLANS.get(("fwd",) + geom)
    split3, h2 = bool(plan and plan[3] & 1024), bool(plan and plan[3] & 4096)
    ax = aw = None
    if h2:                                     # a two-term fp16 plan takes the operands' max |.| (device scalars)
        ax, aw = (torch.zeros(nnkernels.AMAX_REC, device=dev) for _ in range(2))
        for t_, a_ in ((x, ax), (w, aw)):
            _l.check(L.sqd_amax(t_.data_ptr(), t_.numel(), a_.data_ptr(), st_), "amax")
    t = _time_calls(lambda: _l.check(L.sqd_conv_fwd_scaled(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), None if ws is None else ws.data_ptr(), None,
                                                           None if ax is None else ax.data_ptr(), None if aw is None else aw.data_ptr(), None,
                                                           N, H, W, C, K, 1, 1, 1, 0, H, W, 0, st_), "conv_fwd"))
    flops = 2.0 * N * H * W * C * K
    # the bound of the arithmetic the registered plan runs: fp32 MFMA 157.3 TFLOP/s; a three-term bf16 plan issues 6 bf16 products per
    # fp32 product on the 2500 TFLOP/s dense bf16 pipe (MI355X_MICROARCH.md) = 416.7 TFLOP/s of fp32-equivalent work; a two-term fp16
    # plan 3 products on the fp16 pipe (same dense peak) = 833.3
    peak = 2500.0 / 3.0 if h2 else 2500.0 / 6.0 if split3 else 157.3
    arith = "two-term fp16 operands" if h2 else "three-term bf16 operands" if split3 else "fp32 MFMA"
    return {"bound": "mfma", "kernel": "conv_gemm_kernel (1x1 convolution %d -> %d on [%d,%d,%d]: stage-2 MLP expansion; plan (bm, bn, split, bk flags) %s = %s)"
                                       % (C, K, N, H, W, plan, arith),
            "achieved": round(flops / t / 1e12, 1), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(flops / t / 1e12 / peak, 4),
            "frac_of_fp32_mfma_peak": round(flops / t / 1e12 / 157.3, 4),
            "traffic": None, "us_per_launch": round(t * 1e6, 2), "algorithmic_flops_per_launch": flops,
            "note": "achieved = fp32-equivalent FLOP (2 M C K) per second; peak = the matrix-pipe bound of the plan's arithmetic"}


def host_cpu():
    """(model name, logical cpus, physical cores) of this box."""
    import subprocess
    model, sockets, cps = "unknown", 1, None
    try:
        for line in subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout.splitlines():
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "Model name":
                model = v
            elif k == "Socket(s)":
                sockets = int(v)
            elif k == "Core(s) per socket":
                cps = int(v)
    except Exception:
        pass
    logical = os.cpu_count() or 1
    return model, logical, (sockets * cps if cps else logical)


def cpu_baseline(budget_s=45.0):
    """The oracle restatement (oracle/torch_ref.py, pinned to the reference by golden vectors) timed on this box's host cores
    as BASELINE.md §2 prescribes: (1) config A end-to-end, (2) the photometric chain alone at configs B and C, and — the
    `value` of the record, because it is the workload of the GPU line — (3) the config-B training step on a bounded batch."""
    sys.path.insert(0, REPO)
    from oracle import torch_ref as O
    from datasets.synthetic import synthetic_batch
    model, logical, physical = host_cpu()
    # all physical cores, capped at 64: beyond that PyTorch's CPU conv / pooling kernels lose throughput on this many-core host
    # (round 1 measured a B=12 step at 106 s with 256 threads against ~2.4 s with 32)
    cores = max(1, min(physical, 64))
    torch.set_num_threads(cores)
    t_start = time.time()

    def timed(fn, warm, n_max, budget):
        t0 = time.time()
        for _ in range(warm):
            fn()
        w = (time.time() - t0) / max(warm, 1)
        n = max(1, min(n_max, int(budget / max(w, 1e-3))))
        t0 = time.time()
        for _ in range(n):
            fn()
        return (time.time() - t0) / n, n

    # (3) config B, the GPU line's workload, batch 4 (a B=12 step is ~3x that; img/s is per image)
    B, H, W = 4, 192, 640
    enc = O.ResnetEncoderDecoder(50, 256, 32)
    dep = O.QueryTrDecoder(32, 32, 16, 4, 64, 64, min_val=0.001, max_val=80.0, dim_feedforward=1024)
    stepB = O.RefTrainStep(enc, dep, O.PoseCNN(2), (0, -1, 1), H, W)
    inputs, noise = synthetic_batch(B, H, W), torch.randn(B, 2, H, W)
    dtB, nB = timed(lambda: stepB.step(inputs, noise), 1, 3, 8.0)
    del stepB, enc, dep
    # (1) config A end-to-end: ResNet-18 + Lite decoder, 192x640, batch 2, 3 warm-up + up to 20 timed steps
    encA = O.LiteResnetEncoderDecoder(model_dim=32)
    depA = O.QueryTrDecoder(32, 32, 16, 4, 120, 128, min_val=0.001, max_val=80.0, dim_feedforward=512)
    stepA = O.RefTrainStep(encA, depA, O.PoseCNN(2), (0, -1, 1), 192, 640)
    inA, nzA = synthetic_batch(2, 192, 640), torch.randn(2, 2, 192, 640)
    dtA, nA = timed(lambda: stepA.step(inA, nzA), 3, 20, 12.0)
    del stepA, encA, depA

    # (2) photometric chain alone (generate_images_pred + compute_losses + backward) at configs B and C
    def chain(Bc, Hc, Wc):
        batch = synthetic_batch(Bc, Hc, Wc)
        disp = (torch.rand(Bc, 1, Hc // 2, Wc // 2) * 20 + 1).requires_grad_(True)
        poses = {f: ((0.01 * torch.randn(Bc, 1, 1, 3)).requires_grad_(True), (0.5 * torch.randn(Bc, 1, 1, 3)).requires_grad_(True))
                 for f in (-1, 1)}
        colors = {f: batch[("color", f, 0)] for f in (0, -1, 1)}
        nz = torch.randn(Bc, 2, Hc, Wc)

        def run():
            out = O.photometric_chain(disp, poses, batch[("K", 0)], batch[("inv_K", 0)], colors, [0, -1, 1], nz, Hc, Wc)
            out["loss"].backward()
        dt, n = timed(run, 1, 3, 5.0)
        px = Bc * Hc * Wc
        return {"images_per_s": round(Bc / dt, 2), "s_per_pass": round(dt, 3), "passes": n,
                "effective_GBps": round((93 + 84 + 36) * px / dt / 1e9, 3)}
    chainB, chainC = chain(12, 192, 640), chain(8, 320, 1024)
    return {"value": round(B / dtB, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "%d full train steps (fwd+bwd+Adam) of the config-B model (ResNet-50 + Depth_Decoder_QueryTr, 192x640) at batch %d "
                      "after 1 warm-up, oracle/torch_ref.py, %.2f s/step, on %d threads (%s)" % (
                          nB, B, dtB, cores, "capped from %d physical cores: PyTorch's CPU convolutions lose throughput beyond 64 threads on this host"
                          % physical if cores < physical else "all physical cores"),
            "host": {"cpu_model": model, "os_cpu_count": logical, "physical_cores": physical, "threads_used": cores,
                     "torch": torch.__version__},
            "config_A_end_to_end": {"images_per_s": round(2 / dtA, 3), "s_per_step": round(dtA, 3), "timed_steps": nA, "warmup": 3,
                                    "workload": "ResNet-18 + Lite_Depth_Decoder_QueryTr, 192x640, batch 2, fwd+bwd+Adam"},
            "photometric_chain_config_B": chainB, "photometric_chain_config_C": chainC,
            "total_cpu_leg_s": round(time.time() - t_start, 1)}


def workload_name(o):
    """configs[1] of BASELINE.json by default; SQD_BENCH_EXTRA runs say what they changed"""
    base = ("configs[1]: ResNet-50 + Depth_Decoder_QueryTr, KITTI 192x640, batch 12 per GPU, fp32, "
            "2 source frames, fwd+bwd+Adam (num_features 256, model_dim 32, patch 16, Q 64, dim_out 64)")
    if not os.environ.get("SQD_BENCH_EXTRA"):
        return base
    return (os.environ.get("SQD_BENCH_WORKLOAD", "") + " " if os.environ.get("SQD_BENCH_WORKLOAD") else "") + \
           ("NOT configs[1] (SQD_BENCH_EXTRA=%r): backbone %s, %dx%d, batch %d per GPU, %s convolution operands, frames %s, "
            "num_features %d, model_dim %d, patch %d, Q %d, dim_out %d" %
            (os.environ["SQD_BENCH_EXTRA"], o.backbone, o.height, o.width, o.batch_size, "bf16" if o.sqd_bf16 else "fp32",
             list(o.frame_ids) + (["s"] if o.use_stereo else []), o.num_features, o.model_dim, o.patch_size, o.query_nums, o.dim_out))


def conv_source_hash():
    """sha256 over the convolution kernel sources: a pinned plan set names kernels of THIS revision"""
    import hashlib
    h = hashlib.sha256()
    for f in ("conv.hip", "sqd_common.h"):
        h.update(open(os.path.join(REPO, "sfmnext-impl_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def pinned_plans():
    """(path or None, note): the plan set shipped for configs[1], if it was measured on this revision of the convolution kernels"""
    if os.environ.get("SQD_BENCH_EXTRA") or os.environ.get("SQD_BENCH_LIVE_PLANS"):
        return None, "not configs[1] / live plan timing requested: plans measured in step 1"
    if not os.path.exists(PINNED_PLANS):
        return None, "no pinned plan file: plans measured in step 1"
    rec = json.load(open(PINNED_PLANS))
    if rec.get("conv_source_hash") != conv_source_hash():
        return None, "pinned plan file is stale (convolution kernel sources changed since it was measured): plans measured in step 1"
    return PINNED_PLANS, "pinned: %s (measured %s)" % (os.path.relpath(PINNED_PLANS, REPO), rec.get("measured_on", "?"))


def bench_args():
    path, _ = pinned_plans()
    return CONFIG_B + os.environ.get("SQD_BENCH_EXTRA", "").split() + (["--sqd_conv_plans", path] if path else [])


def gpu_state(index=0):
    """shader / memory clock, power draw and cap of the device as rocm-smi reports them (None where it does not)"""
    import subprocess
    out = {}
    try:
        r = subprocess.run(["rocm-smi", "-d", str(index), "-c", "-P", "-M", "--json"], capture_output=True, text=True, timeout=20)
        rec = json.loads(r.stdout)
        card = rec.get("card%d" % index) or next(iter(rec.values()))
        def mhz(v):
            digits = "".join(ch for ch in str(v) if ch.isdigit())
            return int(digits) if digits else str(v)
        for k, v in card.items():
            kl = k.lower()
            if "sclk clock speed" in kl:
                out["gpu_sclk_mhz"] = mhz(v)
            elif "mclk clock speed" in kl:
                out["gpu_mclk_mhz"] = mhz(v)
            elif "max" in kl and "power" in kl:
                out["power_cap_w"] = float(v) if str(v).replace(".", "", 1).isdigit() else str(v)
            elif "power" in kl and "(w)" in kl:
                out["power_w"] = float(v) if str(v).replace(".", "", 1).isdigit() else str(v)
    except Exception as e:                              # noqa: BLE001 — reporting only
        out["error"] = "%s: %s" % (type(e).__name__, e)
    return out


def kernel_time_per_step(timeout_s=240):
    """Sum of the kernel durations of one replayed step (torch.profiler's device activity over 4 steps of the same configuration), in a
    child process: a profiler problem must not cost the bench line."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--kernel-sum-child"], capture_output=True, text=True, timeout=timeout_s,
                           env=dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0"))
        for line in r.stdout.splitlines():
            if line.startswith("{") and "kernel_ms_per_step" in line:
                return json.loads(line)
        return {"error": "child printed no record (rc %d): %s" % (r.returncode, (r.stderr or "").strip().splitlines()[-1:] or "")}
    except Exception as e:                              # noqa: BLE001
        return {"error": "%s: %s" % (type(e).__name__, e)}


def kernel_sum_child():
    from options import MonodepthOptions
    from trainer import Trainer
    from datasets.synthetic import synthetic_batch
    from torch.profiler import profile, ProfilerActivity
    opts = MonodepthOptions().parse(bench_args())
    with contextlib.redirect_stdout(sys.stderr):
        tr = Trainer(opts)
    tr.set_train()
    batches = [synthetic_batch(opts.batch_size, opts.height, opts.width, opts.frame_ids, start=i * opts.batch_size, device=tr.device) for i in range(NBATCH)]
    for i in range(8):
        tr.train_step(dict(batches[i % NBATCH]))
    torch.cuda.synchronize()
    n = 4
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        for i in range(n):
            tr.train_step(dict(batches[i % NBATCH]))
        torch.cuda.synchronize()
    evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CUDA]

    def dur(e):
        return e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total
    total_us = sum(dur(e) for e in evs)
    judged = [dur(e) for e in evs if "photo_tile_kernel<1" in e.name]
    print(json.dumps({"kernel_ms_per_step": round(total_us / n / 1e3, 3), "device_events_per_step": round(len(evs) / n, 1),
                      "fused_fwd_us_in_graph": round(sum(judged) / len(judged), 2) if judged else None, "fused_fwd_launches": len(judged),
                      "source": "torch.profiler device activity over %d replayed steps (kernels + copies), separate process" % n}))


def time_steps(trainer, batches, steps, warmup, sync):
    for i in range(warmup):
        trainer.train_step(dict(batches[i % len(batches)]))
    sync()
    t0 = time.perf_counter()
    for i in range(steps):
        _, losses = trainer.train_step(dict(batches[i % len(batches)]))
    sync()
    return time.perf_counter() - t0, losses


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-diagnostics", action="store_true", help="skip the kernel-time sum and the live plan-timing comparison")
    ap.add_argument("--kernel-sum-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.kernel_sum_child:
        return kernel_sum_child()

    from options import MonodepthOptions
    from trainer import Trainer
    from datasets.synthetic import synthetic_batch

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as a plain `python bench.py --gpus N`: launch the N ranks ourselves (one process per GPU over RCCL); rank 0
        # prints the JSON line to the inherited stdout
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        os.execv(sys.executable, cmd)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was launched with WORLD_SIZE=%d" % (args.gpus, world))
    plan_path, plan_note = pinned_plans()
    opts = MonodepthOptions().parse(bench_args())
    with contextlib.redirect_stdout(sys.stderr):          # the Trainer's banner goes to stderr: stdout carries the one JSON line
        trainer = Trainer(opts)
    trainer.set_train()
    rank, dev = trainer.rank, trainer.device
    # NBATCH different resident batches per rank, fed in rotation: every step copies its batch into the graph's input tensors
    # inside the timed region, as a loader-fed run does
    batches = [synthetic_batch(opts.batch_size, opts.height, opts.width, opts.frame_ids,
                               start=(i * world + rank) * opts.batch_size, device=dev) for i in range(NBATCH)]

    from sqd import ddp, nnkernels

    def sync():
        torch.cuda.synchronize()
        if ddp.COMM is not None:
            ddp.COMM.barrier()          # an all-reduce over RCCL + a stream synchronize: every rank's device work is done
        torch.cuda.synchronize()

    state0 = gpu_state(trainer.local_rank) if rank == 0 else None
    elapsed, losses = time_steps(trainer, batches, args.steps, args.warmup, sync)
    state1 = None
    if rank == 0 and world == 1 and not args.no_diagnostics:
        # shader clock and power UNDER LOAD: rocm-smi polled from a thread while the same steps keep running (outside the timed region)
        import threading
        samples, stop = [], threading.Event()

        def poll():
            while not stop.is_set():
                samples.append(gpu_state(trainer.local_rank))
        th = threading.Thread(target=poll, daemon=True)
        th.start()
        t_end = time.perf_counter() + 2.5
        i = 0
        while time.perf_counter() < t_end:
            trainer.train_step(dict(batches[i % len(batches)]))
            i += 1
            if i % 20 == 0:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        stop.set()
        th.join(timeout=10)
        clk = sorted(x["gpu_sclk_mhz"] for x in samples if isinstance(x.get("gpu_sclk_mhz"), int))
        pw = sorted(x["power_w"] for x in samples if isinstance(x.get("power_w"), float))
        state1 = {"samples": len(samples), "gpu_sclk_mhz_median": clk[len(clk) // 2] if clk else None, "gpu_sclk_mhz_min": clk[0] if clk else None,
                  "gpu_sclk_mhz_max": clk[-1] if clk else None, "power_w_median": pw[len(pw) // 2] if pw else None,
                  "power_cap_w": next((x.get("power_cap_w") for x in samples if "power_cap_w" in x), None)}
    if ddp.COMM is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        ddp.COMM.all_reduce(t, "max")
        elapsed = float(t.item())
    loss = float(losses["loss"].detach().cpu())
    mix, plans_used = nnkernels.plan_mix(), nnkernels.export_plans()
    # operand scales of the two-term fp16 plans: how many max |.| scalars came out of producer kernels ("fused") and how many needed a
    # pass of their own, by call site — Python-side counts of every step that ran eagerly or was captured (replays add nothing)
    scales = {"recorded_by_producers": nnkernels.AMAX_STATS["fused"], "standalone_passes": nnkernels.AMAX_STATS["standalone"],
              "standalone_by_site": dict(nnkernels.AMAX_STATS["sites"])}
    exchange = None
    if trainer.reducer is not None:
        comm = trainer.reducer.comm
        exchange = {"communicator": type(comm).__name__, "ranks": comm.world,
                    "ranks_joined": comm.joined() if hasattr(comm, "joined") else comm.world,
                    "rccl_version": comm.rccl_version() if hasattr(comm, "rccl_version") else None,
                    "buckets": len(trainer.reducer.buckets or ()), "bucket_mb": opts.sqd_bucket_mb,
                    "bucket_bytes": trainer.reducer.bucket_bytes_list(),
                    "defer_wgrad_reduce": bool(trainer._defer_wgrad_reduce),
                    "graph_mode": trainer.graph_mode(), "capture_failures": getattr(trainer, "capture_failures", []),
                    "mode": {"eager": "eager hooks", "overlap": "all-reduces captured in the step graph",
                             "post": "graph of forward+backward, then all-reduce + Adam"}.get(trainer.graph_mode(), trainer.graph_mode())}

    ktime = None
    if rank == 0 and world == 1 and not args.no_diagnostics:
        ktime = kernel_time_per_step()
    roof = None
    if rank == 0 and not args.no_roofline:
        # the dominant kernel of the configuration that ran: the fused warp+SSIM forward (the kernel BASELINE.json's target names)
        # for the ResNet configurations, the depthwise convolution for EfficientNet-b5, the block-MLP GEMM for ConvNeXt-L
        if opts.backbone in ("eff_b5", "tf_efficientnet_b5_ap"):
            roof = roofline_depthwise(trainer)
        elif opts.backbone.startswith("convnext"):
            roof = roofline_mlp_gemm(trainer)
        else:
            roof = roofline_fused_fwd(trainer, batches, graph_us=(ktime or {}).get("fused_fwd_us_in_graph"))
    diag = None
    if rank == 0 and world == 1 and not args.no_diagnostics:
        diag = {"gpu_idle_before": state0, "gpu_under_load": state1, "kernel_time": ktime}
        if plan_path is not None:
            # what this box's own plan timing would have chosen, and what that is worth here: a second Trainer without the pinned set
            del trainer
            nnkernels.reset_plans()
            with contextlib.redirect_stdout(sys.stderr):
                live = Trainer(MonodepthOptions().parse(CONFIG_B))
            live.set_train()
            el, _ = time_steps(live, batches, 20, 6, sync)
            mine = {(e["pass"], tuple(e["geom"])): tuple(e["plan"]) for e in nnkernels.export_plans()["plans"]}
            pinned = {(e["pass"], tuple(e["geom"])): tuple(e["plan"]) for e in plans_used["plans"]}
            differ = sorted(k for k in pinned if k in mine and mine[k] != pinned[k])
            diag["live_plan_timing"] = {"ms_per_step": round(el / 20 * 1e3, 3), "geometries": len(pinned), "disagreed_on": len(differ),
                                        "passes_of_disagreements": {p: sum(1 for k in differ if k[0] == p) for p in ("fwd", "dgrad", "wgrad")},
                                        "note": "plans timed in this process's first step, 20 timed steps after 6 warm-ups; the headline value runs the pinned set"}
            del live
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()
    if rank == 0:
        from sqd import nnops
        ms = elapsed / args.steps * 1e3
        default_cfg = not os.environ.get("SQD_BENCH_EXTRA")
        backbone = {"resnet": "ResNet-%d" % opts.num_layers, "resnet_lite": "ResNet-%d" % opts.num_layers, "resnet18_lite": "ResNet-18",
                    "eff_b5": "EfficientNet-b5", "tf_efficientnet_b5_ap": "EfficientNet-b5"}.get(opts.backbone, opts.backbone)
        out = {"metric": "train images/sec, ResNet-50 640x192" if default_cfg else "train images/sec, %s %dx%d" % (backbone, opts.width, opts.height),
               "value": round(world * opts.batch_size / (ms * 1e-3), 2),
               "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               # the type the path computes in: fp32 tensors, fp32 accumulation everywhere; per layer the convolutions multiply either
               # fp32 operands or their exact three-term bf16 split (fp32-level accuracy) — config.conv_arith counts which; --sqd_bf16
               # rounds the convolution operands to ONE bf16 term instead
               "dtype": "bf16" if opts.sqd_bf16 else "f32",
               "data": "synthetic (%d resident batches in rotation, copied into the step's input tensors inside the timed region)" % NBATCH,
               "config": {"workload": workload_name(opts),
                          "global_batch": world * opts.batch_size, "parallelism": "dp%d" % world,
                          "conv_arith": {"note": "layer geometries per pass by the kernel family their plan runs; 'bf16x3' = every fp32 "
                                                 "operand as the exact sum of three bf16 terms, 6 of 9 partial products on the bf16 matrix cores, "
                                                 "fp32 accumulation (error <= 4x the fp32-MFMA kernel's against fp64: tests/test_gpu_conv.py)"
                                                 "; 'f16x2' = every fp32 operand as two fp16 terms of the tensor scaled by a power of two, 3 of 4 partial "
                                                 "products on the fp16 matrix cores, fp32 accumulation (tested per plan against fp64, tests/test_gpu_f16x2.py: error <= "
                                                 "max(1.25 x the bf16x3 kernel's, the fp32-MFMA kernel's) + 5e-8 and <= 4 x the fp32-MFMA kernel's + 2e-7, "
                                                 "relative to the largest output)",
                                         "plans": plan_note if not opts.sqd_no_conv_tune else "cost model (fp32 MFMA)", **mix,
                                         "operand_scales": scales},
                          "exchange": exchange,
                          "operator_backends": nnops.backend_report()},
               "final_loss": round(loss, 6), "roofline": roof, "cpu_baseline": cpu, "diagnostics": diag}
        print(json.dumps(out))
    ddp.shutdown()


if __name__ == "__main__":
    main()
