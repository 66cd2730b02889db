"""Self-supervised SQLdepth trainer for MI355X — keeps the reference's orchestration surface
(reference trainer.py:30-687): Trainer(options), train(), run_epoch(), process_batch(inputs) ->
(outputs, losses), predict_poses(), generate_images_pred(inputs, outputs), compute_losses(inputs,
outputs), compute_reprojection_loss(), compute_depth_losses(), val(), log(), save_model(),
load_model(); the same dict keys for inputs / outputs / losses; the same checkpoint layout.

What differs by design (MI355X-first):
  * one process per GPU; gradients are averaged by sqd.ddp.GradBucketReducer (RCCL all-reduce
    overlapped with backward) instead of nn.DataParallel scatter/gather (reference trainer.py:74,93);
    every rank owns encoder, depth head AND pose net, and computes its own loss shard;
  * generate_images_pred + compute_losses run as ONE autograd node made of hand-written gfx950
    kernels (sqd.ops.PhotometricChain); the identity-reprojection maps, which depend only on the
    batch, are computed on a side HIP stream while the networks run;
  * the data source is synthetic KITTI-shaped frames unless real KITTI is present (no dataset I/O in
    this build).
Implemented: scale 0, PoseCNN on frame pairs or on all frames (--pose_model_input all), any temporal
frame_ids (0 -1 1, 0 -8 8, ...), mono / mono+stereo / stereo-only frame sets, and the reference's loss
options --no_ssim, --avg_reprojection, --disable_automasking (five option sets pinned by fixture G23);
--v1_multiscale / --predictive_mask and other scales raise NotImplementedError instead of silently
diverging (_check_supported)."""
import json
import os
import time

import numpy as np
import torch
import torch.nn.functional as F
import torch.optim as optim
from torch.utils.data import DataLoader

import datasets
import networks
from layers import compute_depth_errors
from sqd import ddp, nnkernels, ops
from sqd import lib as _sqd_lib
from sqd.optim import FusedAdam
from utils import normalize_image, sec_to_hm_str


class _NullWriter:
    def add_scalar(self, *a, **k):
        pass

    def add_image(self, *a, **k):
        pass


def _make_writer(path):
    try:
        from torch.utils.tensorboard.writer import SummaryWriter
        return SummaryWriter(path)
    except Exception:          # tensorboard is not installed in this image
        return _NullWriter()


class Trainer:
    def __init__(self, options):
        self.opt = options
        self.log_path = os.path.join(self.opt.log_dir, self.opt.model_name)
        self._check_supported()

        self.rank, self.world, self.local_rank = ddp.init_from_env()
        if self.opt.no_cuda or not torch.cuda.is_available():
            raise RuntimeError("the MI355X build runs the hot path in HIP kernels only (no CPU fallback); "
                               "the CPU restatement used for parity lives in oracle/ (test infrastructure)")
        self.device = torch.device("cuda", self.local_rank)
        torch.cuda.set_device(self.device)

        self.num_scales = len(self.opt.scales)
        self.num_input_frames = len(self.opt.frame_ids)
        self.num_pose_frames = 2 if self.opt.pose_model_input == "pairs" else self.num_input_frames
        assert self.opt.frame_ids[0] == 0, "frame_ids must start with 0"
        self.use_pose_net = not (self.opt.use_stereo and self.opt.frame_ids == [0])
        if self.opt.use_stereo and "s" not in self.opt.frame_ids:
            # the other camera of the stereo pair is one more source frame (reference trainer.py:52-53); a new list: the parser's
            # default is one shared object
            self.opt.frame_ids = list(self.opt.frame_ids) + ["s"]

        self.models = {}
        self.models["encoder"] = self._build_encoder().to(self.device)
        self.models["depth"] = self._build_depth_head().to(self.device)
        self.models["pose"] = networks.PoseCNN(self.num_input_frames if self.opt.pose_model_input == "all" else 2).to(self.device)
        if self.opt.load_pretrained_model:
            self._load_pretrained()
        if self.opt.pretrained_pose:
            sd = torch.load(os.path.join(self.opt.pose_net_path, "pose.pth"), map_location=self.device)
            self.models["pose"].load_state_dict({k.replace("module.", ""): v for k, v in sd.items()})

        from sqd import nnops
        nnops.configure(self.opt, self.device)
        if self.opt.sqd_channels_last:
            for m in self.models.values():
                m.to(memory_format=torch.channels_last)
        self.parameters_to_train = list(self.models["encoder"].parameters()) + list(self.models["depth"].parameters())
        if self.opt.diff_lr:
            self.pose_params = list(self.models["pose"].parameters())
            groups = [{"params": self.pose_params, "lr": self.opt.learning_rate / 10},
                      {"params": self.parameters_to_train, "lr": self.opt.learning_rate}]
            self.model_optimizer = FusedAdam(groups, lr=self.opt.learning_rate)
        else:
            self.parameters_to_train += list(self.models["pose"].parameters())
            self.model_optimizer = FusedAdam(self.parameters_to_train, self.opt.learning_rate)
        self.model_lr_scheduler = optim.lr_scheduler.StepLR(self.model_optimizer, self.opt.scheduler_step_size, 0.1)

        self._graph, self._graph_warm, self._capturing = None, 0, False
        self._hwc_keys = []                     # static source frames the captured step keeps channels_last (_capture)
        self.epoch, self.step, self.start_time = 0, 0, time.time()      # (train() resets them, as the reference does)
        all_params = [p for m in self.models.values() for p in m.parameters()]
        self.reducer = ddp.GradBucketReducer(all_params, self.opt.sqd_bucket_mb) if ddp.COMM is not None else None
        if self.reducer is not None:
            self.reducer.broadcast_parameters(self.models.values())
        # the sum of a weight gradient's pixel splits rides on the next BatchNorm-backward launch; with a reducer the filter is announced to
        # it by nnkernels.DEFERRED_GRAD_HOOK once that launch is enqueued (its post-accumulate hook never sees a directly assigned gradient),
        # so that N > 1 runs the kernels N = 1 is measured with
        self._defer_wgrad_reduce = not self.opt.sqd_no_defer_wgrad_reduce

        # single- and multi-rank runs replay the whole step as one hipGraph; with a process group the graph also holds the
        # bucket gathers and the RCCL all-reduces the autograd hooks launch, as branches parallel to the rest of backward
        # (--sqd_graph_ddp post: the round-1 variant — graph of forward+backward, collectives and Adam issued after the replay)
        self._graph_ok = not self.opt.sqd_no_graph and self.device.type == "cuda"
        self._side_stream = torch.cuda.Stream(device=self.device)
        self._pose_stream = torch.cuda.Stream(device=self.device)
        # weight-gradient kernels of the convolutions on their own stream (eager steps)
        self._wgrad_stream = torch.cuda.Stream(device=self.device)
        self._graph_stream = torch.cuda.Stream(device=self.device) if self._graph_ok else None
        self._build_loaders()
        self.writers = {m: (_make_writer(os.path.join(self.log_path, m)) if self.rank == 0 else _NullWriter())
                        for m in ("train", "val")}
        self.depth_metric_names = ["de/abs_rel", "de/sq_rel", "de/rms", "de/log_rms", "da/a1", "da/a2", "da/a3"]
        if self.rank == 0:
            print("Training model named:\n  ", self.opt.model_name)
            print("Models and tensorboard events files are saved to:\n  ", self.opt.log_dir)
            print("Training is using:\n  ", self.device, "x", self.world)
            self.save_opts()

    # ------------------------------------------------------------------------------- construction
    def _check_supported(self):
        o = self.opt
        unsupported = [n for n in ("v1_multiscale", "predictive_mask") if getattr(o, n)]
        if unsupported or list(o.scales) != [0] or o.pose_model_type != "posecnn" or o.pose_model_input not in ("pairs", "all"):
            raise NotImplementedError("MI355X hot path implements the reference's KITTI mono configuration "
                                      "(scale 0, posecnn on pairs or on all frames; --no_ssim / --avg_reprojection / --disable_automasking included); "
                                      "got %s scales=%s pose=%s/%s" % (unsupported, o.scales, o.pose_model_type, o.pose_model_input))
        # any temporal neighbours (reference trainer.py:315-337 loops over frame_ids[1:]; args_files/hisfog/mc and nyu train on 0 -8 8 and
        # 0 -16 16), optionally with the stereo frame; the photometric kernels take up to SQD_MAX_SOURCES source frames
        temporal = [f for f in o.frame_ids if f != "s"]
        if "s" in o.frame_ids and o.frame_ids[-1] != "s":
            raise NotImplementedError("the stereo frame \"s\" must be the last entry of frame_ids (the pose bookkeeping indexes the temporal "
                                      "frames by position, as reference trainer.py:52-53 appends it); got %s" % o.frame_ids)
        if not temporal or any(not isinstance(f, int) for f in temporal) or temporal[0] != 0 or len(set(temporal)) != len(temporal) or \
                (temporal == [0] and not o.use_stereo):
            raise NotImplementedError("frame_ids must be 0 followed by distinct temporal offsets (optionally with --use_stereo), or [0] with "
                                      "--use_stereo; got %s" % o.frame_ids)
        if len(temporal) - 1 + (1 if (o.use_stereo or "s" in o.frame_ids) else 0) > _sqd_lib.MAX_SOURCES:
            raise NotImplementedError("the photometric kernels take at most %d source frames (SQD_MAX_SOURCES); got frame_ids %s"
                                      % (_sqd_lib.MAX_SOURCES, o.frame_ids))
        if o.pose_model_input == "all" and 6 * (len(temporal) - 1) > 16:
            raise NotImplementedError("--pose_model_input all: the pose head kernel writes at most 16 channels (3 temporal frames); got %s" % o.frame_ids)

    def _build_encoder(self):
        o = self.opt
        if o.backbone in ("resnet", "resnet_lite"):
            return networks.ResnetEncoderDecoder(num_layers=o.num_layers, num_features=o.num_features, model_dim=o.model_dim)
        if o.backbone == "resnet18_lite":
            return networks.LiteResnetEncoderDecoder(model_dim=o.model_dim)
        if o.backbone == "eff_b5":
            return networks.BaseEncoder.build(num_features=o.num_features, model_dim=o.model_dim)
        # every other name is a timm backbone under the U-Net decoder (reference trainer.py:63-64): convnext_large, tf_efficientnet_b5_ap
        return networks.Unet(pretrained=(not o.load_pretrained_model), backbone=o.backbone, in_channels=3,
                             num_classes=o.model_dim, decoder_channels=o.dec_channels)

    def _build_depth_head(self):
        o = self.opt
        cls = networks.Lite_Depth_Decoder_QueryTr if o.backbone.endswith("_lite") else networks.Depth_Decoder_QueryTr
        return cls(in_channels=o.model_dim, patch_size=o.patch_size, dim_out=o.dim_out, embedding_dim=o.model_dim,
                   query_nums=o.query_nums, num_heads=4, min_val=o.min_depth, max_val=o.max_depth)

    def _load_pretrained(self):
        for name, fname in (("encoder", "encoder.pth"), ("depth", "depth.pth")):
            sd = torch.load(os.path.join(self.opt.load_pt_folder, fname), map_location=self.device)
            own = self.models[name].state_dict()
            self.models[name].load_state_dict({k: v for k, v in sd.items() if k in own})

    def _build_loaders(self):
        o = self.opt
        if not o.sqd_synthetic:
            raise NotImplementedError("real KITTI input is outside this build (SURVEY.md §2: datasets are host I/O): pass "
                                      "--sqd_synthetic to train on synthetic KITTI-shaped frames (data_path=%r is not read)" % o.data_path)
        n = o.sqd_synthetic_len
        train = datasets.SyntheticKITTIDataset(o.height, o.width, o.frame_ids, n, offset=self.rank * n)
        val = datasets.SyntheticKITTIDataset(o.height, o.width, o.frame_ids, max(o.batch_size, n // 10), offset=10 ** 6)
        self.num_total_steps = len(train) // o.batch_size * o.num_epochs
        self.train_loader = DataLoader(train, o.batch_size, True, num_workers=o.num_workers, pin_memory=True, drop_last=True)
        self.val_loader = DataLoader(val, o.batch_size, True, num_workers=o.num_workers, pin_memory=True, drop_last=True)
        self.val_iter = iter(self.val_loader)

    def set_train(self):
        for m in self.models.values():
            m.train()

    def set_eval(self):
        for m in self.models.values():
            m.eval()

    # ------------------------------------------------------------------------------------ epochs
    def train(self):
        self.epoch, self.step, self.start_time = 0, 0, time.time()
        self.save_model()
        for self.epoch in range(self.opt.num_epochs):
            self.run_epoch()
            self.model_lr_scheduler.step()
            if (self.epoch + 1) % self.opt.save_frequency == 0:
                self.save_model()

    def train_step(self, inputs):
        """forward + backward + Adam on one batch (reference trainer.py:240-244).  On a single device the step is
        captured into a hipGraph after a few eager steps and replayed from then on (--sqd_no_graph: always eager)."""
        self._steps_run = getattr(self, "_steps_run", 0) + 1
        if self._steps_run == 5 and self.opt.sqd_save_conv_plans and self.rank == 0:
            # every layer's forward / data-gradient / weight-gradient plan has been timed (step 1) and, in graph mode, frozen
            # into the capture (step 4): the file pins this exact kernel set for a later run (--sqd_conv_plans)
            with open(self.opt.sqd_save_conv_plans, "w") as f:
                json.dump(nnkernels.export_plans(), f)
        if self._graph_ok:
            if self._graph is not None or self._graph_warm >= 3:
                return self._train_step_graphed(inputs)
            # the eager warm-up steps run on the stream the capture will use: autograd's AccumulateGrad nodes remember
            # the stream they were created on, and a capture must not touch the default stream
            self._graph_warm += 1
            cur = torch.cuda.current_stream()
            self._graph_stream.wait_stream(cur)
            with torch.cuda.stream(self._graph_stream):
                res = self._train_step_eager(inputs)
            cur.wait_stream(self._graph_stream)
            return res
        return self._train_step_eager(inputs)

    def _train_step_graphed(self, inputs):
        if not self.opt.sqd_device_noise and ("noise", 0) not in inputs:
            # host RNG as in the reference (trainer.py:516): drawn outside the graph, handed over as a static input
            o = self.opt
            inputs[("noise", 0)] = torch.randn(o.batch_size, self._identity_planes(), o.height, o.width)
        if self._graph is None:
            err = self._try_capture(inputs)
            if err is not None and self.reducer is not None and self.opt.sqd_graph_ddp != "post":
                # multi-rank: the capture with the all-reduces inside could not be taken — fall back to the graph of forward + backward with
                # the exchange and Adam after the replay (every rank runs the same code on the same shapes: all of them land here)
                self._note_capture_failure("overlap", err, "retrying as --sqd_graph_ddp post")
                self.opt.sqd_graph_ddp = "post"
                self._graph_ok = True
                self.reducer.hooks_enabled = True
                self.reducer.reattach_grad_views()
                err = self._try_capture(inputs)
            if err is not None:
                # (the eager path needs nothing the capture would have set up)
                self._note_capture_failure(self.graph_mode(), err, "continuing with eager steps")
                self._graph, self._graph_ok = None, False
                if self.reducer is not None:
                    self.reducer.hooks_enabled = True
                return self._train_step_eager(inputs)
        # every batch is copied into the graph's input tensors (a loader never feeds one twice): device-resident tensors of the static
        # tensors' dtype in ONE multi-tensor launch, whatever else (host tensors: a loader's pinned batch) one by one
        dst, src, pk_dst, pk_src = [], [], [], []
        for k, v in inputs.items():
            st = self._static_in[k]
            if v is not st:
                if v.is_cuda and v.dtype == st.dtype and v.shape == st.shape and k in self._hwc_keys and v.is_contiguous():
                    pk_dst.append(st)             # planar batch -> the channels_last static frame: one transposing launch for all of them
                    pk_src.append(v)
                elif v.is_cuda and v.dtype == st.dtype and v.shape == st.shape and k not in self._hwc_keys:
                    dst.append(st)
                    src.append(v)
                else:
                    st.copy_(v, non_blocking=True)
            inputs[k] = st                        # as process_batch does in eager mode: the caller's dict now holds device tensors
        if dst:
            torch._foreach_copy_(dst, src)
        if pk_dst:
            ops.pack_pixels(pk_src, out=pk_dst)
        # (the captured step takes the convolution filters' operand-scale records from the previous step's Adam launch: refresh them here if
        #  something else — load_model, an initialiser — has written the weights since)
        if not getattr(self, "_graph_refreshes_records", True):
            nnkernels.refresh_filter_records_if_stale()
        if self.reducer is None or self.opt.sqd_graph_ddp != "post":
            self.model_optimizer.refresh_hyper()
            self._graph.replay()                # forward, backward (+ bucketed all-reduces overlapped with it), Adam
        else:                                   # the graph holds forward + backward; exchange, then Adam
            self._graph.replay()
            self.reducer.allreduce_all()
            self.model_optimizer.step()
        return self._static_out

    def _try_capture(self, inputs):
        """-> None, or the exception that kept the step from being captured (the device is idle again when it returns)"""
        try:
            self._capture(inputs)
            return None
        except Exception as e:                  # noqa: BLE001 — any capture failure: the caller decides how to keep training
            self._graph, self._capturing = None, False
            torch.cuda.synchronize()
            return e

    def _note_capture_failure(self, mode, e, then):
        import sys
        self.capture_failures = getattr(self, "capture_failures", []) + ["%s: %s: %s" % (mode, type(e).__name__, (str(e).splitlines() or [""])[0])]
        print("sqd: hipGraph capture of the training step (%s) failed (%s: %s) — %s" %
              (mode, type(e).__name__, (str(e).splitlines() or [""])[0], then), file=sys.stderr, flush=True)

    def graph_mode(self):
        """what a step runs as right now: 'eager', 'graph' (single rank), 'overlap' (all-reduces captured in the step graph) or
        'post' (graph of forward + backward, then all-reduce + Adam) — bench lines and logs quote it"""
        if not self._graph_ok:
            return "eager"
        if self.reducer is None:
            return "graph"
        return "post" if self.opt.sqd_graph_ddp == "post" else "overlap"

    def _capture(self, inputs):
        """One hipGraph for process_batch + backward + Adam.  The ~1200 kernel launches of a step then cost one graph
        launch on the host (eager: ~23 ms of host time per 26 ms step)."""
        self._static_in = {k: v.to(self.device).clone() for k, v in inputs.items()}
        # the full-resolution source frames have the photometric kernels as their only readers, and those fetch a pixel's taps with fewer
        # gathers from [B,H,W,3] memory: the step's static copies of these frames are channels_last tensors (same shape, same values — the
        # copy-in of a batch transposes instead of copying, sqd_pack_pixels), where every kernel of the chain reads that layout
        self._hwc_keys = []
        o = self.opt
        src_ids = [f for f in o.frame_ids[1:]]
        if "s" not in src_ids and ops.sources_hwc_ok(o.batch_size, len(src_ids), o.height, o.width, 0, self._loss_flags()):
            self._hwc_keys = [("color", f, 0) for f in src_ids if ("color", f, 0) in self._static_in]
            for k in self._hwc_keys:
                self._static_in[k] = self._static_in[k].contiguous(memory_format=torch.channels_last)
        if self.reducer is not None and self.opt.sqd_graph_ddp == "post":
            return self._capture_fwd_bwd()
        opt = self.model_optimizer
        if self.reducer is not None:
            if self.reducer.buckets is None:
                raise RuntimeError("graph capture needs one eager step first (the gradient buckets are built there)")
            self.reducer.zero_grad()            # p.grad = None, bucket arrival counters reset: the captured backward fills them
        else:
            opt.zero_grad(set_to_none=True)
        opt.begin_capture()
        opt.refresh_hyper()                     # allocates the device scalars; the step counts it adds are undone below
        undone = set()
        for st in opt.state.values():
            if "step" in st and id(st["step"]) not in undone:      # (a group's parameters may share one counter tensor)
                undone.add(id(st["step"]))
                st["step"] -= 1
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        self._capturing = True
        # with a communicator: RCCL's own helper threads may call into the HIP runtime while this thread captures; only
        # this thread's calls are part of the capture ("thread_local"), theirs are none of its business
        mode = {} if self.reducer is None else {"capture_error_mode": "thread_local"}
        try:
            with torch.cuda.graph(g, stream=self._graph_stream, **mode):
                try:
                    # per-step use counts of the filters (a filter used once may defer its gradient's split sum); does the captured step
                    # refresh the filters' operand-scale records itself, or does it rely on the previous step's Adam launch for them?
                    self._graph_refreshes_records = nnkernels.begin_step()
                    outputs, losses = self.process_batch(self._static_in)
                    # the reducer's post-accumulate hooks run here, inside the capture: each full bucket is gathered by one
                    # multi-tensor copy and its all-reduce (sqd_comm_allreduce, a plain stream operation) is enqueued on the
                    # communicator's stream — a branch of the graph that runs next to the remaining backward kernels; finish()
                    # joins the branches before Adam reads the averaged buckets
                    self._backward(losses["loss"])
                    if self.reducer is not None:
                        self.reducer.finish()
                    opt.step()
                except BaseException:
                    self._join_captured_branches()
                    raise
        finally:
            self._capturing = False
        self._graph, self._static_out = g, (outputs, losses)

    def _join_captured_branches(self):
        """A step that raises in the middle of a capture leaves its forked branches (pose network, weight gradients, the communicator's
        stream) un-joined, and a capture with un-joined work cannot be ended: every stream of it would stay in capturing state and the
        next capture attempt — the fallback — would die with it.  Join whatever is still capturing, so that the context manager can end
        (and discard) the capture cleanly."""
        cur = torch.cuda.current_stream()
        streams = [self._pose_stream, self._side_stream, self._wgrad_stream]
        if self.reducer is not None and getattr(self.reducer.comm, "stream", None) is not None:
            streams.append(self.reducer.comm.stream)
            self.reducer._inflight = False
        for s in streams:
            try:
                with torch.cuda.stream(s):
                    branch = torch.cuda.is_current_stream_capturing()
                if branch and s.cuda_stream != cur.cuda_stream:
                    cur.wait_stream(s)
            except Exception:                    # noqa: BLE001 — best effort: the original exception is what gets reported
                pass
        nnkernels.WGRAD_STREAM = None
        nnkernels._PENDING_WGRAD.clear()

    def _capture_fwd_bwd(self):
        """Multi-rank variant: process_batch + the bucket memsets + backward in one hipGraph.  The gradients are the bucket
        views GradBucketReducer built during the eager warm-up steps, so the replay leaves them ready for the all-reduce;
        the reducer's autograd hooks stay off (nothing Python-side runs during a replay)."""
        if self.reducer.buckets is None:
            raise RuntimeError("graph capture needs one eager step first (the gradient buckets are built there)")
        self.reducer.hooks_enabled = False
        views = self.reducer.detach_grad_views()          # {param: bucket view}; p.grad = None for the capture
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        self._capturing = True
        try:
            with torch.cuda.graph(g, stream=self._graph_stream, capture_error_mode="thread_local"):
                try:
                    self._graph_refreshes_records = nnkernels.begin_step()
                    outputs, losses = self.process_batch(self._static_in)
                    self._backward(losses["loss"])
                    # fresh gradients (assigned, not accumulated: no memsets, no ~170 accumulate launches) -> the buckets, as
                    # a couple of multi-tensor copies
                    params = [p for p in views if p.grad is not None]
                    torch._foreach_copy_([views[p] for p in params], [p.grad for p in params])
                except BaseException:
                    self._join_captured_branches()
                    raise
        finally:
            self._capturing = False
        for p, v in views.items():                        # the optimiser reads the (all-reduced) bucket memory
            p.grad = v
        self._graph, self._static_out = g, (outputs, losses)

    def _backward(self, loss):
        # eager steps: the convolutions' weight gradients run on their own stream, next to the data gradients (-0.8 ms per
        # step).  Not inside a capture: a hipGraph with ~110 extra cross-branch edges replays 1.3 ms slower than the linear one.
        gb = int(getattr(self.opt, "sqd_graph_wgrad_batch", 0) or 0)
        nnkernels.WGRAD_STREAM = None if getattr(self, "_capturing", False) and gb <= 0 else self._wgrad_stream
        nnkernels.WGRAD_BATCH = gb if getattr(self, "_capturing", False) and gb > 0 else 1
        nnkernels.DEFER_WGRAD_REDUCE = self._defer_wgrad_reduce
        # (the first multi-rank step builds the buckets from whatever gradients exist: nothing to announce yet)
        nnkernels.DEFERRED_GRAD_HOOK = self.reducer.on_deferred_grad if self.reducer is not None and self.reducer.buckets is not None else None
        if self.reducer is not None and self.reducer.buckets is None:
            nnkernels.DEFER_WGRAD_REDUCE = False         # (its parameters have no hooks yet, but finish() reads every gradient right after)
        ops.UNIT_UPSTREAM = loss is getattr(self, "_chain_total", None)      # (one scale: the loss is the chain's total itself, its seed is 1)
        try:
            loss.backward()
            nnkernels.join_wgrad_stream()                # the caller's stream joins it before anything reads the gradients
        finally:
            ops.UNIT_UPSTREAM = False
            self._chain_total = None
            nnkernels.WGRAD_STREAM = None
            nnkernels.DEFER_WGRAD_REDUCE = False
            nnkernels.DEFERRED_GRAD_HOOK = None
            nnkernels.drop_pending_reduce()              # (empty after a pass that returned; a pass that raised must not leak its sums)

    def _train_step_eager(self, inputs):
        nnkernels.begin_step()
        outputs, losses = self.process_batch(inputs)
        if self.reducer is not None:
            self.reducer.zero_grad()
        else:
            self.model_optimizer.zero_grad(set_to_none=True)
        self._backward(losses["loss"])
        if self.reducer is not None:
            self.reducer.finish()
        self.model_optimizer.step()
        return outputs, losses

    def run_epoch(self):
        if self.rank == 0:
            print("Training")
        self.set_train()
        for batch_idx, inputs in enumerate(self.train_loader):
            before = time.time()
            outputs, losses = self.train_step(inputs)
            early = batch_idx % self.opt.log_frequency == 0 and self.step < 2000
            late = self.step % 1000 == 0
            if early or late:
                loss = losses["loss"].detach().cpu()           # synchronises: duration below is device-true
                self.log_time(batch_idx, time.time() - before, loss)
                if "depth_gt" in inputs:
                    self.compute_depth_losses(inputs, outputs, losses)
                self.log("train", inputs, outputs, losses)
                self.val()
            self.step += 1

    # ----------------------------------------------------------------------------- forward pass
    def process_batch(self, inputs):
        """Pass a minibatch through the networks and the photometric chain -> (outputs, losses)."""
        for key, ipt in inputs.items():
            inputs[key] = ipt.to(self.device, non_blocking=True)
        if not self._capturing or self.opt.sqd_early_identity:
            self._launch_identity(inputs)
        # (inside the captured step the identity maps are evaluated right before the fused warp + SSIM kernel instead — see
        #  generate_images_pred: a side stream buys nothing in the graph, and the pass over the target and source frames leaves them
        #  in the Infinity Cache for the kernel that gathers from them next, 15 ms of convolution traffic after the stems read them)
        nnkernels.defer_bn_counters(True)
        try:
            fork = self._capturing and self.use_pose_net
            if fork:
                # inside the graph the pose network (which only needs the input frames) is a parallel branch: its small
                # convolutions — and, through autograd's stream bookkeeping, their backward — fill the gaps the
                # latency-bound ViT / MLP launches of the depth head leave on the device
                cur = torch.cuda.current_stream()
                self._pose_stream.wait_stream(cur)
                with torch.cuda.stream(self._pose_stream):
                    pose_outputs = self.predict_poses(inputs, None)
            enc = self.models["encoder"]
            frame = inputs["color_aug", 0, 0]
            # (the ResNet stems take the dense NCHW frame: layout conversion and normalisation happen in their space-to-depth pass)
            features = enc(frame if getattr(enc, "planar_input", False) and frame.is_contiguous() else self._fmt(frame))
            outputs = self.models["depth"](features)
            if fork:
                cur.wait_stream(self._pose_stream)
                outputs.update(pose_outputs)
            elif self.use_pose_net:
                outputs.update(self.predict_poses(inputs, features))
        finally:
            nnkernels.flush_bn_counters()              # one multi-tensor += 1 for every BatchNorm that ran in training mode
            nnkernels.defer_bn_counters(False)
        self.generate_images_pred(inputs, outputs)
        losses = self.compute_losses(inputs, outputs)
        return outputs, losses

    def _fmt(self, x):
        return x.contiguous(memory_format=torch.channels_last) if self.opt.sqd_channels_last else x

    def _loss_flags(self):
        """SQD_LOSS_* bits of the reference's loss options (trainer.py:447-451, 480-524); the mean over ONE source frame is that frame."""
        o = self.opt
        return _sqd_lib.loss_flags(o.no_ssim, o.avg_reprojection and len(o.frame_ids) - 1 > 1, o.disable_automasking)

    def _identity_planes(self):
        return 1 if self._loss_flags() & _sqd_lib.LOSS_AVG_REPROJECTION else len(self.opt.frame_ids) - 1

    def _launch_identity(self, inputs):
        """Identity-reprojection maps + tie-break noise (reference trainer.py:480-487,514-517) depend only
        on the batch: compute them on a side stream, overlapped with the encoder forward."""
        srcs = [inputs[("color", f, 0)] for f in self.opt.frame_ids[1:]]
        tgt = inputs[("color", 0, 0)]
        B, _, H, W = tgt.shape
        flags, NI = self._loss_flags(), self._identity_planes()
        if flags & _sqd_lib.LOSS_NO_AUTOMASK:           # no identity candidates (trainer.py:520-521): nothing to compute, no noise drawn
            self._identity, self._identity_done = False, None
            return
        if ("noise", 0) in inputs:
            noise = inputs[("noise", 0)]
        elif self.opt.sqd_device_noise:
            noise = torch.randn(B, NI, H, W, device=self.device)
        else:
            noise = torch.randn(B, NI, H, W).to(self.device, non_blocking=True)   # CPU RNG, as the reference
        if self._capturing:
            # inside a graph capture this stays on the capturing stream: as a forked branch it made the replay 1.4 ms SLOWER
            # (22.28 vs 20.92 ms, same box) — unlike the pose-network branch of process_batch, which gains 0.4 ms
            self._identity, self._identity_done = ops.identity_fwd(tgt, srcs, noise, loss_flags=flags), None
            return
        side = self._side_stream
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self._identity = ops.identity_fwd(tgt, srcs, noise, loss_flags=flags)
            self._identity_done = torch.cuda.Event()
            self._identity_done.record(side)
        for t in (tgt, noise, *srcs):
            t.record_stream(side)

    def predict_poses(self, inputs, features):
        """Pose of each source frame relative to the target, pairs in temporal order (reference
        trainer.py:301-337)."""
        outputs = {}
        aug = {f: inputs["color_aug", f, 0] for f in self.opt.frame_ids if f != "s"}
        srcs = [f for f in self.opt.frame_ids[1:] if f != "s"]
        # PoseCNN has no batch statistics, so the pairs of all source frames go through it as ONE batch [S*B,6,H,W] (the
        # reference calls it once per pair, trainer.py:319-334): same numbers per sample, half the launches, and every
        # pose filter is used once per step (its weight gradient is one kernel, not a sum of two).
        # Row b*S + i of the batch is pair i of sample b, so that the head's outputs ARE the [B,S,3] axis-angle / translation
        # arrays the photometric chain reads (no slicing, cat or copies in between, forward or backward).
        B, S = aug[0].shape[0], len(srcs)
        if self.num_pose_frames != 2:
            return self._predict_poses_all(aug, srcs)
        pairs = [((aug[f], aug[0]) if f < 0 else (aug[0], aug[f])) for f in srcs]
        axisangle, translation = self.models["pose"].forward_pairs(pairs)      # [B*S,1,1,3] each; the 6-channel batch is never built
        if not (axisangle.is_contiguous() and translation.is_contiguous()):
            axisangle, translation = axisangle.contiguous(), translation.contiguous()
        aa_all, tr_all = axisangle.view(B, S, 3), translation.view(B, S, 3)
        # the un-scaled cam_T_cam of every pair (reference trainer.py:336-337) in one launch; detached, as the stand-alone
        # transformation_from_parameters returns it (the training path differentiates the pose through PhotometricChain)
        eye = getattr(self, "_eye4", None)
        if eye is None or eye.shape[0] != B or eye.device != aa_all.device:
            eye = self._eye4 = torch.eye(4, device=aa_all.device).repeat(B, 1, 1)
        _, T_all, _ = ops.pose_mats_fwd(aa_all.detach(), tr_all.detach(), [1 if f < 0 else 0 for f in srcs], eye)
        for i, f in enumerate(srcs):
            outputs[("axisangle", 0, f)] = aa_all[:, i].view(B, 1, 1, 3)
            outputs[("translation", 0, f)] = tr_all[:, i].view(B, 1, 1, 3)
            outputs[("cam_T_cam", 0, f)] = T_all[:, i]
        # generate_images_pred reads the [B,S,3] arrays as they are when it is handed these very outputs
        self._pose_all = (aa_all, tr_all, tuple(outputs[("axisangle", 0, f)] for f in srcs))
        return outputs

    def _predict_poses_all(self, aug, srcs):
        """--pose_model_input all (reference trainer.py:339-361): ONE pass of the pose network over the channel concatenation of every
        temporal frame, in frame_ids order; pose i of its [B, F-1, 1, 3] outputs belongs to frame_ids[1 + i], none is inverted, and every
        frame's ("axisangle", 0, f) / ("translation", 0, f) entry is the whole tensor."""
        outputs = {}
        x = torch.cat([aug[f] for f in [0] + srcs], 1).contiguous()                      # [B, 3 F, H, W]
        axisangle, translation = self.models["pose"](x)                                   # [B, F-1, 1, 3] each
        B, S = x.shape[0], len(srcs)
        aa_all, tr_all = axisangle.reshape(B, S, 3), translation.reshape(B, S, 3)
        eye = getattr(self, "_eye4", None)
        if eye is None or eye.shape[0] != B or eye.device != aa_all.device:
            eye = self._eye4 = torch.eye(4, device=aa_all.device).repeat(B, 1, 1)
        _, T_all, _ = ops.pose_mats_fwd(aa_all.detach().contiguous(), tr_all.detach().contiguous(), [0] * S, eye)
        for i, f in enumerate(srcs):
            outputs[("axisangle", 0, f)] = axisangle
            outputs[("translation", 0, f)] = translation
            outputs[("cam_T_cam", 0, f)] = T_all[:, i]
        self._pose_all = None
        return outputs

    def generate_images_pred(self, inputs, outputs):
        """Warps the source frames into the target view (reference trainer.py:386-439) — and, because the
        whole photometric chain is one fused autograd node here, also evaluates the losses that
        compute_losses() then reports."""
        o = self.opt
        srcs_ids = o.frame_ids[1:]
        if getattr(self, "_identity", None) is None:
            self._launch_identity(inputs)
        if self._identity_done is not None:
            torch.cuda.current_stream().wait_event(self._identity_done)
        identity, self._identity = self._identity, None
        if identity is False:
            identity = None
        pose_ids = [f for f in srcs_ids if f != "s"]
        B = inputs[("color", 0, 0)].shape[0]
        cached, self._pose_all = getattr(self, "_pose_all", None), None
        if pose_ids and cached is not None and len(cached[2]) == len(pose_ids) and \
                all(outputs[("axisangle", 0, f)] is t for f, t in zip(pose_ids, cached[2])):
            aa, tr = cached[:2]                                                                          # [B,Sp,3] (predict_poses)
        elif pose_ids and self.num_pose_frames != 2 and o.use_stereo:
            # --pose_model_input all with --use_stereo: T is cam_T_cam of predict_poses (reference trainer.py:408, 412): pose i, not inverted
            aa = outputs[("axisangle", 0, pose_ids[0])][:, :len(pose_ids), 0].contiguous()
            tr = outputs[("translation", 0, pose_ids[0])][:, :len(pose_ids), 0].contiguous()
        elif pose_ids:
            # (with --pose_model_input all and no stereo frame the reference rebuilds T from axisangle[:, 0] / translation[:, 0] of the
            # shared [B, F-1, 1, 3] tensors for EVERY frame, inverted for negative ids — trainer.py:414-421 — and so does this)
            aa = torch.cat([outputs[("axisangle", 0, f)][:, 0] for f in pose_ids], 1).contiguous()
            tr = torch.cat([outputs[("translation", 0, f)][:, 0] for f in pose_ids], 1).contiguous()
        else:
            aa = tr = torch.zeros(B, 0, 3, device=self.device)
        # --use_stereo: T of the temporal frames is the un-scaled cam_T_cam, T of "s" is inputs["stereo_T"] (reference
        # trainer.py:405-421: the mean-inverse-depth scaling of the translation is skipped)
        all_stereo = self.num_pose_frames != 2 and o.use_stereo
        meta = dict(H=o.height, W=o.width, invert=[1 if (f < 0 and not all_stereo) else 0 for f in pose_ids], smooth_weight=o.disparity_smoothness,
                    use_stereo=bool(o.use_stereo), stereo_T=inputs["stereo_T"] if "s" in srcs_ids else None,
                    loss_flags=self._loss_flags())
        # (channels_last frames — the captured step's static source frames — go through as they are: the chain's kernels read that layout)
        srcs = [t if t.is_contiguous(memory_format=torch.channels_last) else t.contiguous() for t in (inputs[("color", f, 0)] for f in srcs_ids)]
        res = ops.PhotometricChain.apply(outputs[("disp", 0)].contiguous(), aa, tr, inputs[("K", 0)].contiguous(),
                                         inputs[("inv_K", 0)].contiguous(), inputs[("color", 0, 0)].contiguous(),
                                         identity, meta, *srcs)
        total, photo, smooth, depth, sel, T = res[:6]
        S = len(srcs)
        outputs[("depth", 0, 0)] = depth
        for i, f in enumerate(srcs_ids):
            outputs[("sample", f, 0)] = res[6 + i]
            outputs[("color", f, 0)] = res[6 + S + i]
            outputs[("color_identity", f, 0)] = inputs[("color", f, 0)]
        outputs[("_chain", 0)] = (total, sel)

    def compute_reprojection_loss(self, pred, target):
        """0.85*SSIM + 0.15*L1 between a predicted and a target image (reference trainer.py:441-453);
        stand-alone form for scripts — the training path evaluates it inside the fused kernel.  Differentiable w.r.t. both images
        (sqd_ssim_bwd + autograd's own abs / mean), as the reference's is."""
        l1 = torch.abs(target - pred).mean(1, True)
        return 0.85 * ops.ssim_map(pred, target).mean(1, True) + 0.15 * l1

    def compute_losses(self, inputs, outputs):
        """Reprojection + smoothness loss of the minibatch (reference trainer.py:455-549)."""
        if ("_chain", 0) not in outputs:
            self.generate_images_pred(inputs, outputs)
        total, sel = outputs.pop(("_chain", 0))
        self._chain_total = total
        if not self.opt.disable_automasking:             # (trainer.py:523-525)
            outputs["identity_selection/0"] = sel
        loss = total / self.num_scales if self.num_scales != 1 else total          # (scale 0 only: no division kernel)
        return {"loss/0": total, "loss": loss}

    def compute_depth_losses(self, inputs, outputs, losses):
        """Depth metrics for monitoring (reference trainer.py:551-579): 375x1242, eigen crop, batch-global
        median scaling."""
        pred = torch.clamp(F.interpolate(outputs[("depth", 0, 0)].detach(), [375, 1242], mode="bilinear", align_corners=False),
                           1e-3, 80)
        gt = inputs["depth_gt"]
        mask = gt > 0
        crop = torch.zeros_like(mask)
        crop[:, :, 153:371, 44:1197] = 1
        mask = mask * crop
        gt, pred = gt[mask], pred[mask]
        pred = torch.clamp(pred * (torch.median(gt) / torch.median(pred)), min=1e-3, max=80)
        for name, v in zip(self.depth_metric_names, compute_depth_errors(gt, pred)):
            losses[name] = np.array(v.cpu())

    def val(self):
        self.set_eval()
        try:
            inputs = next(self.val_iter)
        except StopIteration:
            self.val_iter = iter(self.val_loader)
            inputs = next(self.val_iter)
        with torch.no_grad():
            outputs, losses = self.process_batch(inputs)
            if "depth_gt" in inputs:
                self.compute_depth_losses(inputs, outputs, losses)
            self.log("val", inputs, outputs, losses)
        self.set_train()

    # ------------------------------------------------------------------------- logging / state
    def log_time(self, batch_idx, duration, loss):
        if self.rank != 0:
            return
        sps = self.opt.batch_size * self.world / duration
        sofar = time.time() - self.start_time
        left = (self.num_total_steps / self.step - 1.0) * sofar if self.step > 0 else 0
        print("epoch {:>3} | batch {:>6} | examples/s: {:5.1f} | loss: {:.5f} | time elapsed: {} | time left: {}".format(
            self.epoch, batch_idx, sps, float(loss), sec_to_hm_str(sofar), sec_to_hm_str(left)))

    def log(self, mode, inputs, outputs, losses):
        writer = self.writers[mode]
        for name, v in losses.items():
            writer.add_scalar(name, float(v), self.step)
        for j in range(min(4, self.opt.batch_size)):
            for f in self.opt.frame_ids:
                writer.add_image("color_{}_0/{}".format(f, j), inputs[("color", f, 0)][j].data, self.step)
                if f != 0:
                    writer.add_image("color_pred_{}_0/{}".format(f, j), outputs[("color", f, 0)][j].data, self.step)
            writer.add_image("disp_0/{}".format(j), normalize_image(outputs[("disp", 0)][j]), self.step)
            if not self.opt.disable_automasking:
                writer.add_image("automask_0/{}".format(j), outputs["identity_selection/0"][j][None, ...], self.step)

    def save_opts(self):
        models_dir = os.path.join(self.log_path, "models")
        os.makedirs(models_dir, exist_ok=True)
        with open(os.path.join(models_dir, "opt.json"), "w") as f:
            json.dump(self.opt.__dict__.copy(), f, indent=2)

    def save_model(self):
        """weights_{epoch}/{encoder,depth,pose,adam}.pth with the reference's key names (trainer.py:638-660)."""
        if self.rank != 0:
            return
        folder = os.path.join(self.log_path, "models", "weights_{}".format(self.epoch))
        os.makedirs(folder, exist_ok=True)
        for name, model in self.models.items():
            sd = model.state_dict()
            if name == "encoder":
                sd["height"], sd["width"], sd["use_stereo"] = self.opt.height, self.opt.width, self.opt.use_stereo
            torch.save(sd, os.path.join(folder, "{}.pth".format(name)))
        torch.save(self.model_optimizer.state_dict(), os.path.join(folder, "adam.pth"))

    def load_model(self):
        folder = os.path.expanduser(self.opt.load_weights_folder)
        assert os.path.isdir(folder), "Cannot find folder {}".format(folder)
        for n in self.opt.models_to_load:
            if n not in self.models:
                continue
            own = self.models[n].state_dict()
            sd = torch.load(os.path.join(folder, "{}.pth".format(n)), map_location=self.device)
            own.update({k: v for k, v in sd.items() if k in own})
            self.models[n].load_state_dict(own)
        adam = os.path.join(folder, "adam.pth")
        if os.path.isfile(adam):
            self.model_optimizer.load_state_dict(torch.load(adam, map_location=self.device))
