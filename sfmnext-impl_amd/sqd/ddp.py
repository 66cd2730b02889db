"""Data-parallel gradient exchange for one-process-per-GPU training: RCCL over xGMI through the library's own
communicator (include/sqd.h §19; torch.distributed "gloo" carries the rendezvous and, on CPU, the tests).

Design: every rank owns a full replica and a shard of the batch; after backward the
gradients are averaged with a small number of large all-reduces.  Parameters are grouped into a few
flat fp32 buckets (in reverse registration order, i.e. roughly the order backward produces their
gradients); a post-accumulate-grad hook counts arrivals and, the moment a bucket's last gradient
lands, gathers the bucket with one multi-tensor copy, re-points the parameters' .grad at the bucket
views and launches the bucket's all-reduce asynchronously, so the exchange overlaps the rest of
backward.  (Accumulating straight into zeroed bucket views instead costs a memset plus one small
accumulate launch per parameter — ~170 of them, +0.7 ms per step on MI355X.)
xGMI is point-to-point (7 links x ~153 GB/s per GPU): a 32 MB bucket is one ~0.2-0.5 ms collective,
large enough to run at link bandwidth, small enough that the last bucket's tail is short.

Parameters that never receive a gradient (the torchvision-compatible `fc` of the ResNet trunk —
SURVEY App. B-11) are detected on the first step and left out of the buckets."""
import ctypes
import os

import torch
import torch.distributed as dist

# the host driver supports dmabuf IPC only: RCCL's peer mappings need this in the environment before the HIP runtime starts (it is exported
# on the pool's boxes already; set here at import, long before the first device call, for launchers that scrub the environment)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

COMM = None          # the data-plane communicator of this process (RcclComm on a GPU, GlooComm in the CPU tests), set by init_from_env


class RcclComm:
    """RCCL communicator owned by libsqd.so (include/sqd.h §19, csrc/comm.hip): collectives are plain operations on the
    stream they are given — they overlap and capture into a hipGraph like kernels, and nothing polls events from a helper
    thread (ProcessGroupNCCL's watchdog did, and aborted in-capture exchanges)."""
    _DT = {torch.float32: 0, torch.float64: 1, torch.int32: 2, torch.uint8: 3}
    _OP = {"sum": 0, "avg": 1, "max": 2, "min": 3}
    device_averages = True

    def __init__(self, rank, world, exchange_id):
        """exchange_id(bytes_or_None) -> bytes: ships rank 0's unique id to every rank (the job's existing control channel)."""
        from .lib import lib
        self._lib = lib()
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        self._check(self._lib.sqd_comm_load(path.encode() if os.path.exists(path) else None))
        buf = ctypes.create_string_buffer(128)
        if rank == 0:
            self._check(self._lib.sqd_comm_unique_id(buf))
        uid = exchange_id(buf.raw if rank == 0 else None)
        assert len(uid) == 128
        handle = ctypes.c_void_p()
        self._check(self._lib.sqd_comm_init(uid, rank, world, ctypes.byref(handle)))
        self._h, self.rank, self.world = handle, rank, world
        self.stream = torch.cuda.Stream()        # the exchange runs here, next to the backward kernels

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError("sqd_comm: %s" % self._lib.sqd_last_error().decode())

    def all_reduce(self, t, op="avg", stream=None):
        """in place, on `stream` (default: the caller's current stream)"""
        assert t.is_cuda and t.is_contiguous()
        st = torch.cuda.current_stream() if stream is None else stream
        self._check(self._lib.sqd_comm_allreduce(self._h, t.data_ptr(), t.numel(), self._DT[t.dtype], self._OP[op], st.cuda_stream))

    def broadcast(self, t, root=0, stream=None):
        assert t.is_cuda and t.is_contiguous()
        st = torch.cuda.current_stream() if stream is None else stream
        self._check(self._lib.sqd_comm_broadcast(self._h, t.data_ptr(), t.numel(), self._DT[t.dtype], root, st.cuda_stream))

    def barrier(self):
        t = torch.zeros(1, device="cuda")
        self.all_reduce(t, "sum")
        torch.cuda.current_stream().synchronize()

    def rccl_version(self):
        """NCCL-style version code of the bound librccl as "major.minor.patch" (reporting only)"""
        v = ctypes.c_int(0)
        self._check(self._lib.sqd_comm_rccl_version(ctypes.byref(v)))
        v = v.value
        return "%d.%d.%d" % (v // 10000, (v // 100) % 100, v % 100) if v >= 10000 else str(v)

    def joined(self):
        """ranks RCCL itself counts in this communicator"""
        n = ctypes.c_int(0)
        self._check(self._lib.sqd_comm_joined(self._h, ctypes.byref(n)))
        return n.value

    def close(self):
        if self._h is not None:
            torch.cuda.synchronize()
            self._lib.sqd_comm_destroy(self._h)
            self._h = None


class GlooComm:
    """torch.distributed (gloo) behind the same five calls: the CPU tests of the bucket logic (world size 2, no GPU)."""
    device_averages = False
    stream = None

    def __init__(self, group=None):
        self.group, self.rank, self.world = group, dist.get_rank(group), dist.get_world_size(group)

    def all_reduce(self, t, op="avg", stream=None):
        dist.all_reduce(t, op={"sum": dist.ReduceOp.SUM, "avg": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX,
                               "min": dist.ReduceOp.MIN}[op], group=self.group)
        if op == "avg":
            t.div_(self.world)

    def broadcast(self, t, root=0, stream=None):
        dist.broadcast(t, root, group=self.group)

    def barrier(self):
        dist.barrier(group=self.group)

    def close(self):
        pass


def init_from_env(backend=None):
    """Join the job torchrun's environment describes.  Returns (rank, world, local_rank) and sets ddp.COMM.
    Control plane (rendezvous, shipping RCCL's unique id, CPU-side barriers): a gloo process group over 127.0.0.1 / the
    job's store.  Data plane on a GPU: RcclComm (the library's own communicator) — ProcessGroupNCCL is never created."""
    global COMM
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    force = os.environ.get("SQD_FORCE_DIST") == "1" and "MASTER_ADDR" in os.environ     # 1-rank RCCL runs (tests)
    if (world > 1 or force) and COMM is None:
        gpu = backend != "gloo" and torch.cuda.is_available()
        if gpu:
            torch.cuda.set_device(local)
        if not dist.is_initialized():
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        if gpu:
            def exchange(uid):
                box = [uid]
                dist.broadcast_object_list(box, src=0)
                return box[0]
            COMM = RcclComm(rank, world, exchange)
        else:
            COMM = GlooComm()
    return rank, world, local


def shutdown():
    global COMM
    if COMM is not None:
        COMM.close()
        COMM = None
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def _dense(p):
    """p occupies numel() contiguous elements in some dimension order (contiguous or channels-last)."""
    return p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))


class GradBucketReducer:
    def __init__(self, params, bucket_mb=32.0, comm=None):
        self.comm = COMM if comm is None else comm
        # exchange whenever a communicator exists (a 1-rank communicator still exercises the RCCL calls)
        self.active = self.comm is not None
        self.world = self.comm.world if self.active else 1
        self.params = [p for p in params if p.requires_grad]
        self.bucket_bytes = int(bucket_mb * (1 << 20))
        self.buckets = None          # built after the first backward (unused-parameter detection)
        self._hooks = []
        self._inflight = False        # collectives queued on the communicator's stream since the last finish()
        self.hooks_enabled = True     # False while a hipGraph of forward+backward is captured / replayed (see allreduce_all)
        self._bucket_of, self._by_ptr = {}, {}

    # -- start-up ---------------------------------------------------------------------------------
    def broadcast_parameters(self, modules):
        """Rank 0's parameters and buffers become everyone's (one flat broadcast per dtype)."""
        if not self.active:
            return
        tensors = []
        for m in modules:
            tensors += [p.data for p in m.parameters()] + [b.data for b in m.buffers()]
        # (in first-appearance order: a set's iteration order depends on the process's hash seed, and ranks that walk the
        #  dtypes in different orders issue mismatched broadcasts)
        for dtype in list(dict.fromkeys(t.dtype for t in tensors)):
            group = [t for t in tensors if t.dtype == dtype]
            flat = torch.cat([t.reshape(-1) for t in group])
            if dtype == torch.int64 and flat.is_cuda:        # (BatchNorm's num_batches_tracked) the ABI moves i32
                wire = flat.to(torch.int32)
                self.comm.broadcast(wire, 0)
                flat = wire.to(torch.int64)
            else:
                self.comm.broadcast(flat, 0)
            off = 0
            for t in group:
                n = t.numel()
                t.copy_(flat[off:off + n].view_as(t))
                off += n

    def _build(self):
        used = [p for p in self.params if p.grad is not None]
        self.buckets = []
        cur, cur_bytes = [], 0
        for p in reversed(used):
            cur.append(p)
            cur_bytes += p.numel() * 4
            if cur_bytes >= self.bucket_bytes:
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
        if cur:
            self.buckets.append(cur)
        self.flat, self._pending, self._bucket_of, self.views = [], [], {}, []
        for bi, plist in enumerate(self.buckets):
            flat = torch.zeros(sum(p.numel() for p in plist), dtype=plist[0].dtype, device=plist[0].device)
            off, views = 0, []
            for p in plist:
                n = p.numel()
                # the view carries the PARAMETER's strides (channels-last filters stay KRSC in the bucket): autograd then
                # accumulates in place and FusedAdam reads the bucket memory directly, no layout copy in between
                view = flat.as_strided(p.shape, p.stride(), off) if _dense(p) else flat[off:off + n].view_as(p)
                view.copy_(p.grad)
                p.grad = view
                views.append(view)
                self._bucket_of[p] = bi
                off += n
            self.flat.append(flat)
            self.views.append(views)
            self._pending.append(len(plist))
        self._by_ptr = {p.data_ptr(): p for p in used}
        for p in used:
            self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # -- the exchange -----------------------------------------------------------------------------
    def _exchange(self, flat):
        """Average one bucket across the ranks, asynchronously: on a GPU the collective goes onto the communicator's stream
        behind everything the current stream has queued (an event edge — inside a hipGraph capture: a graph branch);
        finish() joins it."""
        side = self.comm.stream
        if side is None:                                   # CPU (gloo): synchronous
            self.comm.all_reduce(flat, "avg")
            return
        side.wait_stream(torch.cuda.current_stream())
        self.comm.all_reduce(flat, "avg", stream=side)
        self._inflight = True

    def _join(self):
        if self._inflight:
            torch.cuda.current_stream().wait_stream(self.comm.stream)
            self._inflight = False

    # -- per step ---------------------------------------------------------------------------------
    def zero_grad(self):
        """Replaces optimizer.zero_grad(set_to_none=True): backward then produces fresh gradient tensors, which the hook
        moves into the buckets."""
        for p in self.params:
            p.grad = None
        if self.buckets is not None:
            for bi in range(len(self.buckets)):
                self._pending[bi] = len(self.buckets[bi])
        self._forget_deferred()

    @staticmethod
    def _forget_deferred():
        """the filters whose gradient the previous backward pass handed over directly: a pass that raised after Conv2d.backward had
        added its filters, or a loop that drives this reducer without nnkernels.begin_step(), would otherwise leave entries behind that
        make a later, genuinely accumulated gradient of the same filter go uncounted (the bucket would never complete)"""
        from . import nnkernels
        nnkernels.DEFERRED_FILTERS.clear()
        nnkernels.drop_pending_reduce()

    def _on_grad(self, p, announced=False):
        if not self.hooks_enabled:
            return
        if not announced:
            # (torch fires a post-accumulate hook even when the node received no gradient: a filter whose gradient Conv2d.backward
            #  handed over directly gets here right after that backward, before the launch carrying the sum of its pixel splits is
            #  enqueued — it counts when nnkernels announces it, not now)
            from . import nnkernels
            if p.data_ptr() in nnkernels.DEFERRED_FILTERS:
                return
        bi = self._bucket_of.get(p)
        if bi is None:
            return
        self._pending[bi] -= 1
        if self._pending[bi] == 0:
            plist, views = self.buckets[bi], self.views[bi]
            from . import nnkernels
            nnkernels.join_wgrad_stream()                               # the convolutions' weight gradients run on their own stream
            torch._foreach_copy_(views, [q.grad for q in plist])        # gather the bucket: one multi-tensor copy
            for q, v in zip(plist, views):
                q.grad = v                                              # the optimiser reads the (averaged) bucket memory
            if self.active:
                self._exchange(self.flat[bi])

    def on_deferred_grad(self, w):
        """nnkernels.DEFERRED_GRAD_HOOK: a filter whose gradient was handed to it directly (its split sum rides on a BatchNorm-backward
        launch that has just been enqueued) arrives like any accumulated gradient."""
        p = w if w in self._bucket_of else self._by_ptr.get(w.data_ptr())
        if p is not None:
            self._on_grad(p, announced=True)

    def detach_grad_views(self):
        """Graph mode: hand back {parameter: its bucket view} and clear p.grad, so that a captured backward produces fresh
        gradient tensors which the caller copies into the views (and re-attaches the views afterwards)."""
        views = {}
        for plist in self.buckets:
            for p in plist:
                views[p] = p.grad
                p.grad = None
        return views

    def reattach_grad_views(self):
        """every bucketed parameter's .grad is its bucket view again (after a capture attempt that left them half-way)"""
        if self.buckets is None:
            return
        for bi, plist in enumerate(self.buckets):
            for p, v in zip(plist, self.views[bi]):
                p.grad = v
            self._pending[bi] = len(plist)
        self._inflight = False
        self._forget_deferred()

    def bucket_bytes_list(self):
        return [int(f.numel() * f.element_size()) for f in getattr(self, "flat", [])]

    def allreduce_all(self):
        """Exchange every bucket now (--sqd_graph_ddp post: forward+backward were replayed as one hipGraph, which leaves no
        Python hook to overlap with; the 4-5 collectives are issued back to back on the communicator's stream and joined)."""
        if not self.active or self.buckets is None:
            return
        for flat in self.flat:
            self._exchange(flat)
        self._join()

    def finish(self):
        """Call after loss.backward(): the current stream waits for the in-flight buckets."""
        if self.buckets is None:
            # first step: no buckets yet — reduce whatever gradients exist, then build the buckets
            if self.active:
                # the buckets are built from the parameters that received a gradient in THIS backward pass: every rank must have seen the
                # same set, or the ranks would bucket — and from then on exchange — different tensors.  One small max / min exchange.
                dev = self.params[0].device
                mask = torch.tensor([1.0 if p.grad is not None else 0.0 for p in self.params], device=dev, dtype=torch.float32)
                lo, hi = mask.clone(), mask.clone()
                self.comm.all_reduce(lo, "min")
                self.comm.all_reduce(hi, "max")
                if not torch.equal(lo.cpu(), hi.cpu()):
                    differ = [i for i, (a, b) in enumerate(zip(lo.cpu().tolist(), hi.cpu().tolist())) if a != b]
                    raise RuntimeError("sqd.ddp: the ranks disagree on which parameters received a gradient in the first backward pass "
                                       "(parameter indices %s...): their gradient buckets would not line up" % differ[:8])
                grads = [p.grad for p in self.params if p.grad is not None]
                flat = torch.cat([g.reshape(-1) for g in grads])
                self.comm.all_reduce(flat, "avg")
                off = 0
                for g in grads:
                    n = g.numel()
                    g.copy_(flat[off:off + n].view_as(g))
                    off += n
            self._build()
            return
        self._join()
        # every bucket either completed (and was exchanged) or saw no gradient at all this pass: anything in between means an arrival was
        # lost (a stale DEFERRED_FILTERS entry, a hook that did not fire) and the ranks would silently diverge
        if self.hooks_enabled:
            stuck = [bi for bi, plist in enumerate(self.buckets) if 0 < self._pending[bi] < len(plist)]
            # a bucket whose missing arrivals are exactly its parameters WITHOUT a gradient in this pass (a frozen or conditionally unused
            # branch that shares the bucket with used parameters) is exchanged with zeros in their place — torch DDP's treatment of unused
            # parameters; as there, every rank must leave the same parameters unused in a step.  Missing arrivals of parameters that DO hold
            # a gradient are lost ones: no exchange may silently skip them.
            lost = [bi for bi in stuck if self._pending[bi] != sum(1 for q in self.buckets[bi] if q.grad is None)]
            if lost:
                raise RuntimeError("sqd.ddp: gradient bucket(s) %s received only part of their gradients in this backward pass "
                                   "(%s of %s arrivals missing, parameters without a gradient: %s): no all-reduce ran for them" %
                                   (lost, [self._pending[bi] for bi in lost], [len(self.buckets[bi]) for bi in lost],
                                    [sum(1 for q in self.buckets[bi] if q.grad is None) for bi in lost]))
            for bi in stuck:
                from . import nnkernels
                nnkernels.join_wgrad_stream()
                for q, v in zip(self.buckets[bi], self.views[bi]):
                    if q.grad is None:
                        v.zero_()
                    elif q.grad is not v:
                        v.copy_(q.grad)
                    q.grad = v
                self._pending[bi] = 0
                if self.active:
                    self._exchange(self.flat[bi])
            if stuck:
                self._join()
