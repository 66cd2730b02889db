"""FusedAdam — torch.optim.Adam whose step() is ONE libsqd kernel launch per parameter group.

Keeps torch.optim.Adam's state layout (state[p] = {"step", "exp_avg", "exp_avg_sq"}, param_groups),
so `optimizer.state_dict()` / `load_state_dict()` and the reference's adam.pth checkpoints
(trainer.py:659-660, 682-687) stay interchangeable, and StepLR keeps working through param_groups."""
import ctypes

import torch

from . import lib as _l
from .ops import _stream


class FusedAdam(torch.optim.Adam):
    HYPER_RING = 8

    DECOUPLED_DECAY = False       # FusedAdamW: param_groups carry a (decoupled) weight decay

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False):
        if (weight_decay != 0 and not self.DECOUPLED_DECAY) or amsgrad:
            raise NotImplementedError("FusedAdam implements the reference's configuration: no weight decay, no amsgrad")
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, foreach=False, fused=False)
        self._tables = {}
        self._ring = 0
        self._graph_hyper = None      # set by begin_capture(): {group index: (ring of (pinned host [2], event), device [2])}
        self._hyper_slot = {}

    def load_state_dict(self, state_dict):
        """torch.optim.Adam.load_state_dict, then the moments are laid out like their parameters: the kernel pairs p,
        exp_avg, exp_avg_sq and grad by flat memory offset, and a checkpoint written from NCHW-contiguous parameters (the
        reference's adam.pth, trainer.py:659-660) carries contiguous moments while the convolution filters here are
        channels-last."""
        super().load_state_dict(state_dict)
        for group in self.param_groups:
            for p in group["params"]:
                st = self.state.get(p)
                if not st:
                    continue
                for k in ("exp_avg", "exp_avg_sq"):
                    if k in st and (st[k].stride() != p.stride() or st[k].device != p.device or st[k].dtype != p.dtype):
                        st[k] = torch.empty_like(p, memory_format=torch.preserve_format).copy_(st[k])
                if "step" in st and torch.is_tensor(st["step"]) and st["step"].is_cuda:
                    st["step"] = st["step"].detach().cpu()
        self._tables = {}

    def _build(self, gi, plist):
        L = _l.lib()
        chunk = L.sqd_adam_chunk_elems()
        dev = plist[0].device
        recs, chunks = [], []
        for ti, p in enumerate(plist):
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            n = p.numel()
            recs.append([p.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), n])
            chunks += [[ti, c] for c in range((n + chunk - 1) // chunk)]
        shared = self.state[plist[0]]["step"]
        for p in plist:                                # one counter tensor for the whole group
            self.state[p]["step"] = shared
        ghost = [torch.empty(len(plist), dtype=torch.int64).pin_memory() for _ in range(4)]
        tab = {"params": plist, "keys": [(p.data_ptr(), self.state[p]["exp_avg"].data_ptr()) for p in plist], "step": shared,
               "pstrides": [p.stride() for p in plist], "ptr_scratch": [0] * len(plist), "ghost_np": [h.numpy() for h in ghost],
               "recs": torch.tensor(recs, dtype=torch.int64).to(dev), "chunks": torch.tensor(chunks, dtype=torch.int32).to(dev),
               "nchunks": len(chunks), "gdev": torch.empty(len(plist), dtype=torch.int64, device=dev),
               "ghost": ghost}
        self._tables[gi] = tab
        return tab

    # ---- hipGraph support: a captured step() launches sqd_adam_step_dev, whose step-dependent scalars are two device
    # floats per group; refresh_hyper() (outside the graph, before every replay) advances the step counts and rewrites
    # them, so StepLR and the bias corrections keep working across replays.
    def begin_capture(self):
        """Call before torch.cuda.graph(...): allocates what a capturing stream may not (pinned host buffers)."""
        self._graph_hyper = {}
        self._graph_ptrs = [torch.empty(max(1, len(g["params"])), dtype=torch.int64).pin_memory() for g in self.param_groups]

    def refresh_hyper(self):
        L = _l.lib()
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group["params"] if p.grad is not None or "step" in self.state.get(p, ())]
            if not plist:
                continue
            if gi not in self._graph_hyper:
                dev = plist[0].device
                # A ring of pinned staging buffers, each guarded by an event: replays are queued faster than they execute
                # (~1 ms of host time per ~15 ms step), so the copy of step N may still be pending when the host prepares
                # step N+1 — a single buffer would hand later scalars to earlier replays.  A slot is rewritten only after the
                # copy that last read it has completed (back-pressure after HYPER_RING steps of run-ahead).
                ring = [(torch.zeros(2, dtype=torch.float32).pin_memory(), torch.cuda.Event()) for _ in range(self.HYPER_RING)]
                self._graph_hyper[gi] = (ring, torch.zeros(2, dtype=torch.float32, device=dev))
                self._hyper_slot[gi] = 0
            ring, devt = self._graph_hyper[gi]
            slot = self._hyper_slot[gi]
            self._hyper_slot[gi] = (slot + 1) % self.HYPER_RING
            host, done = ring[slot]
            done.synchronize()                 # (a never-recorded event returns at once)
            bumped = set()
            for p in plist:
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if id(st["step"]) not in bumped:           # the group's parameters may share one counter tensor
                    bumped.add(id(st["step"]))
                    st["step"] += 1
            step = int(self.state[plist[0]]["step"].item())
            b1, b2 = group["betas"]
            _l.check(L.sqd_adam_hyper(float(group["lr"]), float(b1), float(b2), step, ctypes.c_void_p(host.data_ptr())), "adam_hyper")
            devt.copy_(host, non_blocking=True)
            done.record()

    def _step_captured(self):
        L = _l.lib()
        todo = []
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group["params"] if p.grad is not None]
            if plist and self._tables.get(gi) is not None:
                todo.append((self._tables[gi], plist))
        fused = self._amax_tables(todo) if todo else False
        for gi, group in enumerate(self.param_groups):
            plist = [p for p in group["params"] if p.grad is not None]
            if not plist:
                continue
            tab = self._tables.get(gi)
            keys = [(p.data_ptr(), self.state[p]["exp_avg"].data_ptr() if "exp_avg" in self.state[p] else 0) for p in plist]
            # the gradient buffers of a captured backward are fixed: their addresses are written once, from a pinned
            # buffer that is never touched again (the copy is part of the graph)
            if tab is None or tab["keys"] != keys:
                raise RuntimeError("FusedAdam (graph capture): run at least one eager step first (the tensor tables are built there)")
            host = self._graph_ptrs[gi][:len(plist)]
            for i, p in enumerate(plist):
                if p.grad.stride() != p.stride():            # a captured copy into the parameter's layout
                    p.grad = p.grad.contiguous(memory_format=torch.channels_last if p.dim() == 4 and
                                               p.is_contiguous(memory_format=torch.channels_last) and
                                               not p.is_contiguous() else torch.contiguous_format)
                host[i] = p.grad.data_ptr()
            tab["graph_host"] = host
            tab["gdev"].copy_(host, non_blocking=True)
            b1, b2 = group["betas"]
            _l.check(L.sqd_adam_step_dev_amax(ctypes.c_void_p(tab["recs"].data_ptr()), ctypes.c_void_p(tab["gdev"].data_ptr()),
                                              ctypes.c_void_p(tab["chunks"].data_ptr()), tab["nchunks"],
                                              ctypes.c_void_p(self._graph_hyper[gi][1].data_ptr()), float(b1), float(b2),
                                              float(group["eps"]), ctypes.c_void_p(tab["amax_dev"].data_ptr()) if fused else None,
                                              _stream()), "adam_step_dev")
        if fused:
            from . import nnkernels
            nnkernels.wam_records_written([p for _, plist in todo for p in plist])

    def _prepare_group(self, gi, group):
        """table of the group's parameters with fresh gradient addresses, and the advanced step count -> (tab, step, plist) or None"""
        plist = [p for p in group["params"] if p.grad is not None]
        if not plist:
            return None
        # (host time matters: an eager step is launch-bound.  The per-parameter checks run when the table is (re)built;
        # per step there is one pass over the gradients and one vectorised write of their addresses.)
        tab = self._tables.get(gi)
        keys = [(p.data_ptr(), self.state[p]["exp_avg"].data_ptr() if "exp_avg" in self.state[p] else 0) for p in plist]
        if tab is None or tab["keys"] != keys:
            for p in plist:
                dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
                if not (p.is_cuda and p.dtype == torch.float32 and dense):
                    raise RuntimeError("FusedAdam: parameters must be dense fp32 device tensors")
            tab = self._build(gi, plist)
        # gradient tensors are re-allocated by every backward: refresh their addresses (one 8*n byte async copy)
        host = tab["ghost"][self._ring % 4]
        self._ring += 1
        ptrs, strides = tab["ptr_scratch"], tab["pstrides"]
        for i, p in enumerate(plist):
            g = p.grad
            if g.stride() != strides[i]:
                g = p.grad = g.contiguous(memory_format=torch.channels_last if p.dim() == 4 and
                                          p.is_contiguous(memory_format=torch.channels_last) and
                                          not p.is_contiguous() else torch.contiguous_format)
            ptrs[i] = g.data_ptr()
        tab["ghost_np"][(self._ring - 1) % 4][:] = ptrs
        tab["gdev"].copy_(host, non_blocking=True)
        # one step counter per group, shared by the states of its parameters (torch.optim.Adam keeps one per parameter;
        # state_dict() still lists it under every parameter)
        st0 = tab["step"]
        st0 += 1
        return tab, int(st0.item()), plist

    def _amax_tables(self, tabs_plists):
        """[(tab, plist)] of one step -> True when the launches of this step leave the convolution filters' max |w| records behind (every live
        registered filter is updated by them: nnkernels.wam_records_for); fills tab["amax_dev"] (device array of record addresses, 0 = none)
        and clears the records on the stream."""
        from . import nnkernels
        info = nnkernels.wam_records_for([p for _, plist in tabs_plists for p in plist])
        if info is None:
            for tab, _ in tabs_plists:
                tab["amax_on"] = False
            return False
        gen, addr = info
        for tab, plist in tabs_plists:
            if tab.get("amax_gen") != gen:
                if torch.cuda.is_current_stream_capturing():
                    # (a table that changed between the warm-up steps and the capture: leave the records to nnkernels.begin_step)
                    for t, _ in tabs_plists:
                        t["amax_on"] = False
                    return False
                tab["amax_dev"] = torch.tensor([addr.get(p.data_ptr(), 0) for p in plist], dtype=torch.int64).to(plist[0].device)
                tab["amax_gen"] = gen
            tab["amax_on"] = True
        nnkernels.wam_clear_records()
        return True

    def _launch_groups(self, ready):
        L = _l.lib()
        fused = self._amax_tables([(tab, plist) for _, _, tab, _, plist in ready]) if ready else False
        for gi, group, tab, step, plist in ready:
            b1, b2 = group["betas"]
            _l.check(L.sqd_adam_step_amax(ctypes.c_void_p(tab["recs"].data_ptr()), ctypes.c_void_p(tab["gdev"].data_ptr()),
                                          ctypes.c_void_p(tab["chunks"].data_ptr()), tab["nchunks"], float(group["lr"]), float(b1),
                                          float(b2), float(group["eps"]), step,
                                          ctypes.c_void_p(tab["amax_dev"].data_ptr()) if fused else None, _stream()), "adam_step")
        if fused:
            from . import nnkernels
            nnkernels.wam_records_written([p for _, _, _, _, plist in ready for p in plist])

    @torch.no_grad()
    def step(self, closure=None):
        # the kernels below write the parameters through raw pointers: the filters' max |.| table of the two-term fp16 convolution
        # plans is stale until the next nnkernels.begin_step() recomputes it (one launch)
        from . import nnkernels
        nnkernels.weights_changed()
        if self._graph_hyper is not None and torch.cuda.is_current_stream_capturing():
            self._step_captured()
            return None
        loss = closure() if closure is not None else None
        ready = []
        for gi, group in enumerate(self.param_groups):
            pr = self._prepare_group(gi, group)
            if pr is not None:
                ready.append((gi, group) + pr)
        self._launch_groups(ready)
        return loss


class FusedAdamW(FusedAdam):
    DECOUPLED_DECAY = True

    """torch.optim.AdamW (decoupled weight decay) with an optional global gradient-norm clip folded into the step — the optimiser
    side of the reference's finetune loop (finetune/train_ft_SQLdepth.py:184 `optim.AdamW(params, weight_decay=args.wd, lr=args.lr)`,
    :281 `nn.utils.clip_grad_norm_(model.parameters(), 0.1)`): one launch pair computes min(1, max_norm / (||g|| + 1e-6)) over the
    gradients of ALL groups into a device scalar, and the AdamW kernel of every group multiplies its gradients by it on the fly — the
    gradients are not rewritten (p.grad stays unclipped).  param_groups / state_dict() are torch.optim.Adam's, so
    torch.optim.lr_scheduler.OneCycleLR (cycle_momentum moves betas[0]) drives it like the reference's optimiser.
    `clip_info` holds the last (coefficient, norm) as a device tensor."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=None):
        # the decay lives where torch.optim.AdamW keeps it — in every param_group (default from the constructor, per-group values from the
        # caller's dicts) — so that state_dict() reports it and a reference AdamW checkpoint's value takes effect when loaded
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.max_grad_norm = max_grad_norm
        self.clip_info = None

    def begin_capture(self):
        raise NotImplementedError("FusedAdamW runs eagerly (the finetune loop is not captured)")

    def _launch_groups(self, ready):
        if not ready:
            return
        L = _l.lib()
        gscale = None
        if self.max_grad_norm is not None:
            dev = ready[0][4][0].device
            part = torch.empty(sum(r[2]["nchunks"] for r in ready), device=dev, dtype=torch.float32)
            # per-chunk sums of squares of every group into one partial array, then one fixed-order total
            off = 0
            for gi, group, tab, step, plist in ready:
                _l.check(L.sqd_grad_sumsq(ctypes.c_void_p(tab["recs"].data_ptr()), ctypes.c_void_p(tab["gdev"].data_ptr()),
                                          ctypes.c_void_p(tab["chunks"].data_ptr()), tab["nchunks"],
                                          ctypes.c_void_p(part.data_ptr() + 4 * off), _stream()), "grad_sumsq")
                off += tab["nchunks"]
            self.clip_info = torch.empty(2, device=dev, dtype=torch.float32)
            _l.check(L.sqd_clip_coef(ctypes.c_void_p(part.data_ptr()), off, float(self.max_grad_norm), ctypes.c_void_p(self.clip_info.data_ptr()),
                                     _stream()), "clip_coef")
            gscale = ctypes.c_void_p(self.clip_info.data_ptr())
        for gi, group, tab, step, plist in ready:
            b1, b2 = group["betas"]
            _l.check(L.sqd_adamw_step(ctypes.c_void_p(tab["recs"].data_ptr()), ctypes.c_void_p(tab["gdev"].data_ptr()),
                                      ctypes.c_void_p(tab["chunks"].data_ptr()), tab["nchunks"], float(group["lr"]), float(b1), float(b2),
                                      float(group["eps"]), float(group["weight_decay"]), step, gscale, _stream()), "adamw_step")
