"""Autograd nodes of the network operators served by libsqd.so (channels-last activations).

Each Function takes/returns tensors whose logical shape is NCHW and whose memory is NHWC
(torch.channels_last); the kernels see them as row-major [M = N*H*W, C] matrices."""
import ctypes

import torch

from . import lib as _l
from .ops import _ptr, _stream

ACT = {None: 0, "relu": 1, "leaky_relu": 2, "swish": 3}


_DEBUG_COPIES = False          # tools/ set it: report activations that reach a kernel in the wrong memory format


def _cl(x):
    """NHWC-contiguous view/copy of a 4-D tensor."""
    if _DEBUG_COPIES and not x.is_contiguous(memory_format=torch.channels_last):
        import traceback
        fr = traceback.extract_stack(limit=4)[:-1]
        print("sqd: layout copy", tuple(x.shape), tuple(x.stride()), " <- ".join("%s:%d" % (f.name, f.lineno) for f in reversed(fr)), flush=True)
    return x.contiguous(memory_format=torch.channels_last)


def _require(t, what):
    if not (t.is_cuda and t.dtype == torch.float32):
        raise RuntimeError("sqd: %s must be an fp32 tensor on the MI355X device (no CPU fallback)" % what)


def bn_supported(C):
    return C % 4 == 0 and C >= 4


# ---------------------------------------------------------------------------------------------------
# Operand scales of the two-term fp16 convolution plans (csrc/conv.hip "f16x2", include/sqd.h section 10b).  Such a plan needs max |.| of
# both operand tensors as device scalars.  Producers record it on the way (the `amax` outputs of the BatchNorm / convolution-epilogue /
# up-sampling / frame-staging kernels): a tensor carries `_sqd_amax = (slot, epoch, tensor version)`, slot = one 4 KB record (64 words in 64 cache lines) of a pool that begin_step() clears
# at the start of every step (inside the captured graph).  A tensor without a valid tag (a gradient summed by autograd, the output of an
# operator that does not record it) gets a standalone pass (sqd_amax) — counted per call site in AMAX_STATS, so that a missing producer
# shows up as a number instead of as a slow step.  Filters: one persistent table, refreshed by ONE multi-tensor launch per step
# (sqd_amax_multi) — they change once per step, in the optimiser.
# ---------------------------------------------------------------------------------------------------
AMAX_ON = True                 # producers record max |output| (nnops.configure: off under --sqd_no_f16x2 / --sqd_bf16)
AMAX_STATS = {"fused": 0, "standalone": 0, "sites": {}}
AMAX_REC = 1024                # SQD_AMAX_RECORD_FLOATS: a record is 64 words in 64 cache lines (the recording kernels spread their atomics over them)
_AM = {"buf": None, "n": 0, "epoch": 0, "size": 1024}
_WAM = {"buf": None, "index": {}, "refs": [], "dirty": True, "recs": None, "chunks": None, "nchunks": 0, "fresh": set(), "gen": 0}


def amax_enable(on):
    global AMAX_ON
    AMAX_ON = bool(on)


def _amax_new(device):
    """a cleared record of the per-step pool (AMAX_REC floats)"""
    if _AM["buf"] is None or _AM["buf"].device != device:
        _AM["buf"] = torch.zeros(_AM["size"] * AMAX_REC, device=device, dtype=torch.float32)
        _AM["n"] = 0
        _AM["epoch"] += 1
    if _AM["n"] >= _AM["size"]:
        # the pool is used up inside one step (a deeper model than any shipped one — they use 500-750 records — or a loop that never calls
        # begin_step): a FRESH pool, never a clear of the old one — records handed out earlier in this step may still be read by a pending
        # data gradient or a deferred side-stream weight gradient.  The old pool stays alive until the next begin_step.
        _AM.setdefault("retired", []).append(_AM["buf"])
        _AM["buf"] = torch.zeros(_AM["size"] * AMAX_REC, device=device, dtype=torch.float32)
        _AM["n"] = 0
    i = _AM["n"]
    _AM["n"] = i + 1
    return _AM["buf"][i * AMAX_REC:(i + 1) * AMAX_REC]


def amax_value(rec):
    """the number a record holds (host read-back: tests and tools)"""
    return float(rec.view(-1, 16)[:, 0].max())


def _amax_tag(t, slot):
    # the tag names the tensor's CONTENT: it carries the tensor's version counter, because autograd's InputBuffer accumulates a second
    # gradient IN PLACE into a tensor whose last reference it holds — the Python attribute survives that sum, the maximum does not
    if t is not None and slot is not None:
        t._sqd_amax = (slot, _AM["epoch"], t._version)
    return t


def _amax_get(t):
    tag = getattr(t, "_sqd_amax", None)
    return tag[0] if tag is not None and tag[1] == _AM["epoch"] and tag[2] == t._version else None


def _colsum_tag(t, cs):
    """column sums of a gradient its producer leaves for the next Conv2d.backward's bias gradient — valid for this content only"""
    t._sqd_colsum = (cs, t._version)


def _colsum_get(t):
    tag = getattr(t, "_sqd_colsum", None)
    return tag[0] if tag is not None and tag[1] == t._version else None


def _amax_out(device):
    """slot for a producer's `amax` output (None when recording is off)"""
    if not AMAX_ON:
        return None
    AMAX_STATS["fused"] += 1
    return _amax_new(device)


def amax_of(t, site):
    """device scalar holding the bits of max |t| — the tensor's tag, or a standalone pass"""
    a = _amax_get(t)
    if a is None:
        a = _amax_new(t.device)
        _l.check(_l.lib().sqd_amax(_ptr(t), t.numel(), _ptr(a), _stream()), "amax")
        AMAX_STATS["standalone"] += 1
        AMAX_STATS["sites"][site] = AMAX_STATS["sites"].get(site, 0) + 1
        _amax_tag(t, a)
    return a


def _wam_slot(w):
    """slot of a leaf filter in the persistent table (registered on first use)"""
    import weakref
    dev = w.device
    if _WAM["buf"] is None or _WAM["buf"].device != dev:
        _WAM.update(buf=torch.zeros(1024 * AMAX_REC, device=dev, dtype=torch.float32), index={}, refs=[], dirty=True, fresh=set())
    key = w.data_ptr()
    ent = _WAM["index"].get(key)
    if ent is not None and (ent[1]() is None or ent[2] != w.numel()):
        ent = None                                   # another tensor lives at that address now
    if ent is None:
        live = {k: e for k, e in _WAM["index"].items() if e[1]() is not None and k != key}
        used = {e[0] for e in live.values()}
        idx = next((i for i in range(1024) if i not in used), None)
        if idx is None:
            raise RuntimeError("sqd.nnkernels: more than 1024 live convolution filters in the operand-scale table (register parameters, not "
                               "per-call derived tensors: a derived filter should carry _sqd_w_src)")
        ent = (idx, weakref.ref(w), w.numel(), [-1])
        live[key] = ent
        _WAM["index"] = live
        _WAM["dirty"] = True
        _WAM["gen"] += 1                                # (optimisers cache record addresses per generation of the table)
    return ent


def _weight_source(w):
    """the leaf parameter a filter tensor was rearranged from (`_sqd_w_src`: regrouped stem filters, Linear weights viewed as 1x1 filters —
    the same values, so the same max |.|), the tensor itself when it is a leaf, else None"""
    src = getattr(w, "_sqd_w_src", None)
    if src is not None:
        return src
    return w if w.is_leaf else None


def amax_of_weight(w):
    """filters: the persistent table for a leaf parameter (or the leaf a rearranged filter came from), a standalone pass otherwise"""
    src = _weight_source(w)
    if src is None:
        return amax_of(w, "filter (not a leaf)")
    w = src
    ent = _wam_slot(w)
    slot = _WAM["buf"][ent[0] * AMAX_REC:(ent[0] + 1) * AMAX_REC]
    # valid when this step's multi-tensor launch covered it and nothing wrote the parameter through torch since
    if ent[3][0] != w._version or ent[0] not in _WAM["fresh"]:
        _l.check(_l.lib().sqd_amax(_ptr(w), w.numel(), _ptr(slot), _stream()), "amax (filter)")
        ent[3][0] = w._version
        _WAM["fresh"].add(ent[0])
        AMAX_STATS["standalone"] += 1
        AMAX_STATS["sites"]["filter (first use)"] = AMAX_STATS["sites"].get("filter (first use)", 0) + 1
    return slot


def weights_changed():
    """an optimiser wrote the parameters through raw pointers: the filter table is stale until the next begin_step()"""
    _WAM["fresh"] = set()


def filter_records_stale():
    """has anything written a registered filter through torch (load_state_dict, an initialiser) since its record was taken — or was it never taken?"""
    for e in _WAM["index"].values():
        w = e[1]()
        if w is not None and (e[0] not in _WAM["fresh"] or e[3][0] != w._version):
            return True
    return False


def refresh_filter_records_if_stale():
    """Before the replay of a captured step whose graph relies on the previous step's optimiser launch for the filter records (no pass of its
    own at its start): one eager sqd_amax_multi when something else has written the weights in between."""
    if AMAX_ON and _WAM["buf"] is not None and filter_records_stale():
        _wam_refresh()


FUSE_ADAM_AMAX = True          # the optimiser's kernel leaves the filters' max |w| records behind (tools may switch it off: tools/ab_bench.py)


def wam_records_for(params):
    """For an optimiser about to update `params` through its own kernel: (generation, {param data_ptr: record address}) if that kernel can
    leave the filter table's records behind — every live registered filter is among `params` (a filter the step does not touch would lose its
    record to the clearing) — else None.  The caller clears the records (wam_clear_records) on the stream before its launches and reports
    wam_records_written(params) after them."""
    if not (AMAX_ON and FUSE_ADAM_AMAX) or _WAM["buf"] is None:
        return None
    live = {k: e for k, e in _WAM["index"].items() if e[1]() is not None}
    if not live:
        return None
    ptrs = {p.data_ptr(): p for p in params}
    for k, e in live.items():
        p = ptrs.get(k)
        if p is None or p.numel() != e[2] or p is not e[1]():
            return None
    base = _WAM["buf"].data_ptr()
    return _WAM["gen"], {k: base + e[0] * AMAX_REC * 4 for k, e in live.items()}


def wam_clear_records():
    """the whole table of filter records, one fill on the stream (slots without a live filter hold nothing anyone reads)"""
    n = max(e[0] for e in _WAM["index"].values()) + 1
    _WAM["buf"][:n * AMAX_REC].zero_()


def wam_records_written(params):
    """the optimiser's launches have left max |w| of every live filter in its record: valid until something else writes the parameters"""
    fresh = set()
    for e in _WAM["index"].values():
        w = e[1]()
        if w is not None:
            e[3][0] = w._version
            fresh.add(e[0])
    _WAM["fresh"] = fresh


def _wam_refresh():
    """one launch: max |w| of every registered filter (begin_step)"""
    live = [(k, e) for k, e in _WAM["index"].items() if e[1]() is not None]
    if not live:
        return
    if _WAM["dirty"] or _WAM["recs"] is None:
        chunk = _l.lib().sqd_adam_chunk_elems()
        live.sort(key=lambda ke: ke[1][0])
        n_slots = max(e[0] for _, e in live) + 1
        recs = [[0, 0, 0, 0] for _ in range(n_slots)]            # table index == slot index (unused slots: empty tensors)
        chunks = []
        for k, e in live:
            recs[e[0]] = [k, 0, 0, e[2]]
            chunks += [[e[0], c] for c in range((e[2] + chunk - 1) // chunk)]
        dev = _WAM["buf"].device
        _WAM.update(recs=torch.tensor(recs, dtype=torch.int64).to(dev), chunks=torch.tensor(chunks, dtype=torch.int32).to(dev),
                    nchunks=len(chunks), nslots=n_slots, index=dict(live), dirty=False)
    _l.check(_l.lib().sqd_amax_multi(_ptr(_WAM["recs"]), _ptr(_WAM["chunks"]), _WAM["nchunks"], _WAM["nslots"], _ptr(_WAM["buf"]), _stream()),
             "amax_multi")
    _WAM["fresh"] = {e[0] for _, e in live}
    for _, e in live:
        w = e[1]()
        if w is not None:
            e[3][0] = w._version


class BatchNormAct(torch.autograd.Function):
    """y = act(BatchNorm2d(x) [+ residual]).  Training: batch statistics + running-stat update
    (in place on the BatchNorm2d buffers); eval: running statistics."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, residual, training, momentum, eps, act, pre_part=None, pre_rows=0, shared=None,
                pool_part=None):
        _require(x, "BatchNormAct input")
        x_in = x
        x = _cl(x)
        N, C, H, W = x.shape
        M = N * H * W
        res = _cl(residual) if residual is not None else None
        y = torch.empty_like(x, memory_format=torch.channels_last)
        L = _l.lib()
        code = ACT[act]
        if training:
            if pre_part is not None and pre_rows > 0:       # statistics partials from the producing convolution's epilogue
                part = pre_part
            else:
                pre_rows = 0
                part = torch.empty(L.sqd_bn_nblk(M, C) * C * 2, device=x.device, dtype=torch.float32)
            mean = torch.empty(C, device=x.device, dtype=torch.float32)
            rstd = torch.empty(C, device=x.device, dtype=torch.float32)
            # sign bits of the pre-activation (1 byte per 4 elements): the backward reads them instead of y
            mask = torch.empty(M * C // 4, device=x.device, dtype=torch.uint8) if code in (1, 2) else None    # (swish: recomputed from x)
            # pool_part [B, chunks, C] (batch_norm_act(pool=True)): the element-wise pass also takes the per-image channel sums of y for the
            # squeeze-and-excite gate that follows
            ay = _amax_out(x.device) if pool_part is None else None
            _l.check(L.sqd_bn_train_fwd_amax(_ptr(x), _ptr(res), _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var),
                                             _ptr(y), _ptr(mask), _ptr(mean), _ptr(rstd), _ptr(part), pre_rows, M, C, float(eps),
                                             float(momentum), code, _ptr(pool_part), N, _ptr(ay), _stream()), "bn_train_fwd")
            _amax_tag(y, ay)
            ctx.save_for_backward(x, mask, gamma, mean, rstd, beta if code == 3 else None)
            ctx.has_res, ctx.code = residual is not None, code
            # what the convolution that consumes y needs to deliver this node's backward statistics from its data-gradient epilogue
            # (sqd_conv_dgrad_bn): filled here, read by Conv2d.backward, answered through "dx" / "part" / "rows" (see batch_norm_act)
            ctx.shared = shared if code != 3 else None
            if ctx.shared is not None:
                shared.update(x=x, mask=mask, mean=mean, rstd=rstd, code=code, dx=None, part=None, rows=0)
            # the residual branch is itself the output of a training-mode BatchNorm without activation (a bottleneck's down-sample branch):
            # this node's backward can take that node's two sums on the way (dres is its whole gradient if this node is its only consumer —
            # its backward checks that the tensor that arrives is the dres written here)
            # x is the output of a convolution with a bias whose weight-gradient plan leaves no bias partials behind: this node's backward writes
            # dx, whose column sums ARE that bias gradient — it takes them on the way (Conv2d.backward finds them tagged on dx: _colsum_get)
            bg = getattr(x_in, "_sqd_bias_geom", None) if FUSE_BN_SIDE_SUMS["bias"] and x is x_in else None
            ctx.want_dxsum = bg is not None and _wgrad_key(bg) in _WGRAD_SETTLED and (L.sqd_conv_wgrad_effective_impl(*bg) & 15) not in (1, 4)
            rs = getattr(residual, "_sqd_bn_src", None) if FUSE_BN_BWD_STATS and FUSE_BN_SIDE_SUMS["res"] else None
            ctx.res_src = rs if rs is not None and rs.get("code") == 0 and rs["x"].shape == x.shape else None
        else:
            _l.check(L.sqd_bn_eval_fwd(_ptr(x), _ptr(res), _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var),
                                       _ptr(y), M, C, float(eps), code, _stream()), "bn_eval_fwd")
            ctx.save_for_backward()
        ctx.training = training
        return y

    @staticmethod
    def backward(ctx, dy):
        if not ctx.training:
            raise NotImplementedError("sqd: BatchNormAct backward is implemented for training mode only")
        x, mask, gamma, mean, rstd, beta = ctx.saved_tensors
        # the gradient that arrives is the very tensor a data-gradient epilogue wrote together with this node's two sums: skip the
        # reduction pass (a gradient summed from several consumers is another tensor: the ordinary path)
        sh = getattr(ctx, "shared", None)
        pre_part, pre_rows = (sh["part"], sh["rows"]) if sh is not None and sh.get("dx") is dy and sh.get("rows", 0) > 0 else (None, 0)
        if sh is not None:
            sh.update(dx=None, part=None, rows=0)
        dy = _cl(dy)
        N, C, H, W = x.shape
        M = N * H * W
        L = _l.lib()
        dx = torch.empty_like(x, memory_format=torch.channels_last)
        dres = torch.empty_like(x, memory_format=torch.channels_last) if ctx.has_res else None
        dgamma = torch.empty(C, device=x.device, dtype=torch.float32)
        dbeta = torch.empty(C, device=x.device, dtype=torch.float32)
        part = pre_part if pre_rows > 0 else torch.empty(L.sqd_bn_nblk(M, C) * C * 2, device=x.device, dtype=torch.float32)
        # a weight gradient whose pixel splits are not summed yet (Conv2d.backward under DEFER_WGRAD_REDUCE): the sum rides along as extra
        # workgroups of this node's finalize launch
        pend = _take_pending_reduce()
        rp, ro, rn, rs = pend[:4] if pend is not None else (None, None, 0, 0)
        adx = _amax_out(x.device)
        adr = _amax_out(x.device) if dres is not None else None
        src2 = getattr(ctx, "res_src", None) if dres is not None else None
        want_cs = getattr(ctx, "want_dxsum", False)
        rows2 = L.sqd_bn_bwd_res_rows(M, C, pre_rows, ctx.code) if src2 is not None or want_cs else 0
        if rows2 > 0:
            part2 = torch.empty(rows2 * C * 2, device=x.device, dtype=torch.float32) if src2 is not None else None
            cpart = torch.empty(rows2, C, device=x.device, dtype=torch.float32) if want_cs else None
            _l.check(L.sqd_bn_train_bwd_res(_ptr(dy), _ptr(x), None, _ptr(mask), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(rstd), _ptr(dx), _ptr(dres),
                                            _ptr(dgamma), _ptr(dbeta), _ptr(part), pre_rows, M, C, ctx.code, _ptr(rp), _ptr(ro), rn, rs,
                                            _ptr(adx), _ptr(adr), _ptr(src2["x"]) if src2 is not None else None, _ptr(src2["mean"]) if src2 is not None else None,
                                            _ptr(src2["rstd"]) if src2 is not None else None, _ptr(part2), _ptr(cpart), _stream()), "bn_train_bwd_res")
            if src2 is not None:
                src2.update(dx=dres, part=part2, rows=rows2)
            if cpart is not None:
                cs = torch.empty(C, device=x.device, dtype=torch.float32)
                _colsum_multi([(cpart, cs, 0)])
                _colsum_tag(dx, cs)                      # column sums of dx [C]: Conv2d.backward takes them as its bias gradient
        else:
            _l.check(L.sqd_bn_train_bwd_amax(_ptr(dy), _ptr(x), None, _ptr(mask), _ptr(gamma), _ptr(beta), _ptr(mean), _ptr(rstd), _ptr(dx), _ptr(dres),
                                             _ptr(dgamma), _ptr(dbeta), _ptr(part), pre_rows, M, C, ctx.code, _ptr(rp), _ptr(ro), rn, rs,
                                             _ptr(adx), _ptr(adr), _stream()), "bn_train_bwd")
        _amax_tag(dx, adx)
        _amax_tag(dres, adr)
        if pend is not None:
            _deferred_grad_done(pend[5])
        return dx, dgamma, dbeta, None, None, dres, None, None, None, None, None, None, None, None


class MaxPool3x3s2(torch.autograd.Function):
    """nn.MaxPool2d(3, 2, 1) of the ResNet stem, channels-last; the backward is a deterministic gather.
    skip=True: -> (y, x') with x' the input handed through this node: a second consumer of x (the decoder's skip connection)
    reads x' instead, and its gradient is added inside the gather kernel instead of a separate accumulation pass."""

    @staticmethod
    def forward(ctx, x, skip=False):
        ctx.set_materialize_grads(False)
        _require(x, "MaxPool3x3s2 input")
        x_in = x
        x = _cl(x)
        N, C, H, W = x.shape
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty((N, C, Ho, Wo), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
        idx = torch.empty(N * Ho * Wo * C, device=x.device, dtype=torch.uint8)
        _l.check(_l.lib().sqd_maxpool3x3s2_fwd(_ptr(x), _ptr(y), _ptr(idx), N, H, W, C, _stream()), "maxpool_fwd")
        ctx.save_for_backward(idx)
        ctx.dims = (N, C, H, W)
        # x is the output of a training-mode BatchNorm + activation (the stem's bn1): the gather of the backward writes that node's whole
        # gradient (skip=True brings the other consumer's in) and can take its two sums on the way — see backward
        ctx.bn_src = getattr(x_in, "_sqd_bn_src", None) if FUSE_BN_BWD_STATS and FUSE_BN_SIDE_SUMS["pool"] and x is x_in else None
        ax = _amax_get(x)
        _amax_tag(y, ax)                         # the outputs are a subset of the inputs: max |x| bounds max |y|
        if skip:
            return y, _amax_tag(x.view_as(x), ax)
        return y

    @staticmethod
    def backward(ctx, dy, g_skip=None):
        idx, = ctx.saved_tensors
        N, C, H, W = ctx.dims
        if dy is None:
            return g_skip, None
        dy = _cl(dy)
        if g_skip is not None:
            g_skip = _cl(g_skip)
        dx = torch.empty((N, C, H, W), device=dy.device, dtype=torch.float32, memory_format=torch.channels_last)
        L = _l.lib()
        src = ctx.bn_src
        rows = L.sqd_maxpool3x3s2_bwd_bn_rows(N, H, W, C) if src is not None and src.get("code") in (0, 1, 2) and tuple(src["x"].shape) == (N, C, H, W) else 0
        if rows > 0:
            part = torch.empty(rows * C * 2, device=dy.device, dtype=torch.float32)
            _l.check(L.sqd_maxpool3x3s2_bwd_bn(_ptr(dy), _ptr(idx), _ptr(g_skip), _ptr(dx), N, H, W, C, _ptr(src["x"]), _ptr(src["mask"]), _ptr(src["mean"]),
                                               _ptr(src["rstd"]), src["code"], _ptr(part), _stream()), "maxpool_bwd_bn")
            src.update(dx=dx, part=part, rows=rows)
        else:
            _l.check(L.sqd_maxpool3x3s2_bwd(_ptr(dy), _ptr(idx), _ptr(g_skip), _ptr(dx), N, H, W, C, _stream()), "maxpool_bwd")
        return dx, None


class UpsampleConcat(torch.autograd.Function):
    """cat([bilinear_resize(x -> skip's H x W, align_corners=True), skip], dim=1), channels-last."""

    @staticmethod
    def forward(ctx, x, skip):
        _require(x, "UpsampleConcat input")
        x_in = x
        x, skip = _cl(x), _cl(skip)
        # x is the output of a training-mode BatchNorm + activation (a decoder stage): the backward's gather writes that node's whole gradient and
        # can take its two sums on the way — see backward
        ctx.bn_src = getattr(x_in, "_sqd_bn_src", None) if FUSE_BN_BWD_STATS and FUSE_BN_SIDE_SUMS["upcat"] and x is x_in else None
        N, Cx, Hi, Wi = x.shape
        _, Cs, Ho, Wo = skip.shape
        out = torch.empty((N, Cx + Cs, Ho, Wo), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
        ao = _amax_out(x.device)
        _l.check(_l.lib().sqd_upcat_fwd_amax(_ptr(x), _ptr(skip), _ptr(out), N, Hi, Wi, Cx, Ho, Wo, Cs, _ptr(ao), _stream()), "upcat_fwd")
        ctx.dims = (N, Hi, Wi, Cx, Ho, Wo, Cs)
        return _amax_tag(out, ao)

    @staticmethod
    def backward(ctx, g_out):
        N, Hi, Wi, Cx, Ho, Wo, Cs = ctx.dims
        g_out = _cl(g_out)
        g_x = torch.empty((N, Cx, Hi, Wi), device=g_out.device, dtype=torch.float32, memory_format=torch.channels_last)
        g_skip = torch.empty((N, Cs, Ho, Wo), device=g_out.device, dtype=torch.float32, memory_format=torch.channels_last)
        L = _l.lib()
        src = getattr(ctx, "bn_src", None)
        rows = L.sqd_upcat_bwd_bn_rows(N, Hi, Wi, Cx, Ho, Wo, Cs) if src is not None and src.get("code") in (0, 1, 2) and tuple(src["x"].shape) == (N, Cx, Hi, Wi) else 0
        if rows > 0:
            part = torch.empty(rows * Cx * 2, device=g_out.device, dtype=torch.float32)
            _l.check(L.sqd_upcat_bwd_bn(_ptr(g_out), _ptr(g_x), _ptr(g_skip), N, Hi, Wi, Cx, Ho, Wo, Cs, _ptr(src["x"]), _ptr(src["mask"]), _ptr(src["mean"]),
                                        _ptr(src["rstd"]), src["code"], _ptr(part), _stream()), "upcat_bwd_bn")
            src.update(dx=g_x, part=part, rows=rows)
        else:
            # (no BatchNorm behind x — the decoder's first 1x1 convolution —: g_x is a convolution's output gradient, and the gather records its
            #  max |.| for that node's two-term fp16 operands instead of a pass of its own)
            agx = _amax_out(g_out.device)
            _l.check(L.sqd_upcat_bwd_bn_amax(_ptr(g_out), _ptr(g_x), _ptr(g_skip), N, Hi, Wi, Cx, Ho, Wo, Cs, None, None, None, None, 0, None, _ptr(agx),
                                             _stream()), "upcat_bwd")
            _amax_tag(g_x, agx)
        return g_x, g_skip


def conv_supported(C, K):
    """Both directions of the native implicit-GEMM convolution (forward + dgrad + wgrad)."""
    return C % 4 == 0 and K % 4 == 0


_PLAN_CACHE = {}


def _conv_ws(mode, geom, device):
    """split-K workspace of sqd_conv_fwd (mode 0) / sqd_conv_dgrad (mode 1); plan sizes cached per geometry."""
    key = (mode,) + tuple(geom)
    n = _PLAN_CACHE.get(key)
    if n is None:
        c = ctypes.c_int64(0)
        _l.lib().sqd_conv_plan(mode, *geom, ctypes.byref(c))
        n = _PLAN_CACHE[key] = c.value
    return torch.empty(n, device=device, dtype=torch.float32) if n else None


TUNE_CONV = False             # set by the Trainer (--sqd_no_conv_tune switches it off)
_TUNED = set()


def set_conv_precision(prec):
    """sqd_conv_set_precision + everything this module remembers about plans: the library drops its measured plans when the
    arithmetic changes (they name kernels of the other mode), so the cached workspace / partial-row counts and the "already
    tuned" marks go with them."""
    L = _l.lib()
    if L.sqd_conv_precision() != prec:
        _PLAN_CACHE.clear()
        _TUNED.clear()
        CHOSEN_PLANS.clear()
    _l.check(L.sqd_conv_set_precision(prec), "conv_set_precision")


CHOSEN_PLANS = {}             # ("fwd" | "dgrad", geom) -> (bm, bn, z, bk);  ("wgrad", N, Ho, Wo, C, K, R, S) -> (impl, splits)
_TUNE_TILES = ((128, 128), (128, 64), (64, 128), (64, 64), (128, 32))
_TUNE_Z = (1, 2, 3, 4, 6, 8, 12, 16)


# Search-space switches of the plan timing.  Defaults are what the product runs; tools/ (A/B scripts) may flip them.
TUNE_SPACE = {
    "input_patch": True,       # bk 32+1024+2048: the input-patch kernel for 3x3 / stride 1 (three-term bf16 operands)
    "bk64": False,             # bk 64+512 on the single-buffered 64x64 tile: picked for 11 layer-modes, no gain on the totals
    "eight_wave": False,       # bk +256: 8-wave workgroups — 1-3 % on a third of the layers, nothing on the step
    "eight_wave_split3": True, # bk 32 + 256 + 1024: the three-term tiles on 8-wave workgroups (conversions of some waves under the MFMAs of others)
    "wgrad_shapes": False,     # smaller register tiles of the direct weight gradient: 7 of 38 layers, nothing on the step
    "wgrad_direct3": True,     # impl 6: three-term bf16 operands straight from memory (even / odd pixel per half wave)
    "wgrad_direct3_8w": True,    # ... its eight-wave workgroups (two waves per SIMD on one tile: conversions of one wave under the MFMAs of the other)
    "wgrad_direct3_wide": True,  # ... its one-wave-per-SIMD tiles (64x128, 128x128)
    "wgrad_rows": True,        # impl 4: the row-window weight gradient of the few-channel / high-resolution layers
    "stats_penalty": False,    # (history: split-K forward plans used to force a BatchNorm statistics pass; their sum takes the partials now)
    "wgrad_transposed": True,  # impl 5: the wide 1x1 layers' weight gradient as a forward GEMM on transposed operands
    "f16x2": True,             # bk + 4096 / impl 7: two-term fp16 operands of the power-of-two scaled tensors (round 5) next to the three-term bf16 plans
    "rounds": 1,               # measurements (of 3 launches each) per candidate plan; the fastest counts
    "log": False,
}


def _time_launch(launch, arg):
    launch(arg)
    best = None
    for _ in range(TUNE_SPACE["rounds"]):         # the fastest of `rounds` measurements of 3 launches
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            launch(arg)
        e1.record()
        e1.synchronize()
        t = e0.elapsed_time(e1)
        best = t if best is None or t < best else best
    return best


def _tune_conv(mode, geom, launch, scaled=False):
    """Time every tile / split-K plan the library accepts for this geometry (4 launches each, HIP events on the current
    stream) and register the fastest (sqd_conv_set_plan).  Runs once per geometry, outside graph capture; geometries with a
    pinned plan (load_plans) are not timed."""
    key = (mode,) + tuple(geom)
    if key in _TUNED or torch.cuda.is_current_stream_capturing():
        return
    _TUNED.add(key)
    L = _l.lib()
    best = None
    by_class = {}                  # (log) the fastest plan of each arithmetic
    # bk + 512 = the single-buffered LDS variants (half the LDS per workgroup, twice the resident workgroups, one more barrier
    # per slice): picked for three quarters of the config-B layers, forward -5 %, data gradient -2 %.
    # bk 32 + 1024 = three-term bf16 operands on the bf16 matrix cores (fp32-level accuracy, 6 products per slice at 16x the fp32
    # MFMA rate, single LDS buffer): 20-30 % faster than the fp32 kernels on most config-B layers (profiles/r02e_conv_split3.md)
    # bk 32 + 1024 + 2048 = the input-patch kernel for 3x3 / stride 1 / pad 1 (three-term bf16 operands; the input patch of a
    # 32-channel chunk is converted and staged once for all nine taps, the filter fragments come straight from L2): 20-40 % faster
    # than the implicit-GEMM plans on the config-B layers above 12x40 pixels (profiles/r02i_conv_input_patch.md)
    # (bk 64 + 1024, the three-term 64x64 tile with 64-channel slices, was tried for the few-pixel / many-channel 1x1 layers of layer 3 / 4:
    # no change of the step in a same-box A/B — not kept)
    if L.sqd_conv_precision() != 0:
        bks = (16, 32)             # --sqd_bf16 runs one kernel family: tile and split-K are all there is to choose
    else:
        bks = (16, 32, 528, 544, 1056) + ((3104,) if TUNE_SPACE["input_patch"] else ()) + ((576,) if TUNE_SPACE["bk64"] else ()) + \
            ((272, 288) if TUNE_SPACE["eight_wave"] else ()) + ((1312,) + ((3360,) if TUNE_SPACE["input_patch"] else ()) if TUNE_SPACE["eight_wave_split3"] else ())
        if scaled and TUNE_SPACE["f16x2"]:
            # + 4096: the same tiles on two-term fp16 operands (the launch carries the operands' max |.|): half the matrix and conversion
            # instructions of the three-term plans (profiles/r05a_f16x2_layers.txt: forward -23 %, data gradient -21 %)
            bks += tuple(b + 4096 for b in bks if b & 1024)
    for bm, bn, z, bk in ((bm, bn, z, bk) for bm, bn in _TUNE_TILES + ((64, 32),) for bk in bks for z in _TUNE_Z):
        if L.sqd_conv_set_plan(mode, *geom, bm, bn, z, bk) != 0:
            continue
        _PLAN_CACHE.pop(key, None)
        ws = _conv_ws(mode, geom, torch.device("cuda", torch.cuda.current_device()))
        t = _time_launch(launch, ws)
        if mode == 0 and z > 1 and TUNE_SPACE["stats_penalty"]:
            # a forward plan that splits the reduction writes no BatchNorm partials: the BatchNorm that follows (nearly every
            # convolution of these networks has one) then reads the output once more for its statistics — charge that pass
            # (output bytes at ~4 TB/s + a launch) for the 3 timed launches
            t += 3.0 * (geom[0] * geom[9] * geom[10] * geom[4] * 4 / 4.0e12 + 3.0e-6) * 1e3
        if best is None or t < best[0]:
            best = (t, bm, bn, z, bk)
        cls = "f16x2" if bk & 4096 else "bf16x3" if bk & 1024 else "fp32"
        if cls not in by_class or t / 3 * 1e3 < by_class[cls][0]:
            by_class[cls] = (round(t / 3 * 1e3, 1), bm, bn, z, bk)
    plan = (0, 0, 0, 16) if best is None else best[1:]
    _register_conv_plan(mode, geom, plan)
    if TUNE_SPACE["log"]:
        print("sqd conv plan", "dgrad" if mode else "fwd", geom, best, "us per launch by arithmetic:", by_class, "scaled" if scaled else "unscaled", flush=True)


def _register_conv_plan(mode, geom, plan):
    key = (mode,) + tuple(geom)
    _PLAN_CACHE.pop(key, None)
    _PLAN_CACHE.pop(("s",) + tuple(geom), None)
    _PLAN_CACHE.pop(("sd",) + tuple(geom), None)
    _l.check(_l.lib().sqd_conv_set_plan(mode, *geom, *plan), "conv_set_plan")
    CHOSEN_PLANS[("dgrad" if mode else "fwd",) + tuple(geom)] = tuple(plan)


def _tune_wgrad(geom, has_bias, launch, launch_t=None, scaled=False):
    """Same for the weight gradient: direct-operand vs LDS-tiled kernel, 1/8x .. 4x the model's pixel splits; launch_t (wide 1x1 layers):
    the forward GEMM on transposed operands."""
    N, H, W, C, K, R, S, stride, pad, Ho, Wo = geom
    key = _wgrad_key(geom)
    if key in _TUNED or torch.cuda.is_current_stream_capturing():
        return
    _TUNED.add(key)
    L = _l.lib()
    L.sqd_conv_wgrad_set_plan(N, Ho, Wo, C, K, R, S, -1, 0)
    _PLAN_CACHE.pop(key, None)
    _, base = _wgrad_part_floats(geom)
    best = None
    by_impl = {}                   # (log) the fastest plan of each kernel family

    def trial(impl, sp):
        nonlocal best
        if L.sqd_conv_wgrad_set_plan(N, Ho, Wo, C, K, R, S, impl, sp) != 0:
            return False
        if L.sqd_conv_wgrad_effective_impl(*geom) != ((impl & 15) if impl & 15 else 0) and (impl & 15) in (4, 6, 7):
            # the table keys on the output geometry: THIS convolution (strided / padded, odd Wo) would run the fp32 direct kernel under
            # that entry — timing it under the candidate's name would rank and report a kernel that did not run
            return False
        _PLAN_CACHE.pop(key, None)
        pf, splits = _wgrad_part_floats(geom)
        extra = max((N * Ho * Wo + 1023) // 1024, splits) * K if has_bias else 0
        t = _time_launch(launch, torch.empty(pf + extra, device="cuda", dtype=torch.float32))
        if best is None or t < best[0]:
            best = (t, impl, sp)
        if (impl & 15) not in by_impl or t / 3 * 1e3 < by_impl[impl & 15][0]:
            by_impl[impl & 15] = (round(t / 3 * 1e3, 1), impl, sp)
        return True
    # direct kernel with its default register tile (the widest that divides: 64 filters x 64 channels), then the LDS-tiled kernel
    shapes = [1]
    if TUNE_SPACE["wgrad_shapes"]:
        shapes += [1 | (kt << 4) | (ct << 8) for kt, ct in ((2, 4), (4, 2), (2, 2)) if K % (16 * kt) == 0 and C % (16 * ct) == 0
                   and (K % 64 == 0 or kt < 4) and (C % 64 == 0 or ct < 4)]
    for impl in shapes + [0]:
        if impl & 1 and (C % 16 or K % 16):
            continue
        tried = set()
        for mult in (0.125, 0.25, 0.5, 1, 2, 4):
            sp = max(1, int(base * mult))
            if sp not in tried:
                tried.add(sp)
                trial(impl, sp)
    # shared-operand kernels (impl 2 = fp32 MFMA, 3 = three-term bf16 operands; + 16 * variant: blocks of 128x128 / 64x128 / 128x64 /
    # 64x64 filters x channels staged once per workgroup in LDS): pixel splits for ~256 .. 1536 workgroups
    for base_impl, v, (tk, tc) in ((b, v, blk) for b in (2, 3) for v, blk in enumerate(((128, 128), (64, 128), (128, 64), (64, 64)))):
        if K % tk or C % tc:
            continue
        blocks = (K // tk) * (C // tc) * R * S
        tried = set()
        for target in (256, 512, 768, 1024, 1536):
            sp = max(1, (target + blocks - 1) // blocks)
            if sp not in tried and trial(base_impl | (v << 4), sp):
                tried.add(sp)
    # three-term bf16 operands straight from memory (impl 6 + 16 * variant; round 4): register tiles of 64x64, 128x64, 64x128, 128x128
    # (filters x channels) per wave — the wider the tile, the more pixel splits it takes to fill the chip
    if TUNE_SPACE["wgrad_direct3"]:
        # pixel splits for a whole number of workgroup rounds: the 64x64 and 128x64 tiles keep two workgroups per CU resident, the
        # 64x128 and 128x128 ones one
        for v, (tk, tc), per_cu in ((0, (64, 64), 2), (1, (64, 32), 2), (2, (32, 64), 2), (3, (128, 64), 2), (4, (64, 128), 1), (5, (128, 128), 1), (6, (128, 64), 1), (7, (64, 64), 1)):
            if v in (1, 2) and K % 64 == 0 and C % 64 == 0:     # (the half-width tiles are for the 32-filter / 32-channel layers the 64x64 tile does not divide)
                continue
            if K % tk or C % tc or (per_cu == 1 and v < 6 and not TUNE_SPACE["wgrad_direct3_wide"]) or (v >= 6 and not TUNE_SPACE["wgrad_direct3_8w"]):
                continue
            tiles = (K // tk) * (C // tc) * R * S
            for impl in (6, 7) if scaled and TUNE_SPACE["f16x2"] else (6,):          # 7: the same tiles on two-term fp16 operands
                tried = set()
                for target in (128, 192, 256, 384, 512, 768, 1024):
                    sp = max(1, (target * per_cu) // tiles)
                    if sp not in tried:
                        tried.add(sp)
                        if not trial(impl | (v << 4), sp):
                            break
    # row-window kernel (impl 4): few channels, stride 1 — the whole filter bank in one workgroup's accumulators, `sp` workgroups
    # (the library refuses the shapes it is not built for)
    if stride == 1 and R == S and TUNE_SPACE["wgrad_rows"]:
        for sp in (256, 384, 512, 768, 1024):
            if not trial(4, sp):
                break
    if launch_t is not None and TUNE_SPACE["wgrad_transposed"]:
        launch_t(None)                                # (times the forward plans of the transposed problem first)
        t = _time_launch(launch_t, None)
        if t < best[0]:
            best = (t, WGRAD_TRANSPOSED, 0)
        else:                                         # not taken: the transposed problem's forward plan is nobody's plan
            gT = (1, 1, K, N * Ho * Wo, C, 1, 1, 1, 0, 1, K)
            if CHOSEN_PLANS.pop(("fwd",) + gT, None) is not None:
                L.sqd_conv_set_plan(0, *gT, 0, 0, 0, 16)
                _PLAN_CACHE.pop((0,) + gT, None)
    _register_wgrad_plan((N, Ho, Wo, C, K, R, S), best[1:])
    if TUNE_SPACE["log"]:
        print("sqd conv plan wgrad", geom, best, "model splits", base, "us per launch by kernel family:", by_impl, "scaled" if scaled else "unscaled", flush=True)


WGRAD_TRANSPOSED = 5          # plan impl handled here, not in the library: the wide 1x1 layers' weight gradient as a forward GEMM on
                              # transposed operands (_wgrad_transposed)


def _register_wgrad_plan(wkey, plan):
    _PLAN_CACHE.pop(("w",) + tuple(wkey), None)
    if plan[0] == WGRAD_TRANSPOSED:
        _l.lib().sqd_conv_wgrad_set_plan(*wkey, -1, 0)
    else:
        _l.check(_l.lib().sqd_conv_wgrad_set_plan(*wkey, *plan), "conv_wgrad_set_plan")
    CHOSEN_PLANS[("wgrad",) + tuple(wkey)] = tuple(plan)


def wgrad_transposed_applies(geom):
    """1x1, stride 1, unpadded, a reduction (pixel count) that the GEMM kernels take as a channel count, filters and channels wide
    enough that the two transposes are small against the product"""
    N, H, W, C, K, R, S, stride, pad, Ho, Wo = geom
    M = N * Ho * Wo
    # (measured: taken for the ConvNeXt-L stage 3 / 4 MLPs — 5120 and 1280 rows, 768..6144 features —, never at 20480+ rows, where the
    #  two transposes outweigh the faster product, nor for ResNet-50's 1x1 layers)
    # fp32 arithmetic only: under --sqd_bf16 the forward kernels round their operands to ONE bf16 term, and the weight gradients of that
    # mode stay fp32 — the product on transposed operands would silently compute them in bf16
    return (R == 1 and S == 1 and stride == 1 and pad == 0 and M % 4 == 0 and 256 <= M <= 8192 and min(C, K) >= 256 and C * K >= 512 * 1024
            and C % 4 == 0 and K % 4 == 0 and _l.lib().sqd_conv_precision() == 0)


def _wgrad_transposed(dy, x, dw, db, geom, ady=None, ax=None):
    """dW [K][C] = sum_m dY[m][k] X[m][c] reduces over the slow axis of both operands — the layout the fp32 weight-gradient kernels are
    built around and the reason they end at ~95 TFLOP/s.  Transposed (two HBM-rate passes, sqd_transpose2d), it is the FORWARD problem
    "K pixels x M channels -> C filters": sqd_conv_fwd on its measured plan (three-term bf16 operands at ~175 TFLOP/s effective on the
    ConvNeXt-L block MLPs).  The bias gradient comes from the column sums the transpose of dY takes on the way."""
    N, H, W, C, K, R, S, stride, pad, Ho, Wo = geom
    M = N * Ho * Wo
    L = _l.lib()
    dyT = torch.empty(K * M, device=dy.device, dtype=torch.float32)
    xT = torch.empty(C * M, device=dy.device, dtype=torch.float32)
    cs = torch.empty(((M + 63) // 64, K), device=dy.device, dtype=torch.float32) if db is not None else None
    _l.check(L.sqd_transpose2d(_ptr(dy), _ptr(dyT), M, K, _ptr(cs), _stream()), "transpose2d")
    _l.check(L.sqd_transpose2d(_ptr(x), _ptr(xT), M, C, None, _stream()), "transpose2d")
    gT = (1, 1, K, M, C, 1, 1, 1, 0, 1, K)           # N, H, W, "channels" = M, "filters" = C, 1x1 -> [1, 1, K] pixels x C
    # (a transpose moves values, it does not change them: the operands' max |.| are those of dy and x)
    run = lambda ws: L.sqd_conv_fwd_scaled(_ptr(dyT), _ptr(xT), None, _ptr(dw), _ptr(ws), None, _ptr(ady), _ptr(ax), None, *gT, 0, _stream())
    if TUNE_CONV:
        _tune_conv(0, gT, run, ady is not None and ax is not None)
    ws = _conv_ws(0, gT, dy.device)
    _l.check(run(ws), "conv_fwd (transposed weight gradient)")
    if db is not None:
        _colsum_multi([(cs, db, 0)])


def export_plans():
    """The measured plan set of this process as a JSON-able record: a run loaded from it (load_plans, --sqd_conv_plans) executes
    the same kernels on the same tiles and splits — last-bit reproducible across boxes, where first-step timing is not."""
    return {"abi": _l.lib().sqd_abi_version(), "precision": _l.lib().sqd_conv_precision(),
            "plans": [{"pass": k[0], "geom": list(k[1:]), "plan": list(v)} for k, v in sorted(CHOSEN_PLANS.items(), key=repr)]}


def load_plans(rec):
    """Register a plan set written by export_plans; the geometries it names are not timed again."""
    if rec.get("precision", 0) != _l.lib().sqd_conv_precision():
        raise RuntimeError("sqd: the plan file was measured with convolution precision %s, this run uses %s"
                           % (rec.get("precision"), _l.lib().sqd_conv_precision()))
    for e in rec["plans"]:
        geom, plan = tuple(e["geom"]), tuple(e["plan"])
        if e["pass"] == "wgrad":
            _register_wgrad_plan(geom, plan)
            _TUNED.add(("w",) + geom)
        else:
            mode = 1 if e["pass"] == "dgrad" else 0
            _register_conv_plan(mode, geom, plan)
            _TUNED.add((mode,) + geom)


# weight-gradient geometries (output-geometry keys) whose backward has run at least once in this process: their plan is settled — timed in that
# first backward, or pinned from a file — so a decision taken from it (BatchNormAct.forward: who delivers the bias gradient) is the same in the run
# that timed the plans and in a run that loads them: both decide from their second step on, both from the same plan (the plan-pinning contract of
# tests/test_gpu_plans.py: a pinned run trains to the bits of the run that wrote the file)
_WGRAD_SETTLED = set()


def reset_plans():
    """Forget every measured / pinned plan (library table and this module's caches) and stop timing plans: the cost-model plans
    apply again."""
    L = _l.lib()
    for k in list(CHOSEN_PLANS):
        if k[0] == "wgrad":
            L.sqd_conv_wgrad_set_plan(*k[1:], -1, 0)
        else:
            L.sqd_conv_set_plan(1 if k[0] == "dgrad" else 0, *k[1:], 0, 0, 0, 16)
    CHOSEN_PLANS.clear()
    _TUNED.clear()
    _PLAN_CACHE.clear()
    _WGRAD_SETTLED.clear()
    global TUNE_CONV
    TUNE_CONV = False             # (nnops.configure switches plan timing on again for the next Trainer that wants it)


def plan_mix():
    """How many geometries of each pass run which arithmetic / kernel family under the registered plans (bench line)."""
    mix = {}
    for k, v in CHOSEN_PLANS.items():
        if k[0] == "wgrad":
            name = {0: "fp32 lds-tiled", 1: "fp32 direct", 2: "fp32 shared-operand", 3: "bf16x3 shared-operand", 4: "fp32 row-window", 5: "transposed forward-gemm (its own fwd plan)",
                    6: "bf16x3 direct-operand", 7: "f16x2 direct-operand"}[v[0] & 15]
        else:
            bk = v[3]
            arith = "f16x2" if bk & 4096 else "bf16x3"
            name = arith + " input-patch" if bk & 2048 else arith + " implicit-gemm" if bk & 1024 else "fp32 implicit-gemm"
            if _l.lib().sqd_conv_precision() == 2:
                name = "bf16 implicit-gemm"
        mix.setdefault(k[0], {})
        mix[k[0]][name] = mix[k[0]].get(name, 0) + 1
    return mix


def _wgrad_key(geom):
    """The library keys its weight-gradient plans by (N, Ho, Wo, C, K, R, S): two convolutions that differ only in stride / padding
    / input size (layer2.0.conv2 at stride 2 and layer2.1.conv2 at stride 1 have the same output) share ONE plan.  The tuner and
    the cached workspace size use the same key — a per-convolution cache went stale when the second one re-tuned the shared
    entry to more pixel splits, and the first one's partial buffer was then too small for its own launch."""
    N, H, W, C, K, R, S, stride, pad, Ho, Wo = geom
    return ("w", N, Ho, Wo, C, K, R, S)


def _wgrad_part_floats(geom):
    N, H, W, C, K, R, S, stride, pad, Ho, Wo = geom
    key = _wgrad_key(geom)
    n = _PLAN_CACHE.get(key)
    if n is None:
        splits, pf = ctypes.c_int(0), ctypes.c_int64(0)
        _l.lib().sqd_conv_wgrad_plan(N, Ho, Wo, C, K, R, S, ctypes.byref(splits), ctypes.byref(pf))
        n = _PLAN_CACHE[key] = (pf.value, splits.value)
    return n


def tf_same_pad(size, k, stride):
    """TensorFlow "SAME" padding of one axis (the tf_efficientnet variants): -> (output size, leading pad)"""
    out = -(-size // stride)
    total = max((out - 1) * stride + k - size, 0)
    return out, total // 2


class DepthwiseConv(torch.autograd.Function):
    """Depthwise k x k convolution (groups == channels), k in {3, 5, 7}, stride in {1, 2}, channels-last.
    forward(x [N,C,H,W], weight [C,1,k,k], stride, pad) — pad: an int (symmetric) or "same" (TensorFlow SAME, asymmetric).
    skip=True (stride 1): -> (y, x') with x' the input handed through this node: the input's other consumer (a block's shortcut) reads
    x' instead, and its gradient is added inside the data-gradient kernel instead of by a separate accumulation pass."""

    @staticmethod
    def forward(ctx, x, weight, stride, pad, skip=False):
        ctx.set_materialize_grads(False)
        _require(x, "DepthwiseConv input")
        x = _cl(x)
        N, C, H, W = x.shape
        k = weight.shape[2]
        if pad == "same":
            (Ho, pt), (Wo, pl) = tf_same_pad(H, k, stride), tf_same_pad(W, k, stride)
        else:
            pt = pl = int(pad)
            Ho, Wo = (H + 2 * pt - k) // stride + 1, (W + 2 * pl - k) // stride + 1
        L = _l.lib()
        wt = torch.empty(k * k, C, device=x.device, dtype=torch.float32)
        _l.check(L.sqd_dw_weight_layout(_ptr(weight.contiguous()), _ptr(wt), C, k, 1, _stream()), "dw_weight_layout")
        y = torch.empty((N, C, Ho, Wo), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
        _l.check(L.sqd_dw_conv_fwd(_ptr(x), _ptr(wt), _ptr(y), N, H, W, C, k, stride, pt, pl, Ho, Wo, _stream()), "dw_conv_fwd")
        ctx.save_for_backward(x, wt)
        ctx.geom = (N, H, W, C, k, stride, pt, pl, Ho, Wo)
        if skip:
            assert stride == 1, "DepthwiseConv: the hand-over needs stride 1"
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, g_skip=None):
        x, wt = ctx.saved_tensors
        N, H, W, C, k, stride, pt, pl, Ho, Wo = ctx.geom
        if dy is None:                                   # only the pass-through output was used
            return g_skip, None, None, None, None
        dy = _cl(dy)
        g_skip = _cl(g_skip) if g_skip is not None else None
        L = _l.lib()
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty((N, C, H, W), device=dy.device, dtype=torch.float32, memory_format=torch.channels_last)
            _l.check(L.sqd_dw_conv_dgrad_add(_ptr(dy), _ptr(wt), _ptr(g_skip), _ptr(dx), N, H, W, C, k, stride, pt, pl, Ho, Wo, _stream()),
                     "dw_conv_dgrad")
        elif g_skip is not None:
            dx = g_skip
        if ctx.needs_input_grad[1]:
            chunks = L.sqd_dw_conv_wgrad_chunks(N, Ho, Wo)
            part = torch.empty(chunks, k * k * C, device=dy.device, dtype=torch.float32)
            _l.check(L.sqd_dw_conv_wgrad(_ptr(dy), _ptr(x), _ptr(part), N, H, W, C, k, stride, pt, pl, Ho, Wo, _stream()), "dw_conv_wgrad")
            dwt = torch.empty(k * k, C, device=dy.device, dtype=torch.float32)
            _colsum_multi([(part, dwt, 0)])
            dw = torch.empty(C, 1, k, k, device=dy.device, dtype=torch.float32)
            _l.check(L.sqd_dw_weight_layout(_ptr(dwt), _ptr(dw), C, k, 0, _stream()), "dw_weight_layout")
        return dx, dw, None, None, None


class LayerNormRows(torch.autograd.Function):
    """LayerNorm over the channels of every pixel of a channels-last map (timm LayerNorm2d; the nn.LayerNorm of a ConvNeXt block
    between its permutes), eps as given; `pre_bias` [C] or None is added to x first (the depthwise convolution's bias).
    forward(x [N,C,H,W] channels-last, pre_bias, gamma [C], beta [C], eps) -> [N,C,H,W] channels-last"""

    @staticmethod
    def forward(ctx, x, pre_bias, gamma, beta, eps):
        _require(x, "LayerNormRows input")
        x = _cl(x)
        N, C, H, W = x.shape
        M = N * H * W
        y = torch.empty_like(x)
        mean, rstd = torch.empty(M, device=x.device, dtype=torch.float32), torch.empty(M, device=x.device, dtype=torch.float32)
        ay = _amax_out(x.device)                 # max |y| for the Linear layer that reads y on two-term fp16 operands (no pass of its own)
        _l.check(_l.lib().sqd_ln_rows_fwd_amax(_ptr(x), _ptr(pre_bias), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(mean), _ptr(rstd), M, C, float(eps),
                                               _ptr(ay), _stream()), "ln_rows_fwd")
        _amax_tag(y, ay)
        ctx.save_for_backward(x, pre_bias, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, pre_bias, gamma, mean, rstd = ctx.saved_tensors
        N, C, H, W = x.shape
        M = N * H * W
        dy = _cl(dy)
        L = _l.lib()
        dx = torch.empty_like(x)
        part = torch.empty(L.sqd_ln_rows_nblk(M), 3 * C, device=x.device, dtype=torch.float32)
        _l.check(L.sqd_ln_rows_bwd(_ptr(dy), _ptr(x), _ptr(pre_bias), _ptr(gamma), _ptr(mean), _ptr(rstd), _ptr(dx), _ptr(part), M, C, _stream()),
                 "ln_rows_bwd")
        sums = torch.empty(3 * C, device=x.device, dtype=torch.float32)
        _colsum_multi([(part, sums, 0)])
        return dx, (sums[2 * C:] if pre_bias is not None else None), sums[:C], sums[C:2 * C], None


class Gelu(torch.autograd.Function):
    """exact (erf) GELU, element-wise; forward(x) -> gelu(x)"""

    @staticmethod
    def forward(ctx, x):
        _require(x, "Gelu input")
        if not (x.is_contiguous() or x.is_contiguous(memory_format=torch.channels_last)):
            x = x.contiguous()
        y = torch.empty_like(x)
        ay = _amax_out(x.device)
        _l.check(_l.lib().sqd_gelu_fwd_amax(_ptr(x), _ptr(y), x.numel(), _ptr(ay), _stream()), "gelu_fwd")
        _amax_tag(y, ay)
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        dy = dy.contiguous(memory_format=torch.channels_last) if x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) else dy.contiguous()
        dx = torch.empty_like(x)
        ad = _amax_out(x.device)
        L = _l.lib()
        if x.dim() == 4 and x.is_contiguous(memory_format=torch.channels_last) and x.shape[1] % 4 == 0 and x.shape[1] * 16 <= 160 * 1024:
            # rows [N*H*W, C]: the pass also leaves the per-block column sums of dx — the bias gradient of the Linear layer in front
            N, C, H, W = x.shape
            M = N * H * W
            part = torch.empty(L.sqd_scale_residual_nblk(M), C, device=x.device, dtype=torch.float32)
            _l.check(L.sqd_gelu_bwd_rows(_ptr(x), _ptr(dy), _ptr(dx), _ptr(part), M, C, _ptr(ad), _stream()), "gelu_bwd_rows")
            cs = torch.empty(C, device=x.device, dtype=torch.float32)
            _colsum_multi([(part, cs, 0)])
            _colsum_tag(dx, cs)                          # column sums of dx [C]: Conv2d.backward takes them as its bias gradient
        else:
            _l.check(L.sqd_gelu_bwd_amax(_ptr(x), _ptr(dy), _ptr(dx), x.numel(), _ptr(ad), _stream()), "gelu_bwd")
        return _amax_tag(dx, ad)


class ScaleResidual(torch.autograd.Function):
    """shortcut + gamma[c] * z: the layer scale and residual add that close a ConvNeXt block.
    forward(res [N,C,H,W], z [N,C,H,W], gamma [C]) (channels-last)"""

    @staticmethod
    def forward(ctx, res, z, gamma):
        _require(z, "ScaleResidual input")
        res, z = _cl(res), _cl(z)
        N, C, H, W = z.shape
        out = torch.empty_like(z)
        _l.check(_l.lib().sqd_scale_residual_fwd(_ptr(res), _ptr(z), _ptr(gamma), _ptr(out), N * H * W, C, _stream()), "scale_residual_fwd")
        ctx.save_for_backward(z, gamma)
        return out

    @staticmethod
    def backward(ctx, dy):
        z, gamma = ctx.saved_tensors
        N, C, H, W = z.shape
        M = N * H * W
        dy = _cl(dy)
        L = _l.lib()
        dz = torch.empty_like(z)
        part = torch.empty(L.sqd_scale_residual_nblk(M), C, device=z.device, dtype=torch.float32)
        ad = _amax_out(z.device)
        part2 = torch.empty_like(part) if C * 32 <= 160 * 1024 else None       # column sums of dz: the bias gradient of the Linear layer that wrote z
        _l.check(L.sqd_scale_residual_bwd_sums(_ptr(dy), _ptr(z), _ptr(gamma), _ptr(dz), _ptr(part), _ptr(part2), M, C, _ptr(ad), _stream()),
                 "scale_residual_bwd")
        dgamma = torch.empty(C, device=z.device, dtype=torch.float32)
        if part2 is not None:
            cs = torch.empty(C, device=z.device, dtype=torch.float32)
            _colsum_multi([(part, dgamma, 0), (part2, cs, 0)])
            _colsum_tag(dz, cs)                          # column sums of dz [C]
        else:
            _colsum_multi([(part, dgamma, 0)])
        return dy, _amax_tag(dz, ad), dgamma


class Upsample2x(torch.autograd.Function):
    """F.interpolate(x, scale_factor=2, mode='bilinear') (align_corners False) on a channels-last map (reference Unet.py:250)"""

    @staticmethod
    def forward(ctx, x):
        _require(x, "Upsample2x input")
        x = _cl(x)
        N, C, H, W = x.shape
        y = torch.empty((N, C, 2 * H, 2 * W), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
        _l.check(_l.lib().sqd_upsample2x_fwd(_ptr(x), _ptr(y), N, H, W, C, _stream()), "upsample2x_fwd")
        ctx.shape = (N, C, H, W)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, C, H, W = ctx.shape
        dy = _cl(dy)
        dx = torch.empty((N, C, H, W), device=dy.device, dtype=torch.float32, memory_format=torch.channels_last)
        _l.check(_l.lib().sqd_upsample2x_bwd(_ptr(dy), _ptr(dx), N, H, W, C, _stream()), "upsample2x_bwd")
        return dx


class SqueezeExcite(torch.autograd.Function):
    """x * sigmoid(W2 . swish(W1 . mean_hw(x) + b1) + b2) — the squeeze-and-excite gate of the MBConv blocks.
    forward(x [B,C,H,W], w1 [R,C,1,1], b1 [R], w2 [C,R,1,1], b2 [C])"""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2):
        _require(x, "SqueezeExcite input")
        pooled = getattr(x, "_sqd_pool_part", None) if x.is_contiguous(memory_format=torch.channels_last) else None
        x = _cl(x)
        B, C, H, W = x.shape
        R, HW = w1.shape[0], H * W
        L = _l.lib()
        W1, W2 = w1.reshape(R, C).contiguous(), w2.reshape(C, R).t().contiguous()      # W2: [R, C] (transposed: coalesced along C)
        dev = x.device
        part = pooled                                   # the producing BatchNorm's element-wise pass took the sums already
        if part is None or tuple(part.shape) != (B, L.sqd_se_chunks(HW), C):
            part = torch.empty(B, L.sqd_se_chunks(HW), C, device=dev, dtype=torch.float32)
            _l.check(L.sqd_se_pool(_ptr(x), None, _ptr(part), B, HW, C, _stream()), "se_pool")
        s, pre1, gate = (torch.empty(B, n, device=dev, dtype=torch.float32) for n in (C, R, C))
        _l.check(L.sqd_se_gate_fwd(_ptr(part), _ptr(W1), _ptr(b1), _ptr(W2), _ptr(b2), _ptr(s), _ptr(pre1), _ptr(gate), B, HW, C, R, _stream()),
                 "se_gate_fwd")
        y = torch.empty_like(x, memory_format=torch.channels_last)
        _l.check(L.sqd_se_scale(_ptr(x), _ptr(gate), None, _ptr(y), B, HW, C, _stream()), "se_scale")
        ctx.save_for_backward(x, W1, W2, s, pre1, gate)
        ctx.shapes = (w1.shape, w2.shape)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W1, W2, s, pre1, gate = ctx.saved_tensors
        B, C, H, W = x.shape
        R, HW = W1.shape[0], H * W
        RP = (R + 3) // 4 * 4
        dy = _cl(dy)
        L = _l.lib()
        dev = dy.device
        dgpart = torch.empty(B, L.sqd_se_chunks(HW), C, device=dev, dtype=torch.float32)
        _l.check(L.sqd_se_pool(_ptr(dy), _ptr(x), _ptr(dgpart), B, HW, C, _stream()), "se_pool")
        dW1p, db1p, dW2p, db2p, ds = (torch.empty(B, n, device=dev, dtype=torch.float32) for n in (R * C, RP, C * R, C, C))
        _l.check(L.sqd_se_gate_bwd(_ptr(dgpart), _ptr(W1), _ptr(W2), _ptr(s), _ptr(pre1), _ptr(gate), _ptr(dW1p), _ptr(db1p), _ptr(dW2p),
                                   _ptr(db2p), _ptr(ds), B, HW, C, R, _stream()), "se_gate_bwd")
        dx = torch.empty_like(x, memory_format=torch.channels_last)
        _l.check(L.sqd_se_scale(_ptr(dy), _ptr(gate), _ptr(ds), _ptr(dx), B, HW, C, _stream()), "se_scale")
        dW1, dW2 = torch.empty(R, C, device=dev, dtype=torch.float32), torch.empty(C, R, device=dev, dtype=torch.float32)
        db1, db2 = torch.empty(RP, device=dev, dtype=torch.float32), torch.empty(C, device=dev, dtype=torch.float32)
        _colsum_multi([(dW1p, dW1, 0), (db1p, db1, 0), (dW2p, dW2, 0), (db2p, db2, 0)])
        return dx, dW1.view(ctx.shapes[0]), db1[:R], dW2.view(ctx.shapes[1]), db2


class PoseHead(torch.autograd.Function):
    """scale * pose_conv(x).mean(3).mean(2) of PoseCNN (reference networks/pose_cnn.py:40-42) as one launch each way.
    forward(x [B,C,h,w] channels-last, weight [J,C,1,1], bias [J], scale, split) -> [B,J], or with split=True (J = 6 F) the
    reference's (axisangle, translation) = (out[..., :3], out[..., 3:]) of out.view(B, F, 1, 6) as two dense [B,F,1,3] tensors
    (pose_cnn.py:44-45): the photometric chain reads them as they are and their gradients come back as two tensors too."""

    @staticmethod
    def forward(ctx, x, weight, bias, scale, split=False):
        _require(x, "PoseHead input")
        x = _cl(x)
        B, C, h, w = x.shape
        J = weight.shape[0]
        wm = weight.reshape(J, C).contiguous()
        if split:
            out = torch.empty(B, J // 6, 1, 3, device=x.device, dtype=torch.float32)
            out2 = torch.empty_like(out)
        else:
            out, out2 = torch.empty(B, J, device=x.device, dtype=torch.float32), None
        mean = torch.empty(B, C, device=x.device, dtype=torch.float32)
        _l.check(_l.lib().sqd_pose_head_fwd(_ptr(x), _ptr(wm), _ptr(bias), _ptr(out), _ptr(out2), _ptr(mean), B, h, w, C, J, float(scale),
                                            _stream()), "pose_head_fwd")
        ctx.save_for_backward(wm, mean)
        ctx.dims = (B, C, h, w, J, float(scale), weight.shape, split)
        return (out, out2) if split else out

    @staticmethod
    def backward(ctx, g, g2=None):
        wm, mean = ctx.saved_tensors
        B, C, h, w, J, scale, wshape, split = ctx.dims
        if split:                                        # an unused half arrives as None
            g = g.contiguous() if g is not None else torch.zeros(B, J // 6, 1, 3, device=wm.device)
            g2 = g2.contiguous() if g2 is not None else torch.zeros(B, J // 6, 1, 3, device=wm.device)
        else:
            g = g.contiguous()
        dx = torch.empty((B, C, h, w), device=g.device, dtype=torch.float32, memory_format=torch.channels_last)
        dWp_c = torch.empty(B, J * C, device=g.device, dtype=torch.float32)
        JP = (J + 3) // 4 * 4
        dbp_c = torch.empty(B, JP, device=g.device, dtype=torch.float32)
        _l.check(_l.lib().sqd_pose_head_bwd(_ptr(g), _ptr(g2) if split else None, _ptr(wm), _ptr(mean), _ptr(dx), _ptr(dWp_c), _ptr(dbp_c), B,
                                            h * w, C, J, scale, _stream()), "pose_head_bwd")
        dW = torch.empty(J, C, device=g.device, dtype=torch.float32)
        db = torch.empty(JP, device=g.device, dtype=torch.float32)
        _colsum_multi([(dWp_c, dW, 0), (dbp_c, db, 0)])
        return dx, dW.view(wshape), db[:J], None, None


def linear_native(x, lin, act=None):
    """nn.Linear on a few rows (the 12-row bins regressor, reference networks/depth_decoder_QTR.py:22-26,49) as a 1x1
    convolution over rows-as-pixels: the implicit-GEMM kernels with split-K read the weight matrix once, the data / weight
    gradients come from the same node.  in_features and out_features must be multiples of 16."""
    rows, K = x.shape[0], lin.out_features
    x4 = x.reshape(rows, lin.in_features, 1, 1)
    w4 = lin.weight.view(K, lin.in_features, 1, 1)
    w4._sqd_w_src = lin.weight
    return Conv2d.apply(x4, w4, lin.bias, 1, 0, act, False, None, None).reshape(rows, K)


def linear_supported(lin, x):
    return x.dim() == 2 and lin.in_features % 4 == 0 and lin.out_features % 4 == 0


def conv_module_supported(conv):
    s, p = conv.stride, conv.padding
    return (conv.in_channels % 4 == 0 and conv.out_channels % 4 == 0 and s[0] == s[1] and p[0] == p[1]
            and conv.dilation == (1, 1) and conv.groups == 1 and not isinstance(p, str))


# stream the weight-gradient kernels run on (None: the caller's stream).  The Trainer sets it; whoever calls backward() must
# call join_wgrad_stream() before the gradients are read (optimiser, all-reduce).
WGRAD_STREAM = None

# Sum of a weight gradient's pixel splits as part of the next BatchNorm-backward launch instead of a launch of its own (the Trainer turns
# this on for single-rank runs: one launch less per convolution + BatchNorm layer, the same bits).  While it is on, a filter gradient is
# valid once the backward pass has returned (an end-of-pass callback sums whatever is still pending) — as with WGRAD_STREAM.  Off by
# default: a multi-rank reducer's hooks read the gradients as they are accumulated.
DEFER_WGRAD_REDUCE = False
_PENDING_REDUCE = {}          # stream handle -> (part, dw, n, splits, stream, filter)
# Multi-rank runs: a deferred filter gradient never passes through autograd's AccumulateGrad, so the reducer's post-accumulate hook
# does not see it; the reducer registers this callable instead (called with the filter tensor once the launch that carries the sum
# of its splits has been enqueued — from then on the gradient is stream-ordered like any other).
DEFERRED_GRAD_HOOK = None
# ... and the filters (by data pointer) whose gradient was handed over directly in the current pass: autograd still runs their
# AccumulateGrad node with an undefined gradient, and torch fires the post-accumulate hooks of such a node — right after
# Conv2d.backward, BEFORE the launch that carries the sum of the splits is enqueued.  The reducer must not count (or gather) on that
# call; it counts the announcement.  Cleared by begin_step().
DEFERRED_FILTERS = set()


def _deferred_grad_done(w):
    if DEFERRED_GRAD_HOOK is not None and w is not None:
        DEFERRED_GRAD_HOOK(w)


def _flush_pending_reduce(stream_handle=None):
    keys = [stream_handle] if stream_handle is not None else list(_PENDING_REDUCE)
    for k in keys:
        rec = _PENDING_REDUCE.pop(k, None)
        if rec is not None:
            part, dw, n, splits, stream, w = rec
            _l.check(_l.lib().sqd_split_reduce(_ptr(part), _ptr(dw), n, splits, ctypes.c_void_p(k)), "split_reduce")
            cur = torch.cuda.current_stream()
            if stream is not None and stream.cuda_stream != cur.cuda_stream:
                # the sum ran on the stream its partials were produced on; whoever reads the gradient next does so on the caller's
                # stream (inside a capture an unjoined branch would otherwise be left behind)
                cur.wait_stream(stream)
            _deferred_grad_done(w)


def _end_of_backward_reduce():
    _flush_pending_reduce()


def drop_pending_reduce():
    """a backward pass that raised leaves its pending sums behind: the next step must not flush them into a stale gradient"""
    _PENDING_REDUCE.clear()


def _set_pending_reduce(part, dw, n, splits, w=None):
    h = torch.cuda.current_stream().cuda_stream
    _flush_pending_reduce(h)                    # two convolutions in a row without a BatchNorm between them: the first sum runs now
    _PENDING_REDUCE[h] = (part, dw, n, splits, torch.cuda.current_stream(), w)
    torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward_reduce)      # (idempotent: nothing pending, nothing launched)


def _take_pending_reduce():
    return _PENDING_REDUCE.pop(torch.cuda.current_stream().cuda_stream, None)
WGRAD_BATCH = 1              # convolutions per cross-stream dependency of the side-stream weight gradients
_PENDING_WGRAD = {}          # raw stream handle -> (stream the operands are produced on, [(launch, tensors)])


_WEIGHT_USES = {}            # id(weight storage) -> forward uses in the current step (reset by begin_step)


def begin_step():
    """Training loops call this before every forward pass (the side-stream weight gradients need per-step use counts; the operand
    scales of the two-term fp16 plans live in a per-step pool, and the filters' are refreshed here).  -> True when the filters' records were
    refreshed by a pass of this call (a captured step then carries that pass), False when the optimiser's launch had left them valid."""
    _WEIGHT_USES.clear()
    DEFERRED_FILTERS.clear()
    if _AM["buf"] is not None:
        _AM["buf"].zero_()                       # (a memset node of the captured step)
        _AM["n"] = 0
        _AM["epoch"] += 1
        # pools an over-long step retired (_amax_new) are dropped one step late: side-stream weight gradients of the step that retired
        # them may still be reading
        _AM["retired_prev"], _AM["retired"] = _AM.get("retired", []), []
    if AMAX_ON and _WAM["buf"] is not None:
        # (no pass when the optimiser's kernel left every record behind — wam_records_written — and nothing has written a filter through torch since)
        if not _WAM["index"] or filter_records_stale() or not FUSE_ADAM_AMAX:
            _wam_refresh()
            return True
    return False


def flush_wgrads():
    """launch the queued weight-gradient kernels on WGRAD_STREAM (after everything their producer streams hold so far)"""
    side = WGRAD_STREAM
    for key in list(_PENDING_WGRAD):
        src, q = _PENDING_WGRAD.pop(key)
        if not q:
            continue
        side.wait_stream(src)
        cur = torch.cuda.current_stream()
        torch.cuda.set_stream(side)
        try:
            for launch, tensors in q:
                launch()
                for t in tensors:
                    if t is not None:
                        t.record_stream(side)
        finally:
            torch.cuda.set_stream(cur)


def join_wgrad_stream():
    if WGRAD_STREAM is not None:
        flush_wgrads()
        torch.cuda.current_stream().wait_stream(WGRAD_STREAM)


def _scaled_plan(kind, geom):
    """does the registered plan of this geometry run on two-term fp16 operands (and so need the operands' max |.|)?"""
    if kind == "wgrad":
        N, H, W, C, K, R, S, stride, pad, Ho, Wo = geom
        v = CHOSEN_PLANS.get(("wgrad", N, Ho, Wo, C, K, R, S))
        return v is not None and (v[0] & 15) == 7
    v = CHOSEN_PLANS.get((kind,) + tuple(geom))
    return v is not None and bool(v[3] & 4096)


def _scales_wanted(kind, geom, tune_key):
    """operand scales are fetched when the plan needs them, or when the plans of this geometry are about to be timed (the two-term
    plans are among the candidates).  Convolutions over a handful of rows (the Linear layers of the bins regressor) stay unscaled."""
    if _scaled_plan(kind, geom):
        return True
    if not AMAX_ON or _l.lib().sqd_conv_precision() != 0 or geom[0] * geom[9] * geom[10] < 256:
        return False
    return TUNE_CONV and TUNE_SPACE["f16x2"] and tune_key not in _TUNED and not torch.cuda.is_current_stream_capturing()


class Conv2d(torch.autograd.Function):
    """nn.Conv2d (square stride / padding, dilation 1, groups 1) on the matrix cores, channels-last.
    forward(x [N,C,H,W], weight [K,C,R,S], bias [K] | None, stride, pad, act, skip) -> y [N,K,Ho,Wo]  (skip: -> (y, x'))

    With skip=True the node also returns its input as a second output x' (same storage).  A consumer that would have read
    x a second time (the residual branch, a down-sample convolution) reads x' instead; autograd then hands both
    gradients to this node, and the data gradient adds the second one in its epilogue (sqd_conv_dgrad's addend) instead
    of ATen running a separate 3-pass add over the activation.

    Plans on two-term fp16 operands (sqd.h section 10b) take the operands' max |.| from the tensors' tags (amax_of); every launch
    records max |output| for the next convolution of a chain."""

    @staticmethod
    def forward(ctx, x, weight, bias, stride, pad, act, skip=False, out_hw=None, stats=None):
        ctx.set_materialize_grads(False)                 # an unused output arrives as None in backward, not as a zero tensor
        _require(x, "Conv2d input")
        x_in = x
        x, w = _cl(x), _cl(weight)
        N, C, H, W = x.shape
        K, _, R, S = w.shape
        Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
        if out_hw is not None:                           # only the top-left out_hw outputs (see conv2d_stem_s2d)
            assert out_hw[0] <= Ho and out_hw[1] <= Wo
            Ho, Wo = out_hw
        y = torch.empty((N, K, Ho, Wo), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
        geom = (N, H, W, C, K, R, S, stride, pad, Ho, Wo)
        L = _l.lib()
        ax, aw = _amax_get(x_in), None
        if _scales_wanted("fwd", geom, (0,) + geom):
            if ax is None:
                ax = amax_of(x, "conv forward: input")
            aw = amax_of_weight(weight)
        ay = _amax_out(x.device)

        def run(ws):
            return L.sqd_conv_fwd_scaled(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), _ptr(ws), _ptr(stats), _ptr(ax), _ptr(aw), _ptr(ay), N, H, W, C, K,
                                         R, S, stride, pad, Ho, Wo, ACT[act], _stream())
        if TUNE_CONV:
            _tune_conv(0, geom, run, aw is not None)
        _l.check(run(_conv_ws(0, geom, x.device)), "conv_fwd")
        _amax_tag(y, ay)
        ctx.save_for_backward(x, w, y if act is not None else None)
        # The weight gradient may run on a side stream only when nothing reads it before the optimiser / bucket gather joins
        # that stream: the filter must be a leaf (a regrouped stem filter feeds StemRegroup.backward at once) and used once
        # per step (autograd sums the gradients of a shared filter on the main stream as soon as the second one arrives).
        ctx.wkey = weight.data_ptr() if weight.is_leaf else None
        if ctx.wkey is not None:
            _WEIGHT_USES[ctx.wkey] = _WEIGHT_USES.get(ctx.wkey, 0) + 1
        ctx.geom = geom
        ctx.has_bias, ctx.act = bias is not None, act
        if bias is not None and act is None:
            y._sqd_bias_geom = geom          # (read by a BatchNormAct node that takes y: its backward can deliver this node's bias gradient)
        ctx.bn_src = getattr(x_in, "_sqd_bn_src", None)  # x is the output of a training-mode BatchNormAct: see backward
        # what the backward pass may need of this step's scales (saved tensors come back as other Python objects: the tags would be lost)
        ctx.am = (ax, aw, _weight_source(weight), _AM["epoch"])
        if skip:
            return y, _amax_tag(x.view_as(x), ax)
        return y

    @staticmethod
    def backward(ctx, dy, g_skip=None):
        x, w, y = ctx.saved_tensors
        N, H, W, C, K, R, S, stride, pad, Ho, Wo = ctx.geom
        if dy is None:                                   # only the pass-through output was used
            return g_skip, None, None, None, None, None, None, None, None
        L = _l.lib()
        dy_in = dy
        dy = _cl(dy)
        if dy is not dy_in:
            _amax_tag(dy, _amax_get(dy_in))
        # column sums of dy from the pass that wrote it (the bias gradient without a pass of its own): only when this node has no activation
        # of its own (its gradient would come between dy and the sums)
        pre_db = _colsum_get(dy_in) if ctx.act is None else None
        g_skip = _cl(g_skip) if g_skip is not None else None
        if ctx.act is not None:                          # the epilogue's ReLU / LeakyReLU: dy * act'(y), one launch
            g = torch.empty_like(dy)
            ag = _amax_out(dy.device)
            _l.check(L.sqd_act_bwd_amax(_ptr(dy), _ptr(y), _ptr(g), dy.numel(), ACT[ctx.act], _ptr(ag), _stream()), "act_bwd")
            dy = _amax_tag(g, ag)
        sx, sw, wt, epoch = ctx.am
        if epoch != _AM["epoch"]:                        # (a backward pass of another step: the pool has been cleared since)
            sx = sw = None
        scales = {}

        def a_dy():
            if "dy" not in scales:
                scales["dy"] = amax_of(dy, "conv backward: output gradient")
            return scales["dy"]

        def a_x():
            if "x" not in scales:
                scales["x"] = sx if sx is not None else amax_of(x, "conv backward: saved input")
            return scales["x"]

        def a_w():
            if "w" not in scales:
                scales["w"] = sw if sw is not None else amax_of_weight(wt) if wt is not None else amax_of(w, "conv backward: filter (not a leaf)")
            return scales["w"]
        dx = dw = db = None
        if ctx.needs_input_grad[1]:
            wkey = _wgrad_key(ctx.geom)
            w_sc = _scales_wanted("wgrad", ctx.geom, wkey)
            ady, axx = (a_dy(), a_x()) if w_sc else (None, None)
            if TUNE_CONV:
                dw = torch.empty((K, C, R, S), device=dy.device, dtype=torch.float32, memory_format=torch.channels_last)
                db = torch.empty(K, device=dy.device, dtype=torch.float32) if ctx.has_bias else None
                _tune_wgrad(ctx.geom, ctx.has_bias,
                            lambda part: L.sqd_conv_wgrad_scaled(_ptr(dy), _ptr(x), _ptr(dw), _ptr(db), _ptr(part), _ptr(ady), _ptr(axx), N, H, W, C, K,
                                                                 R, S, stride, pad, Ho, Wo, None, _stream()),
                            (lambda _: _wgrad_transposed(dy, x, dw, db, ctx.geom, ady, axx)) if wgrad_transposed_applies(ctx.geom) else None, w_sc)
                if not w_sc and _scaled_plan("wgrad", ctx.geom):      # (cannot happen: a two-term plan is only timed with scales at hand)
                    ady, axx = a_dy(), a_x()
            dw = torch.empty((K, C, R, S), device=dy.device, dtype=torch.float32, memory_format=torch.channels_last)
            db = torch.empty(K, device=dy.device, dtype=torch.float32) if ctx.has_bias else None
            db_k = db                                    # what the weight-gradient kernels are asked to fill
            _WGRAD_SETTLED.add(_wgrad_key(ctx.geom))
            if db is not None and pre_db is not None and pre_db.numel() == K:
                # the pass that wrote dy also summed its columns (GELU / layer-scale backward): that IS dbias
                db, db_k = pre_db, None
            # (the plan table keys on the OUTPUT geometry: a strided or padded 1x1 that shares it with a stride-1 layer — or a plan file —
            #  must not take the transposed product, which reads x as [N*Ho*Wo][C] rows)
            transposed = CHOSEN_PLANS.get(("wgrad", N, Ho, Wo, C, K, R, S), (0,))[0] == WGRAD_TRANSPOSED and wgrad_transposed_applies(ctx.geom)
            if transposed:
                part = None
                gT = (1, 1, K, N * Ho * Wo, C, 1, 1, 1, 0, 1, K)
                if ady is None and _scales_wanted("fwd", gT, (0,) + gT):
                    ady, axx = a_dy(), a_x()

                def launch(dy=dy, x=x, dw=dw, db=db_k, geom=ctx.geom, ady=ady, axx=axx):
                    _wgrad_transposed(dy, x, dw, db, geom, ady, axx)
            else:
                pf, splits = _wgrad_part_floats(ctx.geom)
                extra = max((N * Ho * Wo + 1023) // 1024, splits) * K if ctx.has_bias else 0
                part = torch.empty(pf + extra, device=dy.device, dtype=torch.float32)

                def launch(dy=dy, x=x, dw=dw, db=db_k, part=part, ady=ady, axx=axx):
                    _l.check(L.sqd_conv_wgrad_scaled(_ptr(dy), _ptr(x), _ptr(dw), _ptr(db), _ptr(part), _ptr(ady), _ptr(axx), N, H, W, C, K, R, S,
                                                     stride, pad, Ho, Wo, None, _stream()), "conv_wgrad")
            if not transposed and DEFER_WGRAD_REDUCE and WGRAD_STREAM is None and ctx.wkey is not None and _WEIGHT_USES.get(ctx.wkey, 2) == 1 and \
                    ctx.bn_src is not None and w.grad is None and w.is_leaf and w.requires_grad and w.data_ptr() == ctx.wkey and \
                    not w._backward_hooks and (DEFERRED_GRAD_HOOK is not None or not getattr(w, "_post_accumulate_grad_hooks", None)):
                # the partial filter gradients now; x is the output of a training-mode BatchNorm, whose backward is the next node of this
                # stream: its finalize launch carries the sum (the end-of-pass callback is only the safety net).  The tensor is handed to
                # the parameter directly (autograd gets None for it: an AccumulateGrad that decided to copy would copy it before it is
                # written): valid when the backward pass has returned.  Only for the original leaf filter without tensor hooks (a
                # channels-last copy would receive the gradient instead of the parameter; hooks would be skipped) — post-accumulate
                # hooks only when their owner registered DEFERRED_GRAD_HOOK (the multi-rank reducer).
                sp = ctypes.c_int(0)
                _l.check(L.sqd_conv_wgrad_scaled(_ptr(dy), _ptr(x), _ptr(dw), _ptr(db_k), _ptr(part), _ptr(ady), _ptr(axx), N, H, W, C, K, R, S, stride,
                                                 pad, Ho, Wo, ctypes.byref(sp), _stream()), "conv_wgrad_partials")
                w.grad = dw
                DEFERRED_FILTERS.add(w.data_ptr())
                _set_pending_reduce(part, dw, K * R * S * C, sp.value, w)
                dw = None
            elif WGRAD_STREAM is None or ctx.wkey is None or _WEIGHT_USES.get(ctx.wkey, 2) != 1:
                launch()
            else:
                # the weight gradient has no consumer before the optimiser: it is queued and runs on its own stream, next to
                # the data gradients that follow (two kernels that each leave CUs idle in their ramp and tail fill the chip
                # together).  Queued in batches: one cross-stream dependency per WGRAD_BATCH convolutions.
                cur = torch.cuda.current_stream()
                q = _PENDING_WGRAD.setdefault(cur.cuda_stream, (cur, []))[1]
                q.append((launch, (dy, x, part, dw, db)))
                if len(q) >= WGRAD_BATCH:
                    flush_wgrads()
        if ctx.needs_input_grad[0]:
            dx = torch.empty((N, C, H, W), device=dy.device, dtype=torch.float32, memory_format=torch.channels_last)
            d_sc = _scales_wanted("dgrad", ctx.geom, (1,) + tuple(ctx.geom))
            ady, aww = (a_dy(), a_w()) if d_sc else (None, None)
            adx = _amax_out(dy.device)
            src = ctx.bn_src

            def run_d(ws, stats=None):
                bn = src if stats is not None else {}
                return L.sqd_conv_dgrad_scaled(_ptr(dy), _ptr(w), _ptr(g_skip), _ptr(dx), _ptr(ws), _ptr(bn.get("x")), _ptr(bn.get("mask")),
                                               _ptr(bn.get("mean")), _ptr(bn.get("rstd")), bn.get("code", 0), _ptr(stats), _ptr(ady), _ptr(aww), _ptr(adx),
                                               N, H, W, C, K, R, S, stride, pad, Ho, Wo, _stream())
            if TUNE_CONV:
                _tune_conv(1, ctx.geom, run_d, d_sc)
            ws = _conv_ws(1, ctx.geom, dy.device)
            rows = conv_dgrad_stats_rows(ctx.geom) if src is not None and src.get("x") is not None else 0
            if rows > 0:
                # dx (= data gradient + the second path's gradient) is the complete gradient of a BatchNorm's output: the epilogue
                # also writes that BatchNorm backward's per-channel partial sums, and the node finds them through `src`
                stats = torch.empty(rows * C * 2, device=dy.device, dtype=torch.float32)
                _l.check(run_d(ws, stats), "conv_dgrad_bn")
                src.update(dx=dx, part=stats, rows=rows)
            else:
                _l.check(run_d(ws), "conv_dgrad")
            _amax_tag(dx, adx)
        elif g_skip is not None:
            dx = g_skip
        return dx, dw, db, None, None, None, None, None, None


def conv_dgrad_stats_rows(geom):
    """rows of BatchNorm-backward partials the data gradient of this geometry writes under its current plan (0: none)"""
    key = ("sd",) + tuple(geom)
    n = _PLAN_CACHE.get(key)
    if n is None:
        n = _PLAN_CACHE[key] = _l.lib().sqd_conv_dgrad_stats_rows(*geom)
    return n


def conv_stats_rows(geom):
    """rows of BatchNorm partials the forward convolution of this geometry writes under its current plan (0: none)"""
    key = ("s",) + tuple(geom)
    n = _PLAN_CACHE.get(key)
    if n is None:
        n = _PLAN_CACHE[key] = _l.lib().sqd_conv_fwd_stats_rows(*geom)
    return n


def stem_s2d_supported(conv, x):
    """7x7 / stride 2 / pad 3 convolutions on few input channels (the ResNet and PoseCNN stems)."""
    return (conv.kernel_size == (7, 7) and conv.stride == (2, 2) and conv.padding == (3, 3) and conv.dilation == (1, 1)
            and conv.groups == 1 and conv.in_channels % 4 != 0 and conv.out_channels % 4 == 0
            and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0)


def conv2d_stem_s2d(x, conv, act=None, stats=None):
    """A 7x7 stride-2 convolution on C (3 or 6) channels == a 4x4 stride-1 convolution on the space-to-depth(2) image with
    4C channels (padded to a multiple of 16) and the filter regrouped the same way: tap u = 2r' + dy - 1 of the 7 (u = -1 and
    u = 7 are zero taps).  That shape runs on the implicit-GEMM kernels (57-77 % of the multiplies are real), so the stems
    need no vendor convolution; the image needs no gradient, the filter gradient flows back through the regrouping."""
    N, C, H, W = x.shape
    K = conv.out_channels
    Cp = (4 * C + 15) // 16 * 16
    x = _cl(x.detach())
    xs = torch.empty((N, Cp, H // 2, W // 2), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
    _l.check(_l.lib().sqd_space_to_depth2(_ptr(x), _ptr(xs), N, H, W, C, Cp, _stream()), "space_to_depth2")   # channel c*4 + dy*2 + dx
    # (regrouped on every call — one tiny kernel: the optimiser updates the weights through raw pointers, so nothing on the
    # Python side could tell a cached copy that it is stale, and a copy cached before a graph capture would be frozen into it)
    w = StemRegroup.apply(conv.weight, Cp)
    w._sqd_w_src = conv.weight                   # (regrouped values + zero taps: max |w| is the parameter's)
    return Conv2d.apply(xs, w, conv.bias, 1, 2, act, False, (H // 2, W // 2), stats)


def conv2d_stem_s2d_planar(pairs, conv, act=None, stats=None, affine=(0.0, 1.0)):
    """conv2d_stem_s2d on frames that are still planar (NCHW, dense): pairs = [(x0, x1 | None), ...], each x [B,C,H,W]; the batch of
    the convolution is [B * len(pairs)] with row b * len(pairs) + i = pair i of sample b, its channels the concatenation (x0, x1),
    every value (v - affine[0]) / affine[1].  One sqd_space_to_depth2_planar launch per pair replaces the layout conversion, the
    normalisation (two element-wise passes) and the torch.cat / slice copies of the frame staging."""
    x0 = pairs[0][0]
    B, C0, H, W = x0.shape
    C1 = 0 if pairs[0][1] is None else pairs[0][1].shape[1]
    S, K = len(pairs), conv.out_channels
    Cp = (4 * (C0 + C1) + 15) // 16 * 16
    xs = torch.empty((B * S, Cp, H // 2, W // 2), device=x0.device, dtype=torch.float32, memory_format=torch.channels_last)
    per = (H // 2) * (W // 2) * Cp
    axs = _amax_out(x0.device)                   # one scalar for the whole batch: every launch below raises it
    for i, (a, b) in enumerate(pairs):
        a = a.detach()
        b = None if b is None else b.detach()
        if not (a.is_contiguous() and (b is None or b.is_contiguous()) and a.dtype == torch.float32):
            raise RuntimeError("sqd: conv2d_stem_s2d_planar needs dense NCHW float32 frames")
        _l.check(_l.lib().sqd_space_to_depth2_planar_amax(_ptr(a), _ptr(b), ctypes.c_void_p(xs.data_ptr() + 4 * per * i), B, H, W, C0, C1, Cp,
                                                          per * S, float(affine[0]), float(affine[1]), _ptr(axs), _stream()), "space_to_depth2_planar")
    _amax_tag(xs, axs)
    w = StemRegroup.apply(conv.weight, Cp)
    w._sqd_w_src = conv.weight
    return Conv2d.apply(xs, w, conv.bias, 1, 2, act, False, (H // 2, W // 2), stats)


def stem_s2d_planar_supported(conv, x):
    """a 7x7 / stride 2 / pad 3 stem on a dense NCHW float32 frame (see stem_s2d_supported)"""
    return stem_s2d_supported(conv, x) and x.is_contiguous() and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] > 1


class StemRegroup(torch.autograd.Function):
    """w [K,C,7,7] -> the 4x4 filter on the space-to-depth channels, [K,Cp,4,4] channels-last (one kernel each way)."""

    @staticmethod
    def forward(ctx, w, Cp):
        K, C = w.shape[0], w.shape[1]
        cl = w.is_contiguous(memory_format=torch.channels_last) and not w.is_contiguous()
        wc = w if cl else w.contiguous()                 # (a channels-last parameter is read in place: no layout copy)
        ws = torch.empty((K, Cp, 4, 4), device=w.device, dtype=torch.float32, memory_format=torch.channels_last)
        _l.check(_l.lib().sqd_stem_regroup_ex(_ptr(wc), _ptr(ws), K, C, Cp, 0, 1 if cl else 0, _stream()), "stem_regroup")
        ctx.dims = (K, C, Cp, cl)
        return ws

    @staticmethod
    def backward(ctx, g):
        K, C, Cp, cl = ctx.dims
        g = _cl(g)
        gw = torch.empty((K, C, 7, 7), device=g.device, dtype=torch.float32, memory_format=torch.channels_last if cl else torch.contiguous_format)
        _l.check(_l.lib().sqd_stem_regroup_ex(_ptr(g), _ptr(gw), K, C, Cp, 1, 1 if cl else 0, _stream()), "stem_regroup_adjoint")
        return gw, None


_STEM3_INDEX = {}


def conv2d_stem3_same_s2d(x, conv):
    """EfficientNet's stem — a 3x3 / stride 2 convolution with TensorFlow "SAME" padding on the 3-channel frame (reference
    networks/base_encoder.py:41,94: tf_efficientnet_b5_ap.conv_stem) — as a 3x3 / stride 1 / pad 1 convolution on the
    space-to-depth(2) image: for even H, W "SAME" pads one row below and one column right only, so output row i reads input rows
    2i, 2i+1, 2i+2 = (block i, dy 0), (block i, dy 1), (block i+1, dy 0); the filter is scattered accordingly into [K, 16, 3, 3]
    (block offsets -1 and the (block i+1, dy 1) taps are zero).  Runs on the implicit-GEMM / input-patch kernels like every
    other convolution; the frame needs no gradient, the filter gradient comes back through the scatter's adjoint (a gather)."""
    N, C, H, W = x.shape
    K = conv.out_channels
    if not (conv.kernel_size == (3, 3) and conv.stride == (2, 2) and H % 2 == 0 and W % 2 == 0 and conv.groups == 1 and conv.bias is None):
        raise RuntimeError("sqd: the SAME-padded stem kernel takes a bias-free 3x3 / stride 2 convolution on an even-sized frame; got "
                           "kernel %s stride %s on %dx%d" % (tuple(conv.kernel_size), tuple(conv.stride), H, W))
    Cp = (4 * C + 15) // 16 * 16
    xc = _cl(x.detach())
    xs = torch.empty((N, Cp, H // 2, W // 2), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
    _l.check(_l.lib().sqd_space_to_depth2(_ptr(xc), _ptr(xs), N, H, W, C, Cp, _stream()), "space_to_depth2")   # channel c*4 + dy*2 + dx
    key = (C, Cp, x.device)
    idx = _STEM3_INDEX.get(key)
    if idx is None:
        # position of w[k, c, r, s] inside the KRSC memory of the [K, Cp, 3, 3] filter: ((R' * 3) + S') * Cp + c*4 + dy*2 + dx
        pos = [(((r // 2 + 1) * 3 + (s_ // 2 + 1)) * Cp + c * 4 + (r % 2) * 2 + (s_ % 2)) for c in range(C) for r in range(3) for s_ in range(3)]
        idx = _STEM3_INDEX[key] = torch.tensor(pos, device=x.device, dtype=torch.int64)
    flat = torch.zeros((K, 9 * Cp), device=x.device, dtype=torch.float32).index_copy(1, idx, conv.weight.reshape(K, C * 9))
    ws = flat.view(K, 3, 3, Cp).permute(0, 3, 1, 2)          # logical [K, Cp, 3, 3], channels-last memory
    ws._sqd_w_src = conv.weight
    return Conv2d.apply(xs, ws, None, 1, 1, None, False, (H // 2, W // 2), None)


def conv2d_native(x, conv, act=None, skip=False, stats=None):
    s, p = conv.stride, conv.padding
    if s[0] != s[1] or p[0] != p[1] or conv.dilation != (1, 1) or conv.groups != 1:
        raise RuntimeError("sqd: native conv handles square stride/padding, dilation 1, groups 1")
    return Conv2d.apply(x, conv.weight, conv.bias, s[0], p[0], act, skip, None, stats)


def conv_out_geom(x, conv, s2d=False):
    """(N,H,W,C,K,R,S,stride,pad,Ho,Wo) of the launch conv2d_native / conv2d_stem_s2d makes for this module and input"""
    N, C, H, W = x.shape
    K = conv.out_channels
    if s2d:
        Cp = (4 * C + 15) // 16 * 16
        return (N, H // 2, W // 2, Cp, K, 4, 4, 1, 2, H // 2, W // 2)
    R, S, st, pd = conv.kernel_size[0], conv.kernel_size[1], conv.stride[0], conv.padding[0]
    return (N, H, W, C, K, R, S, st, pd, (H + 2 * pd - R) // st + 1, (W + 2 * pd - S) // st + 1)


FUSE_BN_BWD_STATS = True      # BatchNorm-backward sums from the consuming convolution's data-gradient epilogue (tools may switch it off)
# ... and from the other passes that write a BatchNorm's whole gradient: "res" the main branch's element-wise pass (down-sample BatchNorms),
# "pool" the max-pool backward gather (the stem), "upcat" the adjoint of the decoder's up-sampling (tools/ab_bench.py switches them one by one)
# "bias": the column sums of a BatchNorm backward's dx = the bias gradient of the convolution in front of it, where its weight-gradient kernel leaves none
FUSE_BN_SIDE_SUMS = {"res": True, "pool": True, "upcat": True, "bias": True}
_DEFER_COUNTERS = False
_PENDING_COUNTERS = []


def defer_bn_counters(on):
    """Training loops that call flush_bn_counters() once per step set this: the ~60 num_batches_tracked += 1
    launches of a forward pass become one multi-tensor add."""
    global _DEFER_COUNTERS
    _DEFER_COUNTERS = bool(on)


def flush_bn_counters():
    if _PENDING_COUNTERS:
        torch._foreach_add_(_PENDING_COUNTERS, 1)
        _PENDING_COUNTERS.clear()


def batch_norm_act(x, bn, act, residual=None, pre_part=None, pre_rows=0, pool=False):
    """nn.BatchNorm2d module `bn` (parameters, running buffers, momentum, eps) applied through the fused
    kernels; keeps nn.BatchNorm2d's bookkeeping (num_batches_tracked).  pool=True (training, no residual): the output carries the
    per-image channel sums of itself (`_sqd_pool_part`) for the squeeze-and-excite gate that reads it next."""
    training = bn.training or bn.running_mean is None
    if training and bn.num_batches_tracked is not None:
        if _DEFER_COUNTERS:
            _PENDING_COUNTERS.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked.add_(1)
    shared = {} if training and FUSE_BN_BWD_STATS else None
    pool_part = None
    if pool and training and residual is None:
        B, C, H, W = x.shape
        pool_part = torch.empty(B, _l.lib().sqd_se_chunks(H * W), C, device=x.device, dtype=torch.float32)
    out = BatchNormAct.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, residual, training,
                             0.1 if bn.momentum is None else bn.momentum, bn.eps, act, pre_part if training else None,
                             pre_rows if training else 0, shared, pool_part)
    if shared:
        out._sqd_bn_src = shared          # read by the Conv2d node that takes `out` as its input
    if pool_part is not None:
        out._sqd_pool_part = pool_part    # read by the SqueezeExcite node that takes `out` as its input
    return out


# ---------------------------------------------------------------------------------------------------
# token-wise blocks of the patch-token TransformerEncoderLayer (csrc/vit.hip)
# ---------------------------------------------------------------------------------------------------
def _colsum_multi(segs):
    """segs: [(src [nrows, ncols] partials, dst, tr)] -> one launch of fixed-order column sums."""
    n = len(segs)
    PA, IA = ctypes.c_void_p * n, ctypes.c_int * n
    src = PA(*[t[0].data_ptr() for t in segs])
    dst = PA(*[t[1].data_ptr() for t in segs])
    nrows = IA(*[t[0].shape[0] for t in segs])
    ncols = IA(*[t[0].numel() // t[0].shape[0] for t in segs])
    tr = IA(*[t[2] for t in segs])
    _l.check(_l.lib().sqd_colsum_multi(src, dst, nrows, ncols, tr, n, _stream()), "colsum_multi")


def _addln_fwd(x, y, nparts, ybias, mask, gamma, beta, scale, eps):
    E = x.shape[-1]
    rows = x.numel() // E
    out, xhat = torch.empty_like(x), torch.empty_like(x)
    rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    _l.check(_l.lib().sqd_addln_fwd(_ptr(x), _ptr(y), nparts, _ptr(ybias), _ptr(mask), _ptr(gamma), _ptr(beta), _ptr(out), _ptr(xhat),
                                    _ptr(rstd), rows, E, float(scale), float(eps), _stream()), "addln_fwd")
    return out, xhat, rstd


def _addln_bwd(g, g_extra, nextra, xhat, rstd, mask, gamma, scale):
    """-> g_x, g_y, part [nblk, 2E] (partials of g_gamma | g_beta)"""
    E = xhat.shape[-1]
    rows = xhat.numel() // E
    L = _l.lib()
    gx, gy = torch.empty_like(xhat), torch.empty_like(xhat)
    part = torch.empty(L.sqd_addln_nblk(rows), 2 * E, device=g.device, dtype=torch.float32)
    _l.check(L.sqd_addln_bwd(_ptr(g), _ptr(g_extra), nextra, _ptr(xhat), _ptr(rstd), _ptr(mask), _ptr(gamma), _ptr(gx), _ptr(gy),
                             _ptr(part), rows, E, float(scale), _stream()), "addln_bwd")
    return gx, gy, part


def _ffn_fwd(x, W1, b1, W2, mask, scale):
    """-> ypart [G, rows, E] (partials over the hidden groups, without b2)"""
    E, Fh = x.shape[-1], W1.shape[0]
    rows = x.numel() // E
    L = _l.lib()
    ypart = torch.empty(L.sqd_ffn_groups(Fh), rows, E, device=x.device, dtype=torch.float32)
    _l.check(L.sqd_ffn_fwd(_ptr(x), _ptr(W1), _ptr(b1), _ptr(W2), _ptr(mask), _ptr(ypart), rows, E, Fh, float(scale), _stream()), "ffn_fwd")
    return ypart


def _ffn_bwd(x, gy, W1, b1, W2, mask, scale):
    """-> gxpart [G, rows, E], pW1 [T, F*E], pb1 [T, F], pW2T [T, F*E], pb2 [T, E]"""
    E, Fh = x.shape[-1], W1.shape[0]
    rows = x.numel() // E
    L = _l.lib()
    T, G = L.sqd_ffn_tiles(rows), L.sqd_ffn_groups(Fh)
    buf = torch.empty(G * rows * E + T * (2 * Fh * E + Fh + E), device=x.device, dtype=torch.float32)
    sizes = [G * rows * E, T * Fh * E, T * Fh, T * Fh * E, T * E]
    gxpart, pW1, pb1, pW2T, pb2 = torch.split(buf, sizes)
    _l.check(L.sqd_ffn_bwd(_ptr(x), _ptr(gy), _ptr(W1), _ptr(b1), _ptr(W2), _ptr(mask), _ptr(gxpart), _ptr(pW1), _ptr(pb1), _ptr(pW2T),
                           _ptr(pb2), rows, E, Fh, float(scale), _stream()), "ffn_bwd")
    return gxpart.view(G, rows, E), pW1.view(T, Fh * E), pb1.view(T, Fh), pW2T.view(T, Fh * E), pb2.view(T, E)


class AddDropLayerNorm(torch.autograd.Function):
    """LayerNorm(x + dropout(y)) — norm1(x + dropout1(sa)) / norm2(x + dropout2(ff)) of the post-norm encoder layer
    (reference networks/depth_decoder_QTR.py:31-32 builds nn.TransformerEncoderLayer with its defaults).
    mask: uint8 keep-mask shaped like y, or None.  (Stand-alone node; the training path uses EncoderTail.)"""

    @staticmethod
    def forward(ctx, x, y, mask, gamma, beta, scale, eps):
        x, y = x.contiguous(), y.contiguous()
        _require(x, "addln x"), _require(y, "addln y")
        out, xhat, rstd = _addln_fwd(x, y, 1, None, mask, gamma, beta, scale, eps)
        ctx.save_for_backward(xhat, rstd, gamma)
        ctx.mask, ctx.scale = mask, float(scale)
        return out

    @staticmethod
    def backward(ctx, g):
        xhat, rstd, gamma = ctx.saved_tensors
        gx, gy, part = _addln_bwd(g.contiguous(), None, 0, xhat, rstd, ctx.mask, gamma, ctx.scale)
        gg = torch.empty(2, gamma.numel(), device=g.device, dtype=torch.float32)
        _colsum_multi([(part, gg, 0)])
        return gx, gy, None, gg[0], gg[1], None, None


class FeedForward(torch.autograd.Function):
    """linear2(dropout(relu(linear1(x)))) of the encoder layer; the [rows, F] hidden activations stay in registers
    (the backward recomputes them).  mask: uint8 keep-mask [rows, F] or None.  (Stand-alone node; the training path uses
    EncoderTail, which hands the group partials straight to the LayerNorm kernels.)"""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, mask, scale):
        x = x.contiguous()
        _require(x, "ffn x")
        y = _ffn_fwd(x, W1, b1, W2, mask, scale).sum(0).view_as(x) + b2
        ctx.save_for_backward(x, W1, b1, W2)
        ctx.mask, ctx.scale = mask, float(scale)
        return y

    @staticmethod
    def backward(ctx, g):
        x, W1, b1, W2 = ctx.saved_tensors
        gxpart, pW1, pb1, pW2T, pb2 = _ffn_bwd(x, g.contiguous(), W1, b1, W2, ctx.mask, ctx.scale)
        gW1, gb1, gW2 = torch.empty_like(W1), torch.empty_like(b1), torch.empty_like(W2)
        gb2 = torch.empty(x.shape[-1], device=g.device, dtype=torch.float32)
        _colsum_multi([(pW1, gW1, 0), (pW2T, gW2, x.shape[-1]), (pb1, gb1, 0), (pb2, gb2, 0)])
        return gxpart.sum(0).view_as(x), gW1, gb1, gW2, gb2, None, None


class EncoderTail(torch.autograd.Function):
    """Everything of the post-norm encoder layer after self-attention, as one autograd node:
        x1  = norm1(x + dropout1(sa))
        out = norm2(x1 + dropout2(linear2(dropout(relu(linear1(x1))))))
    3 launches forward, 4 backward (ATen: ~45)."""

    @staticmethod
    def forward(ctx, x, sa, m1, mf, m2, g1, be1, W1, b1, W2, b2, g2, be2, scale, eps1, eps2):
        x, sa = x.contiguous(), sa.contiguous()
        _require(x, "encoder tokens"), _require(sa, "attention output")
        x1, xhat1, rstd1 = _addln_fwd(x, sa, 1, None, m1, g1, be1, scale, eps1)
        ypart = _ffn_fwd(x1, W1, b1, W2, mf, scale)
        out, xhat2, rstd2 = _addln_fwd(x1, ypart, ypart.shape[0], b2, m2, g2, be2, scale, eps2)
        ctx.save_for_backward(xhat1, rstd1, x1, xhat2, rstd2, g1, W1, b1, W2, g2)
        ctx.masks, ctx.scale = (m1, mf, m2), float(scale)
        return out

    @staticmethod
    def backward(ctx, g):
        xhat1, rstd1, x1, xhat2, rstd2, g1, W1, b1, W2, g2 = ctx.saved_tensors
        m1, mf, m2 = ctx.masks
        E = x1.shape[-1]
        dz2, gy2, part2 = _addln_bwd(g.contiguous(), None, 0, xhat2, rstd2, m2, g2, ctx.scale)
        gxpart, pW1, pb1, pW2T, pb2 = _ffn_bwd(x1, gy2, W1, b1, W2, mf, ctx.scale)
        gx, gsa, part1 = _addln_bwd(dz2, gxpart, gxpart.shape[0], xhat1, rstd1, m1, g1, ctx.scale)
        gW1, gb1, gW2 = torch.empty_like(W1), torch.empty_like(b1), torch.empty_like(W2)
        small = torch.empty(5, E, device=g.device, dtype=torch.float32)     # g_b2 | g_gamma2, g_beta2 | g_gamma1, g_beta1
        _colsum_multi([(pW1, gW1, 0), (pW2T, gW2, E), (pb1, gb1, 0), (pb2, small[0], 0), (part2, small[1:3], 0), (part1, small[3:5], 0)])
        return gx, gsa, None, None, None, small[3], small[4], gW1, gb1, gW2, small[0], small[1], small[2], None, None, None


def _mha_fwd(x, Win, bin_, Wo, mask, S, B, H, scale):
    """-> ypart [H, rows, E], o_save, ml_save"""
    E = x.shape[-1]
    L = _l.lib()
    ypart = torch.empty(H, S * B, E, device=x.device, dtype=torch.float32)
    o_save = torch.empty(B * H * S * (E // H), device=x.device, dtype=torch.float32)
    ml = torch.empty(B * H * S * 2, device=x.device, dtype=torch.float32)
    _l.check(L.sqd_mha_fwd(_ptr(x), _ptr(Win), _ptr(bin_), _ptr(Wo), _ptr(mask), _ptr(ypart), _ptr(o_save), _ptr(ml), S, B, E, H,
                           float(scale), _stream()), "mha_fwd")
    return ypart, o_save, ml


def _mha_bwd(x, gsa, Win, bin_, Wo, mask, o_save, ml, S, B, H, scale):
    """-> gxpart [H, rows, E], pWin [B, 3E*E], pbin [B, 3E], pWo [B, E*E], pbo [B, E]"""
    E = x.shape[-1]
    sizes = [H * S * B * E, B * 3 * E * E, B * 3 * E, B * E * E, B * E]
    gxpart, pWin, pbin, pWo, pbo = torch.split(torch.empty(sum(sizes), device=x.device, dtype=torch.float32), sizes)
    _l.check(_l.lib().sqd_mha_bwd(_ptr(x), _ptr(gsa), _ptr(Win), _ptr(bin_), _ptr(Wo), _ptr(mask), _ptr(o_save), _ptr(ml), _ptr(gxpart),
                                  _ptr(pWin), _ptr(pbin), _ptr(pWo), _ptr(pbo), S, B, E, H, float(scale), _stream()), "mha_bwd")
    return gxpart.view(H, S * B, E), pWin.view(B, 3 * E * E), pbin.view(B, 3 * E), pWo.view(B, E * E), pbo.view(B, E)


class SelfAttention(torch.autograd.Function):
    """nn.MultiheadAttention(x, x, x, need_weights=False)[0] for tokens [S,B,E] as one kernel pair (stand-alone node for
    tests; the training path uses EncoderStack).  mask: uint8 keep-mask [B,H,S,SP] of the attention dropout, or None."""

    @staticmethod
    def forward(ctx, x, Win, bin_, Wo, bo, mask, H, scale):
        x = x.contiguous()
        _require(x, "attention tokens")
        S, B, E = x.shape
        ypart, o_save, ml = _mha_fwd(x, Win, bin_, Wo, mask, S, B, H, scale)
        ctx.save_for_backward(x, Win, bin_, Wo, o_save, ml)
        ctx.cfg = (mask, H, float(scale))
        return ypart.sum(0).view(S, B, E) + bo

    @staticmethod
    def backward(ctx, g):
        x, Win, bin_, Wo, o_save, ml = ctx.saved_tensors
        mask, H, scale = ctx.cfg
        S, B, E = x.shape
        gxpart, pWin, pbin, pWo, pbo = _mha_bwd(x, g.contiguous(), Win, bin_, Wo, mask, o_save, ml, S, B, H, scale)
        gWin, gbin, gWo = torch.empty_like(Win), torch.empty_like(bin_), torch.empty_like(Wo)
        gbo = torch.empty(E, device=g.device, dtype=torch.float32)
        _colsum_multi([(pWin, gWin, 0), (pbin, gbin, 0), (pWo, gWo, 0), (pbo, gbo, 0)])
        return gxpart.sum(0).view(S, B, E), gWin, gbin, gWo, gbo, None, None, None


def _layer_params(layer):
    a = layer.self_attn
    return [a.in_proj_weight, a.in_proj_bias, a.out_proj.weight, a.out_proj.bias, layer.norm1.weight, layer.norm1.bias,
            layer.linear1.weight, layer.linear1.bias, layer.linear2.weight, layer.linear2.bias, layer.norm2.weight, layer.norm2.bias]


class EncoderStack(torch.autograd.Function):
    """The whole post-norm encoder (all layers) as one autograd node: 4 launches per layer forward, 5 backward; sums over
    heads / hidden groups travel between the kernels as partials, so no stand-alone reduction or residual add is launched.
    cfg: {"H": heads, "eps": [(eps1, eps2)], "masks": [(attn, m1, mf, m2)] | None, "scale": 1/(1-p)}."""

    @staticmethod
    def forward(ctx, tokens, cfg, *params):
        x = tokens.contiguous()
        _require(x, "encoder tokens")
        S, B, E = x.shape
        H, scale = cfg["H"], cfg["scale"]
        saved = []
        for li in range(len(params) // 12):
            Win, bin_, Wo, bo, g1, be1, W1, b1, W2, b2, g2, be2 = params[12 * li:12 * li + 12]
            ma, m1, mf, m2 = cfg["masks"][li] if cfg["masks"] is not None else (None,) * 4
            eps1, eps2 = cfg["eps"][li]
            apart, o_save, ml = _mha_fwd(x, Win, bin_, Wo, ma, S, B, H, scale)
            x1, xhat1, rstd1 = _addln_fwd(x, apart, H, bo, m1, g1, be1, scale, eps1)
            ypart = _ffn_fwd(x1, W1, b1, W2, mf, scale)
            x2, xhat2, rstd2 = _addln_fwd(x1, ypart, ypart.shape[0], b2, m2, g2, be2, scale, eps2)
            saved += [x, o_save, ml, xhat1, rstd1, x1, xhat2, rstd2]
            x = x2
        ctx.save_for_backward(*saved, *params)
        ctx.cfg = cfg
        return x

    @staticmethod
    def backward(ctx, g):
        cfg = ctx.cfg
        H, scale = cfg["H"], cfg["scale"]
        nl = len(cfg["eps"])
        saved, params = ctx.saved_tensors[:8 * nl], ctx.saved_tensors[8 * nl:]
        S, B, E = saved[0].shape
        g, extra, nextra = g.contiguous(), None, 0
        grads = [None] * (12 * nl)
        for li in reversed(range(nl)):
            x, o_save, ml, xhat1, rstd1, x1, xhat2, rstd2 = saved[8 * li:8 * li + 8]
            Win, bin_, Wo, bo, g1, be1, W1, b1, W2, b2, g2, be2 = params[12 * li:12 * li + 12]
            ma, m1, mf, m2 = cfg["masks"][li] if cfg["masks"] is not None else (None,) * 4
            dz2, gy2, part2 = _addln_bwd(g, extra, nextra, xhat2, rstd2, m2, g2, scale)
            fpart, pW1, pb1, pW2T, pb2 = _ffn_bwd(x1, gy2, W1, b1, W2, mf, scale)
            dz1, gsa, part1 = _addln_bwd(dz2, fpart, fpart.shape[0], xhat1, rstd1, m1, g1, scale)
            apart, pWin, pbin, pWo, pbo = _mha_bwd(x, gsa, Win, bin_, Wo, ma, o_save, ml, S, B, H, scale)
            gWin, gbin, gWo = torch.empty_like(Win), torch.empty_like(bin_), torch.empty_like(Wo)
            gW1, gb1, gW2 = torch.empty_like(W1), torch.empty_like(b1), torch.empty_like(W2)
            small = torch.empty(6, E, device=g.device, dtype=torch.float32)   # g_bo | g_b2 | g_gamma1, g_beta1 | g_gamma2, g_beta2
            _colsum_multi([(pWin, gWin, 0), (pbin, gbin, 0), (pWo, gWo, 0), (pbo, small[0], 0), (pW1, gW1, 0), (pW2T, gW2, E),
                           (pb1, gb1, 0), (pb2, small[1], 0), (part1, small[2:4], 0), (part2, small[4:6], 0)])
            grads[12 * li:12 * li + 12] = [gWin, gbin, gWo, small[0], small[2], small[3], gW1, gb1, gW2, small[1], small[4], small[5]]
            g, extra, nextra = dz1, apart, H
        g_tokens = torch.empty((S, B, E), device=g.device, dtype=torch.float32)       # sum of the attention partials + g, one launch
        if (S * B * E) % 4 == 0 and extra.is_contiguous() and g.is_contiguous():
            _l.check(_l.lib().sqd_sum_parts(_ptr(extra), _ptr(g), _ptr(g_tokens), extra.shape[0], S * B * E, _stream()), "sum_parts")
        else:
            g_tokens = extra.sum(0).view(S, B, E).add_(g)
        return (g_tokens, None, *grads)


def _attention_native_ok(layer, S):
    a = layer.self_attn
    E = a.embed_dim
    return (a._qkv_same_embed_dim and a.in_proj_weight is not None and a.in_proj_bias is not None and a.bias_k is None and
            not a.add_zero_attn and a.out_proj.bias is not None and
            bool(_l.lib().sqd_mha_supported(S, E, a.num_heads)))


def encoder_supported(encoder):
    """nn.TransformerEncoder as the depth head builds it: post-norm layers, ReLU, no final norm, fp32 dense weights."""
    import torch.nn.functional as F
    if getattr(encoder, "norm", None) is not None:
        return False
    for layer in encoder.layers:
        W1, W2 = layer.linear1.weight, layer.linear2.weight
        if layer.norm_first or layer.activation is not F.relu or layer.linear1.bias is None or layer.linear2.bias is None:
            return False
        if not _l.lib().sqd_vit_supported(W1.shape[1], W1.shape[0]) or not (W1.is_contiguous() and W2.is_contiguous()):
            return False
        if layer.norm1.weight is None or layer.norm1.bias is None or getattr(layer.self_attn, "batch_first", False):
            return False
    return True


class TokensWithPos(torch.autograd.Function):
    """emb [B,E,h,w] (channels-last memory = [B][T][E]) + pos[:T] -> tokens [T,B,E]: reference networks/depth_decoder_QTR.py:49-51 in one launch
    each way (the gradient of the positional table arrives whole, zero rows included)"""

    @staticmethod
    def forward(ctx, emb, pos):
        B, E, h, w = emb.shape
        T = h * w
        embc = _cl(emb)
        out = torch.empty((T, B, E), device=emb.device, dtype=torch.float32)
        _l.check(_l.lib().sqd_tokens_pos_fwd(_ptr(embc), _ptr(pos), _ptr(out), B, T, E, _stream()), "tokens_pos_fwd")
        ctx.dims = (B, E, h, w, pos.shape[0])
        return out

    @staticmethod
    def backward(ctx, g):
        B, E, h, w, Tmax = ctx.dims
        g = g.contiguous()
        g_emb = torch.empty((B, E, h, w), device=g.device, dtype=torch.float32, memory_format=torch.channels_last)
        g_pos = torch.empty((Tmax, E), device=g.device, dtype=torch.float32)
        _l.check(_l.lib().sqd_tokens_pos_bwd(_ptr(g), _ptr(g_emb), _ptr(g_pos), B, h * w, E, Tmax, _stream()), "tokens_pos_bwd")
        return g_emb, g_pos


class FirstQueries(torch.autograd.Function):
    """tokens [T,B,E] -> tokens[:Q].permute(1, 0, 2) as a dense [B,Q,E] (reference networks/depth_decoder_QTR.py:52), one launch each way"""

    @staticmethod
    def forward(ctx, tokens, Q):
        T, B, E = tokens.shape
        tokens = tokens.contiguous()
        out = torch.empty((B, Q, E), device=tokens.device, dtype=torch.float32)
        _l.check(_l.lib().sqd_first_queries(_ptr(tokens), _ptr(out), T, B, Q, E, 0, _stream()), "first_queries")
        ctx.dims = (T, B, Q, E)
        return out

    @staticmethod
    def backward(ctx, g):
        T, B, Q, E = ctx.dims
        g = g.contiguous()
        gt = torch.empty((T, B, E), device=g.device, dtype=torch.float32)
        _l.check(_l.lib().sqd_first_queries(_ptr(g), _ptr(gt), T, B, Q, E, 1, _stream()), "first_queries_adjoint")
        return gt, None


def transformer_encoder_native(tokens, encoder):
    """tokens [S,B,E] through the encoder on the fused kernels (EncoderStack): S <= 512 tokens (the positional table holds 500; 256 at
    width 56 / 64), head dimension 4 | 8 | 14 | 16.  A token count or head dimension the attention kernel does not take raises — self-attention never
    runs through torch.  One bernoulli launch draws every dropout mask of the pass."""
    x = tokens.contiguous()
    S, B, E = x.shape
    rows = S * B
    layers = list(encoder.layers)
    H = layers[0].self_attn.num_heads
    if not all(_attention_native_ok(l, S) and l.self_attn.num_heads == H for l in layers):
        raise RuntimeError("sqd: self-attention over %d tokens of width %d with %d heads: the fused attention kernel takes up to 512 tokens "
                           "with head dimension 4 | 8 (width 16 / 32) and up to 256 tokens with head dimension 16 (width 64) or 14 (width 56: "
                           "the reference's Cityscapes args files); there is no ATen fallback" % (S, E, H))
    masks, scale = None, 1.0
    if encoder.training:
        ps = {float(p) for l in layers for p in (l.dropout1.p, l.dropout.p, l.dropout2.p)}
        ps |= {float(l.self_attn.dropout) for l in layers}
        if ps != {0.0}:
            if len(ps) != 1:
                raise RuntimeError("sqd: encoder layers with different dropout rates are not supported")
            p0 = ps.pop()
            SP = (S + 3) // 4 * 4
            per_layer = [(B * H * S * SP, rows * E, rows * l.linear1.weight.shape[0], rows * E) for l in layers]
            keep = torch.empty(sum(sum(t) for t in per_layer), device=x.device, dtype=torch.uint8).bernoulli_(1.0 - p0)
            masks = [torch.split(c, list(t)) for c, t in zip(torch.split(keep, [sum(t) for t in per_layer]), per_layer)]
            scale = 1.0 / (1.0 - p0)
    cfg = {"H": H, "eps": [(l.norm1.eps, l.norm2.eps) for l in layers], "masks": masks, "scale": scale}
    return EncoderStack.apply(x, cfg, *[p for l in layers for p in _layer_params(l)])

