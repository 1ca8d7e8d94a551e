"""Keras-semantics layers on the HIP kernels (NHWC fp32).

Each layer is a torch.nn.Module that owns its parameters in Keras layout and
name (`kernel` [R,S,Cin,Cout] / [in,out], `bias`, `gamma`, `beta`,
`moving_mean`, `moving_variance`) and whose forward/backward are launches of
libembnet_hip.so entry points through torch.autograd.Function — PyTorch is the
tape, the allocator and the optimizer, not the compute.

Semantics restated from the layers /root/reference/embedding_net/backbones.py
instantiates: Conv2D default stride 1, 'valid', bias,
glorot_uniform; 'same' pads extra on the bottom/right; BatchNormalization
momentum .99, eps 1e-3, biased batch variance in training; MaxPool2D() 2x2/2;
Dropout inverted scaling; l2(lambda) = lambda * sum(w^2) on kernels.
"""
import math
import os as _os

import torch
from torch import nn

from . import _lib
from ._lib import check, ptr, stream

_WS = {}


# Knob (off): run each conv's wgrad on a side stream beside its dgrad.  Measured on ResNet18/MI355X: +-0 % when
# joined right after the dgrad (both are MFMA-bound), +1.8 % when the join is deferred to the end of backward so
# wgrad overlaps the HBM-bound BN-backward kernels — not worth per-kernel timings that no longer mean anything
# (co-running doubles the wgrad kernel's own duration) and hand-made gradient hand-over (autograd clones dW on
# the main stream as soon as backward() returns it).
OVERLAP_WGRAD = _os.environ.get("EMBNET_OVERLAP_WGRAD", "0") == "1"
_SIDE = {}


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=device)
    return _SIDE[key]


_WS_RETIRED = []
_WS_POISON = _os.environ.get("EMBNET_WS_POISON", "0") == "1"


def workspace(nbytes, device):
    """Grow-only scratch buffer per device (kernels on one stream run in order, so reuse is safe).  A buffer that is outgrown
    is kept alive, never returned to the allocator: a captured HIP graph replays launches that hold its address."""
    n = max((int(nbytes) + 3) // 4, 256)
    key = (device.index, _lib.stream())
    buf = _WS.get(key)
    if buf is None or buf.numel() < n:
        if buf is not None:
            _WS_RETIRED.append(buf)
        buf = torch.empty(int(n * 1.25), dtype=torch.float32, device=device)
        _WS[key] = buf
    if _WS_POISON:             # debug aid (EMBNET_WS_POISON=1): a kernel that reads scratch it did not write sees NaN, not a neighbour's leftovers
        buf.fill_(float("nan"))
    return buf


# Deferred weight-gradient slab sums.  conv2d_wgrad splits its pixel reduction over workgroups into fp32 slabs and adds
# them up with a second, launch-sized kernel (7-10 us each: 21 per ResNet18 step, 7 per simple2 step).  While SLAB_DEFER[0] is
# set (TripletTrainer does, around backward) the split-K kernel writes into a slab buffer of the layer's own, the sum is
# queued, and flush_slab_reduces() adds ALL queued gradients up in one launch (embnet_slab_reduce_multi; same summation
# order, bit-identical) — before the optimizer, or before a gradient bucket's all-reduce.  Nothing else may read a queued
# dw before the flush; outside the trainer the flag is off and every wgrad finishes in place.
# ---- step context ------------------------------------------------------------------------------------------------------
# Fused kernels hand side-products from one autograd node to another (a BatchNorm backward's masked gradient to the conv in
# front of it, a data gradient's column sums to the BatchNorm behind it, planes of a gradient to the patch conv, split-K slabs
# to the one launch that sums them ...).  Every such hand-over lives in a StepContext — not in module globals — and the names
# this module has always used (BN_SUMS, RELU_DONE, GATE_PENDING, POOL_PENDING, DY_PLANES, _BN_FWD_STATS, _ACT_PLANES,
# _SLAB_PENDING) are views of the CURRENT context:
#   * TripletTrainer owns one context per trainer and runs every step inside it (train_step.py); two models trained or
#     evaluated in one process therefore never see each other's entries;
#   * bare autograd use (SiameseNet's loop, user scripts) runs in the default context;
#   * whatever a backward pass leaves unclaimed (a gradient that got a second contribution never meets its entry) is dropped
#     when THAT backward ends — the first entry made during a backward queues an end-of-backward callback on the autograd
#     engine — and counted in `context.unclaimed`, so entries neither pile up nor pin activation-sized tensors.
class StepContext:
    DICTS = ("bn_sums", "relu_done", "gate_pending", "pool_pending", "dy_planes", "dy_range", "bn_fwd_stats", "act_planes", "act_range")
    BACKWARD = ("bn_sums", "relu_done", "gate_pending", "pool_pending", "dy_planes", "dy_range")      # made and consumed inside one backward

    def __init__(self, name="default"):
        self.name = name
        for n in self.DICTS:
            setattr(self, n, {})
        self.slab_pending = []
        self.slab_seen = set()              # kernels (storage addresses) that already produced a weight gradient in the running backward
        self.unclaimed = {}                 # dict name -> entries dropped at the end of a backward, over the context's life
        self._armed = -1

    def leftovers(self):
        return {n: len(getattr(self, n)) for n in self.DICTS if getattr(self, n)}

    def end_of_backward(self):
        self._armed = -1
        self.slab_seen.clear()
        for n in self.BACKWARD:
            d = getattr(self, n)
            if d:
                self.unclaimed[n] = self.unclaimed.get(n, 0) + len(d)
                d.clear()

    def clear(self):
        for n in self.DICTS:
            getattr(self, n).clear()
        self.slab_pending.clear()
        self.slab_seen.clear()


_CONTEXTS = [StepContext("default")]


def current_context():
    return _CONTEXTS[-1]


class step_context:
    """`with step_context(ctx):` — the fused hand-overs of everything run inside go through `ctx`."""

    def __init__(self, ctx):
        self.ctx = ctx

    def __enter__(self):
        _CONTEXTS.append(self.ctx)
        return self.ctx

    def __exit__(self, *exc):
        _CONTEXTS.pop()
        return False


def _arm_cleanup(ctx):
    """Inside a backward pass: make sure ctx.end_of_backward runs when this pass ends."""
    try:
        tid = torch._C._current_graph_task_id()
    except AttributeError:
        return
    if tid != -1 and ctx._armed != tid:
        ctx._armed = tid
        torch.autograd.Variable._execution_engine.queue_callback(ctx.end_of_backward)


class _CtxDict:
    """A module-level name that reads and writes the CURRENT context's dict of that name."""

    def __init__(self, name):
        self._n = name

    def _d(self):
        return getattr(_CONTEXTS[-1], self._n)

    def __setitem__(self, k, v):
        ctx = _CONTEXTS[-1]
        if self._n in StepContext.BACKWARD:
            _arm_cleanup(ctx)
        getattr(ctx, self._n)[k] = v

    def __getitem__(self, k):
        return self._d()[k]

    def __contains__(self, k):
        return k in self._d()

    def __len__(self):
        return len(self._d())

    def __iter__(self):
        return iter(self._d())

    def __bool__(self):
        return bool(self._d())

    def pop(self, *a):
        return self._d().pop(*a)

    def get(self, *a):
        return self._d().get(*a)

    def clear(self):
        self._d().clear()

    def items(self):
        return self._d().items()


class _CtxList:
    def __init__(self, name):
        self._n = name

    def _l(self):
        return getattr(_CONTEXTS[-1], self._n)

    def append(self, v):
        self._l().append(v)

    def clear(self):
        self._l().clear()

    def __iter__(self):
        return iter(self._l())

    def __len__(self):
        return len(self._l())

    def __bool__(self):
        return bool(self._l())


SLAB_DEFER = [False]
SLAB_DEFER_ENABLED = [_os.environ.get("EMBNET_SLAB_DEFER", "1") != "0"]      # [False]: per-layer slab sums everywhere (A/B)
_SLAB_PENDING = _CtxList("slab_pending")          # (slab buffer, dw, elements, splits, kernel) of the current context
_SLAB_BUFS = {}             # kernel storage address -> slab buffer (scratch: any stale content is overwritten before use)


WGRAD_PLANES = [_os.environ.get("EMBNET_WGRAD_PLANES", "1") != "0"]   # [False]: every weight gradient on the gather loop (A/B)
_SLAB_RETIRED = []          # replaced slab buffers: a captured graph may still write into them, so they are never freed


def _slab_buffer(w, need, device):
    """The split-K slab buffer of kernel `w` (scratch, any stale content is overwritten before use).  A buffer that has to grow
    is RETIRED, not freed: a captured HIP graph replays launches that hold its address."""
    buf = _SLAB_BUFS.get(w.data_ptr())
    if buf is None or buf.numel() < need or buf.device != device:
        if buf is not None:
            _SLAB_RETIRED.append(buf)
        buf = _SLAB_BUFS[w.data_ptr()] = torch.empty(need, dtype=torch.float32, device=device)
    return buf


def wgrad_planes_ok(n, h, wd, c, r, s, k, stride, pt, pl, oh, ow):
    return WGRAD_PLANES[0] and bool(_lib.lib().embnet_conv2d_wgrad_planes_supported(n, h, wd, c, r, s, k, stride, pt, pl, oh, ow))


def conv_wgrad(lib, x, dz, dw, w, n, h, wd, c, r, s, k, stride, pt, pl, oh, ow, in_scale=None, in_shift=None, in_act=0,
               x_planes=None, dz_planes=None, dz_range=None, x_range=None):
    """dw[r,s,c,k] = weight gradient of a convolution, its slab sum deferred when SLAB_DEFER.  With the planes of BOTH operands
    (x_planes: kept by the forward patch conv, dz_planes: left by the BatchNormalization behind the conv) and a geometry
    embnet_conv2d_wgrad_planes_supported accepts, the planes kernel computes it (csrc/conv_wgrad_planes.hip) and the fp32
    tensors are not read; otherwise embnet_conv2d_wgrad_f32_ex — on three products when the range slots of BOTH operands are
    given (x_range: the BatchNormalization that wrote x, layers._range_of; dz_range: the range slot of dz, DY_RANGE).
    """
    xr, dr = (_rptr(x_range), _rptr(dz_range)) if (x_range is not None and dz_range is not None and in_scale is None) else (None, None)
    planes = x_planes is not None and dz_planes is not None and in_scale is None and \
        wgrad_planes_ok(n, h, wd, c, r, s, k, stride, pt, pl, oh, ow)
    # (an existing .grad means autograd will ADD dw to it at once — unless dw IS the parameter's gradient sink, which autograd never sees)
    # (only for a leaf kernel: the gradient of a derived one — the channel-padded kernel of an image conv — is consumed by the
    # next backward node at once)
    seen = current_context().slab_seen if SLAB_DEFER[0] else None
    if seen is not None and w.data_ptr() in seen:
        # the same kernel a second time in one backward (a shared layer: the two branches of a Siamese step).  Autograd holds the
        # first gradient in the node's input buffer and is about to ADD this one to it (the parameter's .grad stays None until both
        # have arrived): the first must be final by then and this one must be complete when it is returned — finish the queue if
        # the first is still in it, and never defer a second gradient.  (Round 5 deferred it: the add consumed an unreduced buffer
        # and the late slab sum then replaced the accumulated gradient — found by tests/test_siamese_trainer_gpu.py.)
        if any(e[4] is w for e in _SLAB_PENDING):
            flush_slab_reduces()
    elif SLAB_DEFER[0] and w.grad_fn is None and \
            (getattr(w, "grad", None) is None or (GRAD_SINKS and w.data_ptr() in GRAD_SINKS)):
        seen.add(w.data_ptr())
        if planes:
            splits = lib.embnet_conv2d_wgrad_planes_splits(n, h, wd, c, k)
        else:
            splits = lib.embnet_conv2d_wgrad_splits(n, c, r, s, k, oh, ow)
        if splits > 1:
            if planes:
                buf = _slab_buffer(w, lib.embnet_conv2d_wgrad_planes_workspace_bytes(n, h, wd, c, k) // 4, x.device)
                check(lib.embnet_conv2d_wgrad_planes_f32(ptr(x_planes), ptr(dz_planes), ptr(dw), ptr(buf), buf.numel() * 4,
                                                         n, h, wd, c, k, 0, stream()))
            else:
                buf = _slab_buffer(w, lib.embnet_conv2d_wgrad_workspace_bytes(n, c, r, s, k, oh, ow) // 4, x.device)
                check(lib.embnet_conv2d_wgrad_slabs_f32_ex(ptr(x), ptr(dz), ptr(dw), ptr(buf), buf.numel() * 4, n, h, wd, c, r, s, k,
                                                           stride, pt, pl, oh, ow, in_scale, in_shift, in_act, xr, dr, stream()))
            # (an alias of dw, not dw itself: autograd adopts a returned gradient as .grad only while nobody else holds that
            # tensor object — a second reference would make it clone the still-unreduced buffer)
            _SLAB_PENDING.append((buf, dw.detach(), r * s * c * k, splits, w))
            return
    if planes:
        ws = workspace(lib.embnet_conv2d_wgrad_planes_workspace_bytes(n, h, wd, c, k), x.device)
        check(lib.embnet_conv2d_wgrad_planes_f32(ptr(x_planes), ptr(dz_planes), ptr(dw), ptr(ws), ws.numel() * 4, n, h, wd, c, k,
                                                 1, stream()))
        return
    ws = workspace(lib.embnet_conv2d_wgrad_workspace_bytes(n, c, r, s, k, oh, ow), x.device)
    check(lib.embnet_conv2d_wgrad_f32_ex(ptr(x), ptr(dz), ptr(dw), ptr(ws), ws.numel() * 4, n, h, wd, c, r, s, k, stride, pt, pl,
                                         oh, ow, in_scale, in_shift, in_act, xr, dr, stream()))


def flush_slab_reduces():
    """One launch for the slab sums queued since the last flush (no-op when nothing is queued)."""
    if not _SLAB_PENDING:
        return
    import numpy as np
    rows = np.asarray([(b.data_ptr(), d.data_ptr(), n, sp) for b, d, n, sp, _ in _SLAB_PENDING], dtype=np.int64)
    check(_lib.lib().embnet_slab_reduce_multi(rows.ctypes.data, len(_SLAB_PENDING), stream()))
    for _, d, _, _, w in _SLAB_PENDING:                     # a gradient autograd copied instead of adopting: bring it up to date
        g = getattr(w, "grad", None)
        if g is not None and g.data_ptr() != d.data_ptr() and g.shape == d.shape and not _in_flat_buffer(w, d):
            g.copy_(d)
    _SLAB_PENDING.clear()


def _in_flat_buffer(w, d):
    s = GRAD_SINKS.get(w.data_ptr()) if GRAD_SINKS else None
    return s is not None and s[0].data_ptr() == d.data_ptr()


# Gradient sinks (data-parallel training, parallel.GradReducer.direct): parameter storage address -> (flat-buffer view,
# notify).  A weight-gradient kernel whose parameter has a sink writes its result straight into the view and returns
# no gradient to autograd, so no AccumulateGrad `add_` kernel runs for it; `notify` tells the reducer the slot is final.
# Only valid while every parameter takes part in ONE node per step (the fused TripletTrainer step) — the reducer arms it.
GRAD_SINKS = {}


def _sink(param):
    """-> (tensor to write the parameter's gradient into, notify-or-None)."""
    s = GRAD_SINKS.get(param.data_ptr()) if GRAD_SINKS else None
    if s is None:
        return torch.empty_like(param), None
    return s


def _done(out, notify):
    """What a backward returns for a parameter gradient it has produced in `out`."""
    if notify is None:
        return out
    notify()
    return None


# ---- pre-split operands of the patch convolution (csrc/conv_patch.hip) ------------------------------------------------
# A BatchNormalization whose output feeds a 3x3 stride-1 Conv2D writes that output ALSO as "planes" (the three bf16 pieces
# of every value, chunk-major) and hangs them on the tensor (`y._planes`); the Conv2D then runs the patch kernel.  In
# backward the BatchNormalization behind such a conv writes its input gradient also as planes and leaves them in DY_PLANES
# under the gradient's address, where the conv's backward picks them up for its data gradient.  Kernel planes are rebuilt
# when the weights changed: WEIGHT_EPOCH is bumped by KerasOptimizer.step(); other writers show in the tensor version.
# A DY_PLANES entry holds the gradient tensor beside its planes: while the entry lives that address cannot pass to another
# tensor, and autograd cannot accumulate a second gradient into it in place (it does so only into buffers nobody else holds).
# conv -> ReLU -> BatchNormalization blocks (the small backbones, reference backbones.py:44-68): a Conv2D with a fused ReLU
# and a bias tags its output (`y._relu_conv = (bias,)`); the training-mode BatchNormalization that reads it then runs its
# backward with the ReLU's backward folded in (embnet_bn_bwd_inrelu: dz = dx * [x > 0] and the bias gradient, one pass) and
# leaves (dz alias, dbias) in RELU_DONE under dz's address; the conv's backward finds its incoming gradient there and skips
# its own relu_bwd_colsum pass.  A conv whose output has a second consumer never finds the entry (autograd hands it the SUM,
# another tensor) and masks again — harmless, dz is already zero where the mask is.
# BatchNorm-backward column sums from the data gradient of the conv behind the BatchNormalization (csrc/conv.hip BnSums):
# BatchNormalization.forward tags its output with (x, stats, act); a stride-1 Conv2D on the gather kernels hands them to
# embnet_conv2d_dgrad_bnsums_f32 and leaves the partial sums here under its dx's address; the BatchNormalization backward that
# receives exactly that tensor starts at its finalize kernel.  (The entry keeps an alias of dx: autograd then never
# accumulates a second consumer's gradient into it in place, so a hit means dy IS that data gradient.)
FUSE_BN_SUMS = [__import__("os").environ.get("EMBNET_FUSE_BN_SUMS", "1") == "1"]
# ... and a stride-1 3x3 Conv2D on the patch kernel to embnet_conv2d_patch_bnsums_f32 (the same sums from that kernel's epilogue).
# OFF by default: the epilogue's reads of the BatchNormalization's input cost the data-gradient launches more (+0.5 ms per
# ResNet18 step) than the thirteen reduction launches they replace (0.3 ms): C2 10.75 -> 10.97 ms, C3 106.1 -> 107.1
# (profiles/r05_exp_patch_bnsums.txt).
PATCH_BN_SUMS = [__import__("os").environ.get("EMBNET_PATCH_BN_SUMS", "0") == "1"]
# the pooled branch's gradient (squeeze-and-excite) added inside the BatchNorm-backward passes instead of by a pass of its own
# the squeeze-and-excite multiply's backward (dy * gate) applied inside the BatchNorm backward too (MBConv opts in: lazy_scale)
# BatchNorm apply + DropConnect + Add of an MBConv tail as one pass, the drop factor applied inside the BatchNorm backward
FUSE_DROP_ADD = [__import__("os").environ.get("EMBNET_FUSE_DROP_ADD", "1") == "1"]
DW_EMIT_STATS = [__import__("os").environ.get("EMBNET_DW_EMIT_STATS", "1") == "1"]     # depthwise forward emits the next BN's statistics
DW_BN_SUMS = [__import__("os").environ.get("EMBNET_DW_BN_SUMS", "1") == "1"]   # ... and its stride-1 data gradient the previous BN's backward sums
SE_TWO_STAGE = [__import__("os").environ.get("EMBNET_SE_TWO_STAGE", "1") == "1"]  # ... and the activated tensor is never written (se_gate)
POOL_PENDING = _CtxDict("pool_pending")
SE_BN_SUMS = [__import__("os").environ.get("EMBNET_SE_BN_SUMS", "1") == "1"]      # ... and its reduction pass rides on the gate's gradient pass
FUSE_GATE_BN = [__import__("os").environ.get("EMBNET_FUSE_GATE_BN", "1") == "1"]
GATE_PENDING = _CtxDict("gate_pending")
FUSE_GAP_BN = [__import__("os").environ.get("EMBNET_FUSE_GAP_BN", "1") == "1"]
BN_SUMS = _CtxDict("bn_sums")
_BN_FWD_STATS = _CtxDict("bn_fwd_stats")

# a Dropout directly behind a BatchNormalization rides on the BatchNormalization's kernels (backbones.Seq); 0: separate passes
FUSE_DROPOUT_BN = [__import__("os").environ.get("EMBNET_FUSE_DROPOUT_BN", "1") == "1"]
FUSE_RELU_BN = [_os.environ.get("EMBNET_FUSE_RELU_BN", "1") != "0"]
# conv -> ReLU -> MaxPool (the 'simple' backbone): the same hand-over from MaxPool2D's backward (embnet_maxpool_relu_bwd_colsum)
FUSE_RELU_POOL = [_os.environ.get("EMBNET_FUSE_RELU_POOL", "1") != "0"]
RELU_DONE = _CtxDict("relu_done")
PATCH_CONV = [_os.environ.get("EMBNET_CONV_PATCH", "1") != "0"]      # [False]: every conv on the gather kernels (A/B)
# 1x1 convs marked `planes1x1` (backbones: the bottleneck's conv3, whose input is the thin tensor) run their FORWARD on the planes GEMM
# (csrc/conv_patch.hip conv1x1_planes_kernel).  OFF by default: back to back the kernel beats the ranged gather kernel on every
# stride-1 ResNet50 layer (profiles/r06_exp_conv1x1_planes.txt: 1024 -> 256 at 14x14 150 -> 81 us), but in the step the
# BatchNormalization in front must write the planes BESIDE the fp32 copy the gather weight gradient still reads: C3 86.8 -> 87.8 ms
# (profiles/r06_exp_conv1x1_step.txt).  It pays once a 1x1 planes weight gradient lets that tensor exist as planes only.
CONV1X1_PLANES = [_os.environ.get("EMBNET_CONV_1X1_PLANES", "0") != "0"]
# 1x1 convs whose operands both carry a range: the activation operand read as fp32 by LDS-DMA and split in the matrix waves
# (csrc/conv_patch.hip conv1x1_a32_kernel) instead of the gather loop — forward, and the stride-1 data gradients that do not carry
# BatchNorm-backward sums; output width >= 128.  Built, tested (tests/test_conv1x1_dma_gpu.py), OFF: 1.1 - 1.3 x the gather kernel
# back to back, 86.0 -> 86.2 ms in the C3 step (profiles/r06_exp_conv1x1_dma_step.txt) — ResNet50's 1x1 layers are HBM-bound there.
CONV1X1_DMA = [int(_os.environ.get("EMBNET_CONV_1X1_DMA_MODE", "0"))]
STEM_CONV = [_os.environ.get("EMBNET_STEM_CONV", "1") != "0"]        # [False]: the ResNet stem's forward on the gather kernel (A/B)


def _dma1x1_ok(n, h, wd, c, k, stride, oh, ow):
    return bool(CONV1X1_DMA[0]) and k >= 128 and bool(_lib.lib().embnet_conv2d_dma1x1_supported(n, h, wd, c, k, stride, oh, ow))
DY_PLANES = _CtxDict("dy_planes")
_ACT_PLANES = _CtxDict("act_planes")
_ACT_RANGE = _CtxDict("act_range")
WEIGHT_EPOCH = [0]


def _wplanes_entry(w):
    """The planes of kernel tensor `w` live on the tensor object itself (`w._embnet_wplanes`): an address-keyed cache would
    hand a new model the planes of a freed one whose storage it inherited."""
    e = getattr(w, "_embnet_wplanes", None)
    if e is None or e["ptr"] != w.data_ptr() or e["shape"] != tuple(w.shape):
        import numpy as np
        r, s, c, k = w.shape
        dev = w.device
        e = dict(ptr=w.data_ptr(), shape=tuple(w.shape), epoch=-1, version=-1,
                 fwd=torch.empty(3 * w.numel(), dtype=torch.int16, device=dev),
                 bwd=torch.empty(3 * w.numel(), dtype=torch.int16, device=dev) if k % 16 == 0 else None)
        rows = [(w.data_ptr(), e["fwd"].data_ptr(), r | (s << 32), c | (k << 32), 0)]
        if e["bwd"] is not None:
            rows.append((w.data_ptr(), e["bwd"].data_ptr(), r | (s << 32), c | (k << 32), 1))
        ce = _lib.lib().embnet_conv_weight_planes_chunk_elems()
        e["rows"] = rows
        e["table"] = torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(dev)
        e["chunks"] = torch.tensor([(i, j) for i in range(len(rows)) for j in range(-(-w.numel() // ce))], dtype=torch.int32, device=dev)
        w._embnet_wplanes = e
        _SIDE_GEN[0] += 1
    return e


def weight_planes(w, flip):
    """bf16 planes of a Conv2D kernel for the patch kernel (forward: flip 0, stride-1 data gradient: flip 1), rebuilt
    (one launch for both) when the kernel changed since they were made."""
    e = _wplanes_entry(w)
    if e["epoch"] != WEIGHT_EPOCH[0] or e["version"] != w._version:
        check(_lib.lib().embnet_conv_weight_planes(e["table"].data_ptr(), len(e["rows"]), e["chunks"].data_ptr(), e["chunks"].shape[0], stream()))
        e["epoch"], e["version"] = WEIGHT_EPOCH[0], w._version
    return e["bwd"] if flip else e["fwd"]


def _current(e, w):
    return e["epoch"] == WEIGHT_EPOCH[0] and e["version"] == w._version


def _refresh_planes_of(ws, holder):
    """ONE launch for the planes of the kernels `ws` (all of them have planes); the launch plan is cached on `holder`.  Nothing is
    launched when every entry is current (a trainer and its optimizer may both ask after the same update)."""
    import numpy as np
    if not ws or all(_current(w._embnet_wplanes, w) for w in ws):
        return
    key = tuple(id(w._embnet_wplanes) for w in ws)
    plan = getattr(holder, "_wplanes_plan", None)
    if plan is None or plan["key"] != key:
        rows = [r for w in ws for r in w._embnet_wplanes["rows"]]
        ce = _lib.lib().embnet_conv_weight_planes_chunk_elems()
        sizes = [w.numel() for w in ws for _ in w._embnet_wplanes["rows"]]
        dev = ws[0].device
        plan = dict(key=key, n=len(rows), table=torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(dev),
                    chunks=torch.tensor([(i, j) for i, n in enumerate(sizes) for j in range(-(-n // ce))], dtype=torch.int32, device=dev))
        holder._wplanes_plan = plan
    check(_lib.lib().embnet_conv_weight_planes(plan["table"].data_ptr(), plan["n"], plan["chunks"].data_ptr(), plan["chunks"].shape[0], stream()))
    for w in ws:
        e = w._embnet_wplanes
        e["epoch"], e["version"] = WEIGHT_EPOCH[0], w._version


def _with_entry(tensors, attr):
    out = []
    for w in tensors:
        e = getattr(w, attr, None)
        if e is not None and e["ptr"] == w.data_ptr():
            out.append(w)
    return out


def refresh_weight_planes(module):
    """Rebuild the planes of every kernel that has them, in ONE launch (called by the trainer right after the optimizer
    step, so that the next forward finds them current; inside a captured step this launch is part of the graph)."""
    kernels = [m.kernel for m in module.modules() if isinstance(m, Conv2D)]
    _refresh_ranges_of(_with_entry(kernels, "_embnet_wrange"), module)        # (the gather convs' kernel ranges ride along)
    _refresh_planes_of(_with_entry(kernels, "_embnet_wplanes"), module)


_SIDE_GEN = [0]             # bumped whenever a kernel tensor gains a planes / range entry


def refresh_tensors(tensors, holder):
    """The same for a list of parameter tensors — what KerasOptimizer.step() calls on its own parameters right after the update
    launch, so that ANY training loop (SiameseNet's, a user's) gets the one-launch refresh instead of one lazy launch per kernel
    at the next forward (ResNet50: 53 kernels -> 2 + 4 launches).  Which tensors have entries is cached on `holder`."""
    plan = getattr(holder, "_side_plan", None)
    if plan is None or plan[0] != _SIDE_GEN[0]:
        plan = holder._side_plan = (_SIDE_GEN[0], _with_entry(tensors, "_embnet_wrange"), _with_entry(tensors, "_embnet_wplanes"))
    _refresh_ranges_of(plan[1], holder)
    _refresh_planes_of(plan[2], holder)


# ---- three products per fp32 product on the gather convs (csrc/conv.hip "Ranges"; include/embnet.h ABI 21) --------------------------
# A Conv2D with `f16 = True` (the zoo ResNets' convs, backbones._rn_conv) that does NOT run the patch kernel hands the library the
# RANGE SLOT of both operands of each of its three passes as ARGUMENTS (embnet_conv2d_*_f32_ex), and the kernels then multiply in
# the planes kernels' two-piece fp16 format (three matrix products per fp32 product instead of six).  EVERY operand has a range
# — there is no default scale (round 5 ran activations at scale 1: a precision cliff below amplitude 2^-3):
#   * kernels: one uint32 slot per kernel, refreshed like the kernel planes (tensor version / WEIGHT_EPOCH; refresh_weight_planes
#     does all of a model's in two launches) — the exact maximum;
#   * activations: the training-mode BatchNormalization that writes the tensor leaves an upper bound of max |y| in a slot before
#     its apply pass runs (nn_kernels.hip channel_bound: from the statistics partials) and hangs it on the tensor (`y._range`);
#   * gradients: the conv tags its output (`y._wants_dy_range`), the BatchNormalization (or BatchNorm + MaxPool) that reads y makes
#     its backward leave max |dx| in a slot (`dx_range` of embnet_bn_bwd_ex) and files it in DY_RANGE under dx's address — with an
#     alias of dx, so that address cannot pass to another tensor while the entry lives; the conv's backward claims it.
# A pass that lacks either operand's range (an inference-mode BatchNormalization, a gradient that reaches the conv from anywhere
# else) runs the six-term kernel.  Entries a backward leaves unclaimed are dropped — and counted — by the step context's
# end-of-backward sweep (StepContext.unclaimed['dy_range']); nothing is ever dropped by size.
CONV_F16 = [_os.environ.get("EMBNET_CONV_F16", "1") != "0"]           # [False]: six-term products everywhere (A/B)
DY_RANGE = _CtxDict("dy_range")


def _range_entry(w):
    e = getattr(w, "_embnet_wrange", None)
    if e is None or e["ptr"] != w.data_ptr() or e["n"] != w.numel():
        import numpy as np
        dev = w.device
        slot = torch.zeros(1, dtype=torch.int32, device=dev)
        ce = _lib.lib().embnet_range_chunk_elems()
        rows = [(w.data_ptr(), w.numel(), slot.data_ptr())]
        e = dict(ptr=w.data_ptr(), n=w.numel(), epoch=-1, version=-1, slot=slot, rows=rows,
                 table=torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(dev),
                 chunks=torch.tensor([(0, j) for j in range(-(-w.numel() // ce))], dtype=torch.int32, device=dev))
        w._embnet_wrange = e
        _SIDE_GEN[0] += 1
    return e


def weight_range(w):
    """The range slot of kernel tensor `w` (uint32 bit pattern of max |w|, on the device), refreshed when the kernel changed."""
    e = _range_entry(w)
    if e["epoch"] != WEIGHT_EPOCH[0] or e["version"] != w._version:
        check(_lib.lib().embnet_range_multi(e["table"].data_ptr(), 1, e["chunks"].data_ptr(), e["chunks"].shape[0], stream()))
        e["epoch"], e["version"] = WEIGHT_EPOCH[0], w._version
    return e["slot"]


def _refresh_ranges_of(ws, holder):
    """Every kernel range of `ws` in one call (two launches); nothing when all are current."""
    import numpy as np
    if not ws or all(_current(w._embnet_wrange, w) for w in ws):
        return
    key = tuple(id(w._embnet_wrange) for w in ws)
    plan = getattr(holder, "_wrange_plan", None)
    if plan is None or plan["key"] != key:
        rows = [w._embnet_wrange["rows"][0] for w in ws]
        ce = _lib.lib().embnet_range_chunk_elems()
        dev = ws[0].device
        plan = dict(key=key, n=len(rows), table=torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(dev),
                    chunks=torch.tensor([(i, j) for i, w in enumerate(ws) for j in range(-(-w.numel() // ce))],
                                        dtype=torch.int32, device=dev))
        holder._wrange_plan = plan
    check(_lib.lib().embnet_range_multi(plan["table"].data_ptr(), plan["n"], plan["chunks"].data_ptr(), plan["chunks"].shape[0], stream()))
    for w in ws:
        e = w._embnet_wrange
        e["epoch"], e["version"] = WEIGHT_EPOCH[0], w._version


def refresh_weight_ranges(module):
    """Every kernel range of `module` that exists, in one call (two launches)."""
    _refresh_ranges_of(_with_entry([m.kernel for m in module.modules() if isinstance(m, Conv2D)], "_embnet_wrange"), module)


def _rptr(r):
    """Raw device address of a range slot: a uint32 tensor, or (owner tensor, address) for a slot that lives inside another
    tensor (a BatchNormalization's statistics rows: no allocation of its own per layer and step)."""
    if r is None:
        return None
    return r[1] if isinstance(r, tuple) else ptr(r)


def _range_of(x):
    """The range slot the producer of activation tensor x left on it (BatchNormalization.forward), or None."""
    return getattr(x, "_range", None)


def _take_dy_range(dy, keep=False):
    """keep: the same gradient tensor goes on to a second conv (the projection shortcut behind a fused Add) — the entry stays for
    it; an identity shortcut's BatchNormalization releases it instead (_BatchNormFn.backward)."""
    if not DY_RANGE:
        return None
    e = DY_RANGE.get(dy.data_ptr()) if keep else DY_RANGE.pop(dy.data_ptr(), None)
    return e[0] if (e is not None and e[1].shape == dy.shape) else None


def _emit_dx_range(dx):
    """A fresh range slot for `dx` (the caller passes it to the embnet_bn_bwd*_ex call that writes dx), filed in DY_RANGE."""
    slot = torch.empty(_lib.lib().embnet_range_slot_words(), dtype=torch.int32, device=dx.device)
    DY_RANGE[dx.data_ptr()] = (slot, dx.detach())
    return slot


_BN_SCALAR = _os.environ.get("EMBNET_BN_SCALAR", "0") not in ("0", "")


def _planes_range_ok(x):
    """A BatchNorm backward that writes dx as planes AND as fp32 can leave dx's range too: in the two-piece format its dry run
    finds the maximum anyway."""
    return _lib.lib().embnet_conv_planes_mfma_terms() == 3


PLANES_ONLY = [_os.environ.get("EMBNET_PLANES_ONLY", "1") != "0"]     # [False]: every planes tensor keeps its fp32 copy (A/B)


def _placeholder(shape, device):
    """A tensor of `shape` WITHOUT its fp32 storage: stands in autograd's graph for a tensor that exists only as planes.  All
    strides are 0 over a three-float storage at offset 1 — a signature no ordinary tensor has (the gradient of `.sum()` is also
    an all-zero-stride view, of ONE float at offset 0: that one is a real tensor and must be read) — and it is not
    contiguous, so _lib.ptr() refuses it: a kernel that would read its values fails loudly."""
    base = torch.empty(3, device=device, dtype=torch.float32)
    return base.as_strided(tuple(shape), (0,) * len(shape), 1)


def _is_placeholder(t):
    return (t.dim() == 4 and t.numel() > 1 and t.stride() == (0, 0, 0, 0) and t.storage_offset() == 1
            and t.untyped_storage().nbytes() == 12)


def _take_dy_planes(dy):
    e = DY_PLANES.pop(dy.data_ptr(), None)
    return e[0] if (e is not None and e[1].shape == dy.shape) else None


def patch_ok(n, h, wd, c, r, s, k, stride, oh, ow):
    return bool(_lib.lib().embnet_conv2d_patch_supported(n, c, r, s, k, stride, oh, ow))


def _patch_dgrad(dy_planes, w, dx, n, h, wd, c, r, s, k, pt, pl, oh, ow, dx_add, bn_src=None):
    """dx[n,h,wd,c] (+ dx_add) = stride-1 data gradient through the patch kernel: the correlation of the dy planes
    [n,oh,ow,k] with the flipped kernel planes.
    bn_src = (bn_x, bn_stats, bn_act) (dx_add None): the conv's input was act(BatchNorm(bn_x)) — the kernel's epilogue also emits
    that layer's backward sums (BN_SUMS, as the gather-loop data gradient does)."""
    lib = _lib.lib()
    ws = workspace(lib.embnet_conv2d_patch_workspace_bytes(n, k, r, s, c, h, wd), dx.device)
    if bn_src is not None and dx_add is None and PATCH_BN_SUMS[0]:
        bn_x, bn_stats, bn_act = bn_src
        rows = lib.embnet_conv2d_patch_stats_rows(n, h, wd)
        partial = torch.empty((2, c, rows), device=dx.device, dtype=torch.float32)
        sp = bn_stats.data_ptr()
        check(lib.embnet_conv2d_patch_bnsums_f32(ptr(dy_planes), ptr(weight_planes(w, 1)), ptr(dx), n, oh, ow, k, r, s, c,
                                                 r - 1 - pt, s - 1 - pl, h, wd, ptr(bn_x), sp + 8 * c, sp + 12 * c, sp, sp + 4 * c,
                                                 int(bn_act), ptr(partial), rows, ptr(ws), ws.numel() * 4, stream()))
        while len(BN_SUMS) >= 8:     # unclaimed entries (the gradient got a second contribution) pin a dx each: keep few
            BN_SUMS.pop(next(iter(BN_SUMS)))
        BN_SUMS[dx.data_ptr()] = (partial, rows, bn_x.data_ptr(), dx.detach())
        return
    check(lib.embnet_conv2d_patch_f32(ptr(dy_planes), ptr(weight_planes(w, 1)), None, ptr(dx), n, oh, ow, k, r, s, c,
                                      r - 1 - pt, s - 1 - pl, h, wd, 0, ptr(dx_add), None, ptr(ws), ws.numel() * 4, stream()))


def _c(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def same_pad(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return out, total // 2


# ----------------------------------------------------------------------------- conv
class _Conv2dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, geom, relu, residual=None, in_stats=None, in_act=0, out_stats=None, with_skip=False,
                planes=None, bn_src=None, w_range=None, x_range=None, dma=False):
        """planes: the input's pre-split planes (layers.DY_PLANES note above) -> the patch kernel computes the forward.
        bn_src = (bn_x, bn_stats, bn_act): x is act(BatchNorm(bn_x)) — the data gradient also emits that layer's backward sums.
        w_range: the kernel's range slot (weight_range), x_range: the input's (layers._range_of) -> the gather kernels multiply
        on three products in every pass whose two operands both have a range (forward: x, w; data gradient: dy (DY_RANGE), w;
        weight gradient: x, dy)."""
        w = _c(w)
        if not (planes is not None and _is_placeholder(x)):       # (a planes-only input has no fp32 values to read)
            x = _c(x)
        n, h, wd, c = x.shape
        r, s, c2, k = w.shape
        if c2 != c:
            raise _lib.EmbnetError(f"conv2d: input has {c} channels, kernel expects {c2}")
        stride, pt, pl, oh, ow = geom
        # in_stats [4,C] (mean, rstd, scale, shift of the BatchNormalization in front): the kernels read
        # act(x*scale + shift) on the fly, x being the BN's INPUT (layers.Deferred)
        in_scale = (in_stats.data_ptr() + 8 * in_stats.shape[1]) if in_stats is not None else None
        in_shift = (in_stats.data_ptr() + 12 * in_stats.shape[1]) if in_stats is not None else None
        if residual is not None:
            if relu:
                raise _lib.EmbnetError("conv2d: a fused residual add goes with a linear conv (no fused ReLU)")
            residual = _c(residual)
            if tuple(residual.shape) != (n, oh, ow, k):
                raise _lib.EmbnetError(f"Add: shapes differ {(n, oh, ow, k)} vs {tuple(residual.shape)}")
        y = torch.empty((n, oh, ow, k), device=x.device, dtype=torch.float32)
        lib = _lib.lib()
        if planes is not None and r == 1:           # 1x1 on the planes (csrc/conv_patch.hip conv1x1_planes_kernel): forward only —
            ws = workspace(lib.embnet_conv2d_patch_workspace_bytes(n, c, 1, 1, k, oh, ow), x.device)   # backward stays on the ranged gather kernels
            check(lib.embnet_conv2d_planes1x1_f32(ptr(planes), ptr(weight_planes(w, 0)), ptr(bias), ptr(y), n, h, wd, c, k, stride,
                                                  oh, ow, int(relu), ptr(residual), ptr(out_stats), ptr(ws), ws.numel() * 4, stream()))
        elif planes is not None:
            ws = workspace(lib.embnet_conv2d_patch_workspace_bytes(n, c, r, s, k, oh, ow), x.device)
            check(lib.embnet_conv2d_patch_f32(ptr(planes), ptr(weight_planes(w, 0)), ptr(bias), ptr(y), n, h, wd, c, r, s, k,
                                              pt, pl, oh, ow, int(relu), ptr(residual), ptr(out_stats), ptr(ws),
                                              ws.numel() * 4, stream()))
        elif dma:                                   # 1x1, both ranges known: x by LDS-DMA as fp32, the kernel planes (Conv2D.forward decided)
            ws = workspace(lib.embnet_conv2d_patch_workspace_bytes(n, c, 1, 1, k, oh, ow), x.device)
            check(lib.embnet_conv2d_dma1x1_f32(ptr(x), ptr(weight_planes(w, 0)), ptr(bias), ptr(y), n, h, wd, c, k, stride, oh, ow,
                                               int(relu), ptr(residual), ptr(out_stats), _rptr(x_range), ptr(ws), ws.numel() * 4, stream()))
        else:
            ws = workspace(lib.embnet_conv2d_fwd_workspace_bytes(n, c, r, s, k, oh, ow), x.device)
            fr = w_range is not None and x_range is not None and in_stats is None
            check(lib.embnet_conv2d_fwd_f32_ex(
                ptr(x), ptr(w), ptr(bias), ptr(y), n, h, wd, c, r, s, k, stride, pt, pl, oh, ow, int(relu), ptr(residual),
                in_scale, in_shift, int(in_act), ptr(out_stats), ptr(ws), ws.numel() * 4,
                _rptr(x_range) if fr else None, _rptr(w_range) if fr else None, stream()))
        one = planes is not None and r == 1         # (a 1x1 planes forward: data and weight gradient run the gather kernels, on their ranges)
        ctx.patch = planes is not None and not one
        ctx.w_range = w_range if ((planes is None or one) and in_stats is None) else None
        ctx.x_range = x_range if ((planes is None or one) and in_stats is None) else None
        ctx.x_planes = None if one else planes   # kept for the weight gradient (conv_wgrad)
        ctx.bn_src = bn_src
        ctx.geom, ctx.relu, ctx.has_bias, ctx.has_res = geom, relu, bias is not None, residual is not None
        ctx.in_act = int(in_act)
        ctx.bias_ref = bias                 # only its address / shape are used (gradient sink lookup)
        ctx.save_for_backward(x, w, y if relu else None, in_stats)
        if with_skip:                       # second output: x itself, for a skip connection (see backward)
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        """dskip: gradient of the pass-through copy of x (with_skip) — added in the data-gradient epilogue."""
        x, w, y, in_stats = ctx.saved_tensors
        lib = _lib.lib()
        n, h, wd, c = x.shape
        r, s, _, k = w.shape
        stride, pt, pl, oh, ow = ctx.geom
        dy_only_planes = _is_placeholder(dy)     # the BatchNormalization behind this conv wrote its dx as planes ONLY
        if not dy_only_planes:
            dy = _c(dy)
        dskip = _c(dskip) if dskip is not None else None
        dx = dw = db = None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        w_range = getattr(ctx, "w_range", None)
        dz_range = None                      # (fused ReLU) the range of dz, from the pass that applied the mask
        done = RELU_DONE.pop(dy.data_ptr(), None) if (ctx.relu and RELU_DONE) else None
        if done is not None and done[0].shape == dy.shape:
            dz = dy                          # the BatchNormalization / MaxPool2D behind this conv already applied the ReLU mask ...
            if want_db:                      # ... and summed the bias gradient (into the bias' sink when it has one)
                db = _done(done[1], done[2])
                want_db = False
            if w_range is not None:          # ... and left the range of dz (the conv's output was tagged `_wants_dy_range`)
                dz_range = _take_dy_range(dy)
        elif ctx.relu:
            dz = torch.empty_like(dy)
            if want_db:                      # dz and its column sums (the bias gradient) in one pass — and its range when wanted
                db, db_note = _sink(ctx.bias_ref)
                ws = workspace(lib.embnet_colsum_workspace_bytes(dy.numel() // k, k), x.device)
                if w_range is not None:
                    dz_range = _new_range_slot(x.device)
                check(lib.embnet_relu_bwd_colsum_ex(ptr(dy), ptr(y), dy.numel() // k, k, ptr(dz), ptr(db), ptr(ws),
                                                    ws.numel() * 4, ptr(dz_range), stream()))
                db = _done(db, db_note)
                want_db = False
            else:
                check(lib.embnet_relu_bwd(ptr(dy), ptr(y), dy.numel(), ptr(dz), stream()))
        else:
            dz = dy
        in_scale = (in_stats.data_ptr() + 8 * in_stats.shape[1]) if in_stats is not None else None
        in_shift = (in_stats.data_ptr() + 12 * in_stats.shape[1]) if in_stats is not None else None

        # planes of dy left by the BatchNormalization behind this conv (only usable when dz IS dy: no fused ReLU)
        dy_planes = _take_dy_planes(dy) if (ctx.patch and not ctx.relu) else None
        # ... or its range, for the three-product gather kernels (same condition)
        if ctx.relu:
            dy_range = dz_range
        else:
            dy_range = _take_dy_range(dy, keep=ctx.has_res) if (w_range is not None and not dy_only_planes) else None
        if dy_only_planes and dy_planes is None:
            raise _lib.EmbnetError("conv2d backward: the gradient exists only as planes and they are gone (DY_PLANES)")
        need_dx, need_dw = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if need_dw and dy_planes is None and _is_placeholder(x):
            # the input exists only as planes and nobody left planes of the gradient (its producer was not the BatchNormalization
            # the forward saw): split it here, one extra pass
            dy_planes = torch.empty(3 * dz.numel(), device=dz.device, dtype=torch.int16)
            check(lib.embnet_planes_from_f32(ptr(dz), dz.numel() // k, k, ptr(dy_planes), stream()))

        def run_wgrad():
            conv_wgrad(lib, x, dz, dw, w, n, h, wd, c, r, s, k, stride, pt, pl, oh, ow, in_scale, in_shift, ctx.in_act,
                       getattr(ctx, "x_planes", None), dy_planes, dy_range, getattr(ctx, "x_range", None))

        dw_note = None
        if need_dw:
            dw, dw_note = _sink(w)
        overlap = OVERLAP_WGRAD and need_dx and need_dw
        if overlap:
            main = torch.cuda.current_stream()
            side = _side_stream(x.device)
            side.wait_stream(main)
            with torch.cuda.stream(side):
                run_wgrad()
        if need_dx:
            dx = torch.empty(x.shape, device=x.device, dtype=torch.float32)
            if dy_planes is not None and patch_ok(n, oh, ow, k, r, s, c, 1, h, wd):
                _patch_dgrad(dy_planes, w, dx, n, h, wd, c, r, s, k, pt, pl, oh, ow, dskip,
                             getattr(ctx, "bn_src", None) if dskip is None else None)
            elif (r == 1 and s == 1 and stride == 1 and dy_range is not None and w_range is not None and w.shape[3] % 16 == 0
                  and _dma1x1_ok(n, oh, ow, k, c, 1, h, wd)
                  and (dskip is not None or getattr(ctx, "bn_src", None) is None
                       or lib.embnet_conv2d_dgrad_bnsums_rows(n, h, wd, c, r, s, k, stride) <= 0)):
                # 1x1 stride-1 data gradient without BatchNorm sums: dz by LDS-DMA as fp32 against the flipped kernel planes (c and k
                # swap roles)
                dws = workspace(lib.embnet_conv2d_patch_workspace_bytes(n, k, 1, 1, c, h, wd), x.device)
                check(lib.embnet_conv2d_dma1x1_f32(ptr(dz), ptr(weight_planes(w, 1)), None, ptr(dx), n, oh, ow, k, c, 1, h, wd, 0,
                                                   ptr(dskip), None, _rptr(dy_range), ptr(dws), dws.numel() * 4, stream()))
            else:
                # its own scratch: with OVERLAP_WGRAD the wgrad slabs are in flight on the side stream's buffer
                dws = workspace(lib.embnet_conv2d_dgrad_workspace_bytes(n, h, wd, c, r, s, k, stride), x.device)
                bn_src = getattr(ctx, "bn_src", None) if dskip is None else None
                rows = lib.embnet_conv2d_dgrad_bnsums_rows(n, h, wd, c, r, s, k, stride) if bn_src is not None else 0
                if rows > 0:
                    bn_x, bn_stats, bn_act = bn_src
                    partial = torch.empty((3, c, rows), device=x.device, dtype=torch.float32)      # sums, and max |dz| per band
                    sp = bn_stats.data_ptr()
                    check(lib.embnet_conv2d_dgrad_bnsums_f32_ex(
                        ptr(dz), ptr(w), ptr(dx), n, h, wd, c, r, s, k, stride, pt, pl, oh, ow, ptr(bn_x), sp + 8 * c, sp + 12 * c,
                        sp, sp + 4 * c, int(bn_act), ptr(partial), rows, ptr(dws), dws.numel() * 4,
                        _rptr(dy_range), _rptr(w_range) if dy_range is not None else None, stream()))
                    while len(BN_SUMS) >= 8:     # unclaimed entries (the gradient got a second contribution) pin a dx each: keep few
                        BN_SUMS.pop(next(iter(BN_SUMS)))
                    BN_SUMS[dx.data_ptr()] = (partial, rows, bn_x.data_ptr(), dx.detach())
                else:
                    check(lib.embnet_conv2d_dgrad_f32_ex(
                        ptr(dz), ptr(w), ptr(dx), n, h, wd, c, r, s, k, stride, pt, pl, oh, ow, 0, ptr(dskip), ptr(dws),
                        dws.numel() * 4, _rptr(dy_range), _rptr(w_range) if dy_range is not None else None, stream()))
        if overlap:
            torch.cuda.current_stream().wait_stream(side)
        elif need_dw:
            run_wgrad()
        if need_dw:
            dw = _done(dw, dw_note)
        if want_db:
            db, db_note = _sink(ctx.bias_ref)
            _colsum(dz.view(-1, k), out=db)
            db = _done(db, db_note)
        if dskip is not None and dx is None and ctx.needs_input_grad[0]:
            dx = dskip
        if ctx.has_res and dy_only_planes:
            # (the residual branch would receive a placeholder — three uninitialised floats behind zero strides)
            raise _lib.EmbnetError("conv2d backward: a conv with a fused Add received its gradient as planes only; the Add's other "
                                   "branch needs the fp32 gradient (BatchNormalization(owns_input=True) behind conv(residual=...))")
        return dx, dw, db, None, None, (dy if ctx.has_res else None), None, None, None, None, None, None, None, None, None


class _ConvPairFn(torch.autograd.Function):
    """Two bias-free linear convs on the SAME input (a residual unit's first 3x3 and its 1x1 projection shortcut).
    One autograd node, so backward writes the input gradient once: the second data-gradient kernel adds into the
    first one's output in its epilogue (and skips the pixels its taps never reach) instead of leaving two tensors
    for autograd to add."""

    @staticmethod
    def forward(ctx, x, w1, geom1, w2, geom2, in_stats, in_act, out_stats1, planes=None, w_ranges=None, x_range=None):
        """planes: pre-split planes of x -> the FIRST conv (the 3x3) runs the patch kernel.
        w_ranges = (range slot of w1 or None, of w2 or None), x_range: the input's: see _Conv2dFn.forward."""
        w_ranges = tuple(w_ranges) if (w_ranges is not None and in_stats is None and x_range is not None) else (None, None)
        x, w1, w2 = _c(x), _c(w1), _c(w2)
        lib = _lib.lib()
        n, h, wd, c = x.shape
        in_scale = (in_stats.data_ptr() + 8 * in_stats.shape[1]) if in_stats is not None else None
        in_shift = (in_stats.data_ptr() + 12 * in_stats.shape[1]) if in_stats is not None else None
        ys = []
        for w, geom, st in ((w1, geom1, out_stats1), (w2, geom2, None)):
            r, s, c2, k = w.shape
            if c2 != c:
                raise _lib.EmbnetError(f"conv2d: input has {c} channels, kernel expects {c2}")
            stride, pt, pl, oh, ow = geom
            y = torch.empty((n, oh, ow, k), device=x.device, dtype=torch.float32)
            if planes is not None and w is w1:
                ws = workspace(lib.embnet_conv2d_patch_workspace_bytes(n, c, r, s, k, oh, ow), x.device)
                check(lib.embnet_conv2d_patch_f32(ptr(planes), ptr(weight_planes(w, 0)), None, ptr(y), n, h, wd, c, r, s, k,
                                                  pt, pl, oh, ow, 0, None, ptr(st), ptr(ws), ws.numel() * 4, stream()))
            else:
                ws = workspace(lib.embnet_conv2d_fwd_workspace_bytes(n, c, r, s, k, oh, ow), x.device)
                wr = w_ranges[0 if w is w1 else 1]
                check(lib.embnet_conv2d_fwd_f32_ex(
                    ptr(x), ptr(w), None, ptr(y), n, h, wd, c, r, s, k, stride, pt, pl, oh, ow, 0, None, in_scale, in_shift,
                    int(in_act), ptr(st), ptr(ws), ws.numel() * 4, _rptr(x_range) if wr is not None else None, _rptr(wr), stream()))
            ys.append(y)
        ctx.patch = planes is not None
        ctx.w_ranges = (w_ranges[0] if planes is None else None, w_ranges[1])
        ctx.x_range = x_range
        ctx.x_planes = planes
        ctx.geoms, ctx.in_act = (geom1, geom2), int(in_act)
        ctx.save_for_backward(x, w1, w2, in_stats)
        return ys[0], ys[1]

    @staticmethod
    def backward(ctx, dy1, dy2):
        x, w1, w2, in_stats = ctx.saved_tensors
        lib = _lib.lib()
        n, h, wd, c = x.shape
        in_scale = (in_stats.data_ptr() + 8 * in_stats.shape[1]) if in_stats is not None else None
        in_shift = (in_stats.data_ptr() + 12 * in_stats.shape[1]) if in_stats is not None else None
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dws, first = [], True
        for w, geom, dy, need_dw in ((w1, ctx.geoms[0], dy1, ctx.needs_input_grad[1]),
                                     (w2, ctx.geoms[1], dy2, ctx.needs_input_grad[3])):
            if dy is None:
                dws.append(None)
                continue
            if not _is_placeholder(dy):
                dy = _c(dy)
            r, s, _, k = w.shape
            stride, pt, pl, oh, ow = geom
            dy_planes = _take_dy_planes(dy) if (ctx.patch and w is w1) else None
            if _is_placeholder(dy) and dy_planes is None:
                raise _lib.EmbnetError("conv_pair backward: the gradient exists only as planes and they are gone (DY_PLANES)")
            w_range = getattr(ctx, "w_ranges", (None, None))[0 if w is w1 else 1]
            dy_range = _take_dy_range(dy) if (w_range is not None and not _is_placeholder(dy)) else None
            if dx is not None:
                if dy_planes is not None and first and patch_ok(n, oh, ow, k, r, s, c, 1, h, wd):
                    _patch_dgrad(dy_planes, w, dx, n, h, wd, c, r, s, k, pt, pl, oh, ow, None)
                else:
                    sc = workspace(lib.embnet_conv2d_dgrad_workspace_bytes(n, h, wd, c, r, s, k, stride), x.device)
                    check(lib.embnet_conv2d_dgrad_f32_ex(
                        ptr(dy), ptr(w), ptr(dx), n, h, wd, c, r, s, k, stride, pt, pl, oh, ow, 0 if first else 1, None,
                        ptr(sc), sc.numel() * 4, _rptr(dy_range), _rptr(w_range) if dy_range is not None else None, stream()))
                first = False
            dw = None
            if need_dw:
                dw, note = _sink(w)
                conv_wgrad(lib, x, dy, dw, w, n, h, wd, c, r, s, k, stride, pt, pl, oh, ow, in_scale, in_shift, ctx.in_act,
                           getattr(ctx, "x_planes", None) if w is w1 else None, dy_planes, dy_range,
                           getattr(ctx, "x_range", None) if w_range is not None else None)
                dw = _done(dw, note)
            dws.append(dw)
        if dx is not None and first:
            dx.zero_()
        return dx, dws[0], None, dws[1], None, None, None, None, None, None, None


def conv_pair(x, conv1, conv2, emit_stats=False):
    """(conv1(x), conv2(x)) for two Conv2D layers reading the same tensor (or Deferred BN output); fused into one
    autograd node when both are plain linear convs (no bias, no activation).  emit_stats applies to conv1."""
    if conv1.bias is not None or conv2.bias is not None or conv1.relu or conv2.relu:
        return conv1(x, emit_stats=emit_stats), conv2(x)
    in_stats, in_act = None, 0
    if isinstance(x, Deferred):
        if any(cv.kernel.shape[2] % 4 or cv.kernel.shape[3] % 4 for cv in (conv1, conv2)):
            x = x.materialize()
        else:
            x, in_stats, in_act = x.raw, x.stats, x.act
    g1, g2 = conv1.geometry(x.shape[1], x.shape[2]), conv2.geometry(x.shape[1], x.shape[2])
    planes = getattr(x, "_planes", None) if in_stats is None else None
    if planes is not None and not conv1.patch_capable(x.shape):
        planes = None
    out_stats = None
    if emit_stats:
        r, s, c, k = conv1.kernel.shape
        if planes is not None:
            rows = _lib.lib().embnet_conv2d_patch_stats_rows(x.shape[0], g1[3], g1[4])
        else:
            rows = _lib.lib().embnet_conv2d_fwd_stats_rows(x.shape[0], c, r, s, k, g1[3], g1[4])
        if rows > 0:
            out_stats = torch.empty((2, k, rows), device=x.device, dtype=torch.float32)
    wr = (conv1.range_for(x, planes, in_stats), conv2.range_for(x, None, in_stats))
    y1, y2 = _ConvPairFn.apply(x, conv1.kernel, g1, conv2.kernel, g2, in_stats, in_act, out_stats, planes,
                               wr if (wr[0] is not None or wr[1] is not None) else None, _range_of(x))
    if out_stats is not None:
        y1._bn_partials = out_stats
    if wr[0] is not None and torch.is_grad_enabled():
        y1._wants_dy_range = True
    if wr[1] is not None and torch.is_grad_enabled():
        y2._wants_dy_range = True
    if planes is not None:
        y1._wants_dy_planes = True
        if conv1.planes_only_gradient(x.shape, g1):
            y1._dy_planes_only = True
    return y1, y2


def _colsum(x2d, out=None):
    lib = _lib.lib()
    m, c = x2d.shape
    if out is None:
        out = torch.empty((c,), device=x2d.device, dtype=torch.float32)
    ws = workspace(lib.embnet_colsum_workspace_bytes(m, c), x2d.device)
    check(lib.embnet_colsum(ptr(x2d), m, c, ptr(out), ptr(ws), ws.numel() * 4, stream()))
    return out


def glorot_uniform_(t, gen):
    shape = t.shape
    rf = int(math.prod(shape[:-2])) if len(shape) > 2 else 1
    fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return t.uniform_(-lim, lim, generator=gen)


def he_uniform_(t, gen):
    shape = t.shape
    rf = int(math.prod(shape[:-2])) if len(shape) > 2 else 1
    lim = math.sqrt(6.0 / (shape[-2] * rf))
    return t.uniform_(-lim, lim, generator=gen)


class _PadKernelFn(torch.autograd.Function):
    """kernel [R,S,C,K] -> [R,S,Cp,K] with zero input channels appended (forward) / gradient of the first C channels (backward):
    lets a conv whose input has 3 channels run the 16-byte-gather kernels on a 4-channel copy of the image."""

    @staticmethod
    def forward(ctx, w, cp):
        r, s, c, k = w.shape
        wp = torch.zeros((r, s, cp, k), device=w.device, dtype=torch.float32)
        wp[:, :, :c, :].copy_(w)
        ctx.c = c
        return wp

    @staticmethod
    def backward(ctx, dwp):
        return dwp[:, :, :ctx.c, :].contiguous(), None


def _new_range_slot(device):
    return torch.empty(_lib.lib().embnet_range_slot_words(), dtype=torch.int32, device=device)


def pad_channels(x, cp, with_range=False):
    """NHWC image batch with C channels -> the same with zero channels appended up to cp (no gradient).
    with_range: the copy carries its exact range (`_range`, layers.CONV_F16 note) — the pass reads every value anyway."""
    x = _c(x.detach())
    n, h, w, c = x.shape
    xp = torch.empty((n, h, w, cp), device=x.device, dtype=torch.float32)
    slot = _new_range_slot(x.device) if with_range else None
    check(_lib.lib().embnet_pad_channels_ex(ptr(x), n * h * w, c, cp, ptr(xp), ptr(slot), stream()))
    if slot is not None:
        xp._range = slot
    return xp


# a first-layer conv on a 3-channel image with a large kernel (the `simple` backbone's 10x10x3 -> 64,
# /root/reference/embedding_net/backbones.py:21-22) or on a large batch of large images (EfficientNet's 3x3 stride-2 stem at
# 224x224 x 256: 234 + 309 us forward + weight gradient) runs the scalar-gather kernels: 18-82 TFLOP/s
PAD_INPUT_CONV = [_os.environ.get("EMBNET_PAD_INPUT_CONV", "1") != "0"]


class Conv2D(nn.Module):
    """Keras Conv2D.  padding: 'valid' | 'same' | int (a ZeroPadding2D(p) in front of a valid conv)."""

    def __init__(self, in_channels, filters, kernel_size, strides=1, padding="valid", activation=None,
                 use_bias=True, kernel_initializer="glorot_uniform", l2=0.0, gen=None):
        super().__init__()
        if activation not in (None, "relu"):
            raise ValueError("Conv2D supports activation None or 'relu'")
        self.k, self.stride, self.padding, self.relu, self.l2 = kernel_size, strides, padding, activation == "relu", l2
        w = torch.empty(kernel_size, kernel_size, in_channels, filters)
        {"he_uniform": he_uniform_, "conv_normal": conv_normal_}.get(kernel_initializer, glorot_uniform_)(w, gen)
        self.kernel = nn.Parameter(w)
        self.bias = nn.Parameter(torch.zeros(filters)) if use_bias else None
        self.f16 = False          # True: three-product gather kernels where the operands' ranges are known (layers.CONV_F16 note)

    def range_for(self, x, planes, in_stats, kernel=None):
        """The kernel's range slot when this conv, on that input, runs the gather kernels on three products; else None (among
        the reasons: the input carries no range — nobody vouches for its magnitude, so six exact bf16 terms it is).
        kernel: the tensor the kernels read when it is not self.kernel (the channel-padded copy of a first-layer kernel: its
        range is self.kernel's — the padding is zeros).  A fused ReLU is no obstacle: the gradient behind the mask comes with
        its range from the pass that applies the mask (MaxPool2D / BatchNormalization backward, embnet_relu_bwd_colsum_ex)."""
        if not (CONV_F16[0] and getattr(self, "f16", False)) or (planes is not None and self.k != 1) or in_stats is not None:
            return None
        if _range_of(x) is None:
            return None
        kernel = self.kernel if kernel is None else kernel
        if kernel.shape[2] % 4 or kernel.shape[3] % 4 or x.shape[-1] % 4 or _BN_SCALAR:
            return None
        return weight_range(self.kernel)

    def geometry(self, h, w):
        k, s = self.k, self.stride
        if self.padding == "same":
            oh, pt = same_pad(h, k, s)
            ow, pl = same_pad(w, k, s)
        else:
            p = 0 if self.padding == "valid" else int(self.padding)
            pt = pl = p
            oh, ow = (h + 2 * p - k) // s + 1, (w + 2 * p - k) // s + 1
        if oh <= 0 or ow <= 0:
            raise _lib.EmbnetError(f"Conv2D {k}x{k}/{s} '{self.padding}' does not fit a {h}x{w} input")
        return (s, pt, pl, oh, ow)

    def forward(self, x, residual=None, emit_stats=False, with_skip=False):
        """residual: the other input of the Add layer that follows this conv (added in the conv epilogue).
        with_skip=True: returns (conv(x), x) — use the second value for the skip connection that also consumes x; its
        gradient is then added in this conv's data-gradient epilogue instead of by an autograd accumulation pass.
        x may be a Deferred BatchNormalization output: the conv then applies the BN affine + activation itself.
        emit_stats: a training-mode BatchNormalization reads this output next — the conv epilogue produces its
        per-channel sums while the tiles are in registers (attached to the result as `_bn_partials`)."""
        in_stats, in_act = None, 0
        if isinstance(x, Deferred):
            if self.kernel.shape[2] % 4 or self.kernel.shape[3] % 4:
                x = x.materialize()
            else:
                x, in_stats, in_act = x.raw, x.stats, x.act
        geom = self.geometry(x.shape[1], x.shape[2])
        kernel = self.kernel
        if (PAD_INPUT_CONV[0] and in_stats is None and x.shape[-1] % 4 and not x.requires_grad and kernel.shape[3] % 4 == 0
                and (kernel.shape[0] * kernel.shape[1] * kernel.shape[2] >= 128 or x.numel() // x.shape[-1] >= (1 << 20))
                and residual is None and not with_skip):
            cp = (x.shape[-1] + 3) // 4 * 4           # image input, large kernel: 4-channel copy, 16-byte gathers
            x = pad_channels(x, cp, with_range=bool(CONV_F16[0] and getattr(self, "f16", False) and not _BN_SCALAR))
            kernel = _PadKernelFn.apply(kernel, cp)
        planes = getattr(x, "_planes", None) if in_stats is None else None
        if planes is not None and not self.patch_capable(x.shape):
            planes = None
        w_range = self.range_for(x, planes, in_stats, kernel)
        dma = bool(self.k == 1 and planes is None and w_range is not None and kernel is self.kernel
                   and _dma1x1_ok(x.shape[0], x.shape[1], x.shape[2], x.shape[3], kernel.shape[3], geom[0], geom[3], geom[4]))
        out_stats = None
        if emit_stats:
            r, s, c, k = kernel.shape
            if planes is not None or dma:
                rows = _lib.lib().embnet_conv2d_patch_stats_rows(x.shape[0], geom[3], geom[4])
            else:
                rows = _lib.lib().embnet_conv2d_fwd_stats_rows(x.shape[0], c, r, s, k, geom[3], geom[4])
            if rows > 0:
                out_stats = torch.empty((2, k, rows), device=x.device, dtype=torch.float32)
        bn_src = getattr(x, "_bn_src", None)
        if not (FUSE_BN_SUMS[0] and bn_src is not None and in_stats is None and self.stride == 1 and not with_skip
                and torch.is_grad_enabled() and x.requires_grad and kernel is self.kernel):
            bn_src = None
        out = _Conv2dFn.apply(x, kernel, self.bias, geom, self.relu, residual, in_stats, in_act, out_stats, with_skip,
                              planes, bn_src, w_range, _range_of(x) if w_range is not None else None, dma)
        y = out[0] if with_skip else out
        if out_stats is not None:
            y._bn_partials = out_stats
        if torch.is_grad_enabled() and (w_range is not None or residual is not None):
            # the BatchNormalization reading y leaves the range of its dx in DY_RANGE (backward) — for this conv, for the conv behind
            # the fused Add (the projection shortcut), which receives the same gradient tensor, or for the BatchNormalization at the
            # head of an identity shortcut, which adds that gradient to its own dx and bounds the sum with it (no dry run)
            y._wants_dy_range = True
        if planes is not None and not self.relu and self.k != 1:
            y._wants_dy_planes = True          # the BatchNormalization reading y writes its dx also as planes (backward)
            # ... and ONLY as planes when this conv takes both of its gradients from them and the BatchNormalization is told
            # that nobody else reads its dx (BatchNormalization.forward(owns_input=True))
            # (never behind a fused Add: backward hands dy on to the Add's other branch, which needs the fp32 values — ADVICE r05)
            if self.planes_only_gradient(x.shape, geom) and kernel is self.kernel and not with_skip and residual is None:
                y._dy_planes_only = True
        if (self.relu and self.bias is not None and (FUSE_RELU_BN[0] or FUSE_RELU_POOL[0]) and self.kernel.shape[3] % 4 == 0
                and residual is None):
            y._relu_conv = (self.bias,)        # see RELU_DONE
        if with_skip and _range_of(x) is not None:
            out[1]._range = _range_of(x)       # the pass-through copy is the same tensor: same range
        return (y, out[1]) if with_skip else y

    def planes_only_input(self, x_shape):
        """True when no pass of this conv reads an fp32 copy of its input: forward on the patch kernel, weight gradient on the
        planes kernel (the data gradient never reads the input)."""
        if not (PLANES_ONLY[0] and WGRAD_PLANES[0] and self.k == 3 and self.patch_capable(x_shape)):
            return False
        n, h, w, c = x_shape
        stride, pt, pl, oh, ow = self.geometry(h, w)
        return wgrad_planes_ok(n, h, w, c, 3, 3, self.kernel.shape[3], stride, pt, pl, oh, ow)

    def planes_only_gradient(self, x_shape, geom=None):
        """True when this conv's backward reads NO fp32 copy of its output gradient: data gradient on the patch kernel, weight
        gradient on the planes kernel, no bias gradient, no fused ReLU."""
        if not (PLANES_ONLY[0] and WGRAD_PLANES[0] and self.bias is None and not self.relu and self.k == 3 and self.patch_capable(x_shape)):
            return False
        n, h, w, c = x_shape
        stride, pt, pl, oh, ow = geom if geom is not None else self.geometry(h, w)
        k = self.kernel.shape[3]
        return wgrad_planes_ok(n, h, w, c, 3, 3, k, stride, pt, pl, oh, ow) and patch_ok(n, oh, ow, k, 3, 3, c, 1, h, w)

    def patch_capable(self, x_shape):
        """True when this conv on an input of that shape runs the patch kernel (csrc/conv_patch.hip): 3x3, stride 1,
        C % 16 == 0, K % 4 == 0 and an LDS budget the library checks."""
        if len(x_shape) != 4 or not PATCH_CONV[0]:
            return False
        n, h, w, c = x_shape
        if self.k == 1:          # the planes GEMM (forward only): where a BatchNormalization was told to write this conv's input as planes
            if not (CONV1X1_PLANES[0] and getattr(self, "planes1x1", False)) or self.padding not in ("valid", "same", 0):
                return False
            _, _, _, oh, ow = self.geometry(h, w)
            return patch_ok(n, h, w, c, 1, 1, self.kernel.shape[3], self.stride, oh, ow)
        if self.k != 3 or self.stride != 1:
            return False
        _, _, _, oh, ow = self.geometry(h, w)
        return patch_ok(n, h, w, c, 3, 3, self.kernel.shape[3], 1, oh, ow)


# ----------------------------------------------------------------------------- dense
class _DenseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, relu):
        x, w = _c(x), _c(w)
        m, i = x.shape
        i2, o = w.shape
        if i != i2:
            raise _lib.EmbnetError(f"dense: input width {i} != kernel rows {i2}")
        y = torch.empty((m, o), device=x.device, dtype=torch.float32)
        lib = _lib.lib()
        nws = lib.embnet_dense_fwd_workspace_bytes(m, i, o)          # > 0: few output tiles, long reduction -> K split
        ws = workspace(nws, x.device) if nws else None
        check(lib.embnet_dense_fwd_f32(ptr(x), ptr(w), ptr(bias), ptr(y), m, i, o, int(relu), ptr(ws),
                                       ws.numel() * 4 if nws else 0, stream()))
        ctx.relu, ctx.has_bias, ctx.bias_ref = relu, bias is not None, bias
        ctx.save_for_backward(x, w, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        lib = _lib.lib()
        m, i = x.shape
        o = w.shape[1]
        dy = _c(dy)
        dx = dw = db = None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.relu:
            dz = torch.empty_like(dy)
            if want_db:
                db, note = _sink(ctx.bias_ref)
                ws = workspace(lib.embnet_colsum_workspace_bytes(m, o), x.device)
                check(lib.embnet_relu_bwd_colsum(ptr(dy), ptr(y), m, o, ptr(dz), ptr(db), ptr(ws), ws.numel() * 4, stream()))
                db, want_db = _done(db, note), False
            else:
                check(lib.embnet_relu_bwd(ptr(dy), ptr(y), dy.numel(), ptr(dz), stream()))
        else:
            dz = dy
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            check(lib.embnet_dense_dgrad_f32(ptr(dz), ptr(w), ptr(dx), m, i, o, stream()))
        if ctx.needs_input_grad[1]:
            dw, note = _sink(w)
            check(lib.embnet_dense_wgrad_f32(ptr(x), ptr(dz), ptr(dw), m, i, o, stream()))
            dw = _done(dw, note)
        if want_db:
            db, note = _sink(ctx.bias_ref)
            _colsum(dz, out=db)
            db = _done(db, note)
        return dx, dw, db, None


# The squeeze-and-excite gate sigmoid(Dense(swish(Dense(pooled)))) as one forward and two backward launches (csrc/se_mlp.hip)
# instead of twelve dense / activation / column-sum launches of 6-16 us each (EMBNET_SE_MLP=0: the composed form).
SE_MLP = [__import__("os").environ.get("EMBNET_SE_MLP", "1") == "1"]


class _SEMlpFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pooled, w1, b1, w2, b2):
        pooled, w1, w2 = _c(pooled), _c(w1), _c(w2)
        n, c = pooled.shape
        s = w1.shape[1]
        z1 = torch.empty((n, s), device=pooled.device, dtype=torch.float32)
        gate = torch.empty((n, c), device=pooled.device, dtype=torch.float32)
        check(_lib.lib().embnet_se_mlp_fwd(ptr(pooled), ptr(w1), ptr(b1), ptr(w2), ptr(b2), n, c, s, ptr(z1), ptr(gate), stream()))
        ctx.refs = (w1, b1, w2, b2)
        ctx.save_for_backward(pooled, w1, w2, z1, gate)
        return gate

    @staticmethod
    def backward(ctx, dgate):
        pooled, w1, w2, z1, gate = ctx.saved_tensors
        n, c = pooled.shape
        s = w1.shape[1]
        dgate = _c(dgate)
        dev = pooled.device
        outs, notes = [], []
        for i, ref in enumerate(ctx.refs):                 # dw1, db1, dw2, db2: the parameters' gradient sinks, or scratch
            if ctx.needs_input_grad[1 + i]:
                t, note = _sink(ref)
            else:
                t, note = torch.empty(ref.shape, device=dev, dtype=torch.float32), None
            outs.append(t); notes.append(note)
        dz1 = torch.empty_like(z1)
        dpooled = torch.empty_like(pooled)
        check(_lib.lib().embnet_se_mlp_bwd(ptr(dgate), ptr(gate), ptr(z1), ptr(pooled), ptr(w1), ptr(w2), n, c, s, ptr(dz1),
                                           ptr(dpooled), ptr(outs[0]), ptr(outs[1]), ptr(outs[2]), ptr(outs[3]), stream()))
        grads = [(_done(t, note) if ctx.needs_input_grad[1 + i] else None) for i, (t, note) in enumerate(zip(outs, notes))]
        return (dpooled if ctx.needs_input_grad[0] else None, *grads)


def se_mlp(pooled, reduce, expand):
    """sigmoid(expand(swish(reduce(pooled)))) for two Dense layers with bias — an MBConv block's squeeze-and-excite gate
    (reference backbones.py:84-98 via efficientnet's MBConv)."""
    if (SE_MLP[0] and pooled.dim() == 2 and pooled.is_cuda and not reduce.relu and not expand.relu
            and _lib.lib().embnet_se_mlp_supported(pooled.shape[0], pooled.shape[1], reduce.kernel.shape[1])):
        return _SEMlpFn.apply(pooled, reduce.kernel, reduce.bias, expand.kernel, expand.bias)
    return sigmoid(expand(swish(reduce(pooled))))


class Dense(nn.Module):
    def __init__(self, in_features, units, activation=None, l2=0.0, gen=None):
        super().__init__()
        if activation not in (None, "relu"):
            raise ValueError("Dense supports activation None or 'relu'")
        self.relu, self.l2 = activation == "relu", l2
        self.kernel = nn.Parameter(glorot_uniform_(torch.empty(in_features, units), gen))
        self.bias = nn.Parameter(torch.zeros(units))

    def forward(self, x):
        return _DenseFn.apply(x, self.kernel, self.bias, self.relu)


# ----------------------------------------------------------------------------- batch norm
def _bn_train_fwd(x, m, c, gamma, beta, eps, momentum, act, y, stats, moving_mean, moving_var, partials=None, with_range=False):
    """embnet_bn_train_fwd_ex; partials [2,C,P] = sums / sums of squares of x by row band from the producing conv's
    epilogue (Conv2D(..., emit_stats=True)), which then replace the statistics pass over x.
    stats with >= 5 rows: row 4 receives the per-channel bounds of |act(BN(x))| (nn_kernels.hip channel_bound); with_range (needs a
    sixth row and y): the apply pass folds them into the range slot of y — the first word of row 5 (_stats_range); a seventh row
    receives the per-channel bounds of |xhat| (what the backward needs to bound its dx without a dry run)."""
    lib = _lib.lib()
    ws = workspace(lib.embnet_bn_workspace_bytes(m, c), x.device)
    if partials is not None and tuple(partials.shape[:2]) != (2, c):
        raise _lib.EmbnetError(f"BatchNormalization: statistics partials {tuple(partials.shape)} for {c} channels")
    sp, sc = stats.data_ptr(), stats.shape[1]
    bound = sp + 16 * sc if stats.shape[0] >= 5 else None
    yr = sp + 20 * sc if (with_range and y is not None and stats.shape[0] >= 6) else None
    xh = sp + 24 * sc if stats.shape[0] >= 7 else None
    check(lib.embnet_bn_train_fwd_ex(ptr(x), m, c, ptr(gamma), ptr(beta), eps, momentum, int(act), ptr(y),
                                     sp, sp + 4 * sc, sp + 8 * sc, sp + 12 * sc,
                                     ptr(moving_mean), ptr(moving_var), ptr(partials),
                                     partials.shape[2] if partials is not None else 0, ptr(ws), ws.numel() * 4, bound, yr, xh, stream()))


def _stats_range(stats):
    """The range slot inside a BatchNormalization's statistics tensor (rows: mean, rstd, scale, shift, bound, [range word]): (owner,
    address) — see _rptr."""
    return (stats, stats.data_ptr() + 20 * stats.shape[1])


def _partials_of(x, training):
    p = getattr(x, "_bn_partials", None) if training else None
    return p if (p is not None and p.shape[1] == x.shape[-1]) else None


def _bn_grad_targets(ctx, c, device, gamma_idx=1, beta_idx=2):
    """Where a BatchNormalization backward writes dgamma / dbeta: the parameters' gradient sinks when a data-parallel
    reducer armed them (layers.GRAD_SINKS), scratch otherwise.  -> (dgamma tensor, dbeta tensor, finish) where
    finish() returns the (dgamma, dbeta) values to hand back to autograd."""
    want_g = ctx.has_gamma and ctx.needs_input_grad[gamma_idx]
    want_b = ctx.needs_input_grad[beta_idx]
    tg, ng = _sink(ctx.gamma_ref) if want_g else (torch.empty((c,), device=device, dtype=torch.float32), None)
    tb, nb = _sink(ctx.beta_ref) if want_b else (torch.empty((c,), device=device, dtype=torch.float32), None)

    def finish():
        return (_done(tg, ng) if want_g else None), (_done(tb, nb) if want_b else None)
    return tg, tb, finish


class _BatchNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, moving_mean, moving_var, eps, momentum, relu, training, partials=None,
                with_skip=False, emit_planes=False, emit_dx_planes=False, in_relu_bias=None, dropout=None,
                planes_only=False, dx_planes_only=False, emit_dx_range=False):
        """dropout=(rate, seed): a Dropout layer directly behind this one rides on its passes (embnet_affine_act_dropout,
        embnet_bn_bwd_inrelu_dropout; the mask and arithmetic of embnet_dropout).
        emit_planes: the output is ALSO written as bf16 planes for a patch conv (left in _ACT_PLANES under the output's
        address; BatchNormalization.forward hangs them on the tensor).  emit_dx_planes: backward writes dx also as planes
        into DY_PLANES (the producer of x is a patch conv, whose data gradient reads them).
        planes_only (with emit_planes): the fp32 output is NOT written — the returned tensor is a placeholder (_placeholder)
        whose only consumer, the caller promises, is a conv that reads the planes in all three of its passes.
        dx_planes_only (with emit_dx_planes): likewise for dx in backward — its only reader is the conv in front."""
        x = _c(x)
        lib = _lib.lib()
        c = x.shape[-1]
        m = x.numel() // c
        planes_only = bool(planes_only and emit_planes and not dropout)
        y = _placeholder(x.shape, x.device) if planes_only else torch.empty_like(x)
        # mean, rstd, scale, shift; in training also row 4 = the per-channel bounds of |y| and row 5 = y's range slot (first word)
        ranged = bool(training and c % 4 == 0 and not _BN_SCALAR)
        stats = torch.empty((7 if ranged else 4, c), device=x.device, dtype=torch.float32)     # (row 6: the bounds of |xhat|, for backward)
        yk = None if (emit_planes or dropout) else y                         # planes / dropout: statistics first, then one pass
        if training:
            _bn_train_fwd(x, m, c, gamma, beta, eps, momentum, relu, yk, stats, moving_mean, moving_var, partials, with_range=ranged)
        else:
            check(lib.embnet_bn_infer_fwd(ptr(x), m, c, ptr(gamma), ptr(beta), ptr(moving_mean), ptr(moving_var), eps,
                                          int(relu), ptr(yk), (stats.data_ptr() + 8 * stats.shape[1]), (stats.data_ptr() + 12 * stats.shape[1]), stream()))
        if emit_planes:
            planes = torch.empty(3 * x.numel(), device=x.device, dtype=torch.int16)
            # the planes' scale: from the bounds (training) — or, without them (inference statistics), from a dry run of the pass
            check(lib.embnet_affine_act_planes_ex(ptr(x), m, c, (stats.data_ptr() + 8 * c), (stats.data_ptr() + 12 * c), int(relu),
                                                  None if planes_only else ptr(y), ptr(planes),
                                                  (stats.data_ptr() + 16 * c) if ranged else None,
                                                  (stats.data_ptr() + 20 * c) if (ranged and not planes_only) else None, stream()))
            _ACT_PLANES[y.data_ptr()] = planes
        ctx.dropout = None
        if dropout:
            rate, seed = dropout
            ctx.dropout = (rate, seed, GRAPH_TICK)
            check(lib.embnet_affine_act_dropout(ptr(x), m, c, (stats.data_ptr() + 8 * c), (stats.data_ptr() + 12 * c), int(relu),
                                                rate, seed, GRAPH_TICK, ptr(y), stream()))
            if ranged:       # |dropout(a)| <= |a| / (1 - rate): the range word from the channel bounds (no apply pass folded them)
                check(lib.embnet_range_from_bound(stats.data_ptr() + 16 * c, c, 1.0 / (1.0 - float(rate)), None,
                                                  stats.data_ptr() + 20 * c, stream()))
        if ranged and not planes_only:
            _ACT_RANGE[y.data_ptr()] = _stats_range(stats)
        if training and FUSE_BN_SUMS[0] and c % 4 == 0 and not dropout:
            if len(_BN_FWD_STATS) > 64:
                _BN_FWD_STATS.clear()
            _BN_FWD_STATS[y.data_ptr()] = (x, stats, int(relu))
        ctx.emit_dx_range = bool(emit_dx_range) and c % 4 == 0 and training and not _BN_SCALAR
        ctx.emit_dx_planes = bool(emit_dx_planes) and c % 16 == 0
        ctx.dx_planes_only = bool(dx_planes_only) and ctx.emit_dx_planes and not with_skip
        ctx.in_relu_bias = in_relu_bias if (in_relu_bias is not None and not with_skip and c % 4 == 0) else None
        ctx.relu, ctx.training, ctx.has_gamma = relu, training, gamma is not None
        ctx.gamma_ref, ctx.beta_ref = gamma, beta
        ctx.save_for_backward(x, stats)
        if with_skip:                       # second output: x itself, for the identity shortcut (see backward)
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        """dskip: gradient of the pass-through copy of x (with_skip) — folded into the dx kernel instead of a
        separate autograd accumulation pass over the tensor."""
        x, stats = ctx.saved_tensors
        lib = _lib.lib()
        c = x.shape[-1]
        m = x.numel() // c
        dy = _c(dy)
        dskip = _c(dskip) if dskip is not None else None
        dskip_range = None
        if dskip is not None and DY_RANGE:
            # the identity shortcut ends here: nobody else will claim that gradient's range — it bounds what this pass adds to its dx
            e = DY_RANGE.pop(dskip.data_ptr(), None)
            if e is not None and e[1].shape == dskip.shape:
                dskip_range = e[0]
        xh = (stats.data_ptr() + 24 * c) if stats.shape[0] >= 7 else None     # bounds of |xhat| from the forward statistics
        only = getattr(ctx, "dx_planes_only", False) and dskip is None
        dx = _placeholder(x.shape, x.device) if only else torch.empty_like(x)
        dxp = None if only else ptr(dx)
        tg, tb, finish = _bn_grad_targets(ctx, c, x.device)
        ws = workspace(lib.embnet_bn_workspace_bytes(m, c), x.device)
        mean = stats.data_ptr() if ctx.training else None
        rstd = (stats.data_ptr() + 4 * stats.shape[1]) if ctx.training else None
        planes = None
        if getattr(ctx, "emit_dx_planes", False):
            planes = torch.empty(3 * x.numel(), device=x.device, dtype=torch.int16)
            if len(DY_PLANES) > 64:                  # entries nobody collected (a consumer fell back to the fp32 kernel)
                DY_PLANES.clear()
            DY_PLANES[dx.data_ptr()] = (planes, dx)
        drop = getattr(ctx, "dropout", None)
        inrelu = getattr(ctx, "in_relu_bias", None) is not None and planes is None and dskip is None
        hit = BN_SUMS.pop(dy.data_ptr(), None) if BN_SUMS else None
        if (hit is not None and hit[2] == x.data_ptr() and hit[3].shape == dy.shape and ctx.training and c % 4 == 0
                and not inrelu and drop is None):
            # dy is the data gradient of the conv behind this layer, which already produced the column sums
            dxr = None
            if getattr(ctx, "emit_dx_range", False) and not only and (planes is None or _planes_range_ok(x)):
                dxr = _emit_dx_range(dx)
            check(lib.embnet_bn_bwd_partials_ex(ptr(dy), ptr(x), m, c, mean, rstd, (stats.data_ptr() + 8 * c), (stats.data_ptr() + 12 * c),
                                                int(ctx.relu), ptr(hit[0]), hit[1], ptr(dskip), dxp, ptr(tg), ptr(tb), ptr(planes),
                                                ptr(dxr), int(hit[0].shape[0]), xh, ptr(dskip_range), stream()))
            dgamma, dbeta = finish()
            return (dx, dgamma, dbeta) + (None,) * 15
        if drop is not None and not inrelu:      # the Dropout's backward as a pass of its own in front of the BN backward
            dyd = torch.empty_like(dy)
            check(lib.embnet_dropout(ptr(dy), dy.numel(), drop[0], drop[1], drop[2], ptr(dyd), stream()))
            dy = dyd
        if inrelu:
            # x is the output of a conv with a fused ReLU: its backward (mask + bias gradient) rides on this pass
            (bias,) = ctx.in_relu_bias
            db, db_note = _sink(bias)
            # (the conv in front tagged its output `_wants_dy_range`: the exact range of dz rides on the pass, for its two gradients)
            dzr = _emit_dx_range(dx) if getattr(ctx, "emit_dx_range", False) else None
            if drop is not None:
                check(lib.embnet_bn_bwd_inrelu_dropout_ex(ptr(dy), ptr(x), m, c, mean, rstd, (stats.data_ptr() + 8 * stats.shape[1]),
                                                          (stats.data_ptr() + 12 * stats.shape[1]), int(ctx.relu), int(ctx.training),
                                                          drop[0], drop[1], drop[2], ptr(dx), ptr(tg), ptr(tb), ptr(db), ptr(ws),
                                                          ws.numel() * 4, ptr(dzr), stream()))
            else:
                check(lib.embnet_bn_bwd_inrelu_ex(ptr(dy), ptr(x), m, c, mean, rstd, (stats.data_ptr() + 8 * stats.shape[1]),
                                                  (stats.data_ptr() + 12 * stats.shape[1]), int(ctx.relu), int(ctx.training), ptr(dx),
                                                  ptr(tg), ptr(tb), ptr(db), ptr(ws), ws.numel() * 4, ptr(dzr), stream()))
            if len(RELU_DONE) > 64:
                RELU_DONE.clear()
            RELU_DONE[dx.data_ptr()] = (dx.detach(), db, db_note)
        else:
            dxr = None
            if getattr(ctx, "emit_dx_range", False) and not only and (planes is None or _planes_range_ok(x)):
                dxr = _emit_dx_range(dx)
            check(lib.embnet_bn_bwd_ex(ptr(dy), ptr(x), m, c, mean, rstd, (stats.data_ptr() + 8 * stats.shape[1]), (stats.data_ptr() + 12 * stats.shape[1]),
                                       int(ctx.relu), int(ctx.training), ptr(dskip), dxp, ptr(tg), ptr(tb), ptr(planes), ptr(ws),
                                       ws.numel() * 4, ptr(dxr), xh, ptr(dskip_range), stream()))
        dgamma, dbeta = finish()
        return (dx, dgamma, dbeta) + (None,) * 15


class _BNGapFn(torch.autograd.Function):
    """BatchNormalization (+ activation) that also returns the GlobalAveragePooling of its output, from the same pass
    (embnet_affine_act_gap): the squeeze-and-excite block pools the tensor this layer writes.  Training or inference
    statistics as _BatchNormFn; the pooled gradient is broadcast into dy in one kernel before the usual BN backward."""

    @staticmethod
    def forward(ctx, x, gamma, beta, moving_mean, moving_var, eps, momentum, act, training, partials=None, lazy_scale=False):
        """lazy_scale: the caller promises that the first output's ONLY consumer is channel_scale(y, s, lazy=True); that
        function then hands this layer the gradient of the gated tensor unscaled (GATE_PENDING) and backward applies s."""
        x = _c(x)
        lib = _lib.lib()
        n, c = x.shape[0], x.shape[-1]
        m = x.numel() // c
        ctx.lazy_scale = bool(lazy_scale)
        stats = torch.empty((4, c), device=x.device, dtype=torch.float32)   # mean, rstd, scale, shift
        if training:
            _bn_train_fwd(x, m, c, gamma, beta, eps, momentum, act, None, stats, moving_mean, moving_var, partials)
        else:
            check(lib.embnet_bn_infer_fwd(ptr(x), m, c, ptr(gamma), ptr(beta), ptr(moving_mean), ptr(moving_var), eps,
                                          int(act), None, (stats.data_ptr() + 8 * c), (stats.data_ptr() + 12 * c), stream()))
        y = torch.empty_like(x)
        g = torch.empty((n, c), device=x.device, dtype=torch.float32)
        check(lib.embnet_affine_act_gap(ptr(x), n, m // n, c, (stats.data_ptr() + 8 * c), (stats.data_ptr() + 12 * c), int(act),
                                        ptr(y), ptr(g), stream()))
        ctx.relu, ctx.training, ctx.has_gamma = act, training, gamma is not None
        ctx.gamma_ref, ctx.beta_ref = gamma, beta
        ctx.save_for_backward(x, stats)
        if lazy_scale:                                      # for channel_scale(lazy=True): this layer's input, statistics, activation
            _BN_FWD_STATS[y.data_ptr()] = (x, stats, int(act))
        return y, g

    @staticmethod
    def backward(ctx, dy, dg):
        x, stats = ctx.saved_tensors
        lib = _lib.lib()
        n, c = x.shape[0], x.shape[-1]
        m = x.numel() // c
        dy = _c(dy)
        dx = torch.empty_like(x)
        tg, tb, finish = _bn_grad_targets(ctx, c, x.device)
        ws = workspace(lib.embnet_bn_workspace_bytes(m, c), x.device)
        gate = None
        if getattr(ctx, "lazy_scale", False):
            ent = GATE_PENDING.pop(dy.data_ptr(), None)
            if ent is None or ent[1].shape != dy.shape:
                raise _lib.EmbnetError("BatchNormalization(lazy_scale=True): the gradient of the gated tensor did not arrive "
                                       "as channel_scale(lazy=True) left it — the BatchNormalization output has another consumer")
            gate = ent[0]
            if dg is None:
                dg = torch.zeros((n, c), device=x.device, dtype=torch.float32)
            if ent[2] is not None and ctx.training and m * (c // 4) < 2 ** 31 - 1:
                # channel_scale's backward already summed everything this layer's dbeta / dgamma need (embnet_se_bn_sums)
                check(lib.embnet_bn_bwd_gap_sums(ptr(dy), ptr(_c(dg)), ptr(gate), ptr(ent[2]), n, m // n, ptr(x), c, stats.data_ptr(),
                                                 stats.data_ptr() + 4 * c, stats.data_ptr() + 8 * c, stats.data_ptr() + 12 * c,
                                                 int(ctx.relu), ptr(dx), ptr(tg), ptr(tb), stream()))
                dgamma, dbeta = finish()
                return dx, dgamma, dbeta, None, None, None, None, None, None, None, None
        if dg is not None and ctx.training and (FUSE_GAP_BN[0] or gate is not None) and m * (c // 4) < 2 ** 31 - 1:
            # d(output) = dy (* gate) + dg / hw formed inside the two BatchNorm-backward passes: the summed tensor is never written
            check(lib.embnet_bn_bwd_gap(ptr(dy), ptr(_c(dg)), ptr(gate), n, m // n, ptr(x), c, stats.data_ptr(), stats.data_ptr() + 4 * c,
                                        stats.data_ptr() + 8 * c, stats.data_ptr() + 12 * c, int(ctx.relu), ptr(dx), ptr(tg), ptr(tb),
                                        ptr(ws), ws.numel() * 4, stream()))
            dgamma, dbeta = finish()
            return dx, dgamma, dbeta, None, None, None, None, None, None, None, None
        if gate is not None:                                # (not reached: lazy_scale is only granted in training mode)
            raise _lib.EmbnetError("BatchNormalization(lazy_scale=True) needs training-mode statistics")
        if dg is not None:                                  # d(output) = dy + dg / hw, written once
            dz = torch.empty_like(dy)
            check(lib.embnet_gap_bwd(ptr(_c(dg)), n, m // n, c, ptr(dy), ptr(dz), stream()))
            dy = dz
        mean = stats.data_ptr() if ctx.training else None
        rstd = (stats.data_ptr() + 4 * c) if ctx.training else None
        check(lib.embnet_bn_bwd(ptr(dy), ptr(x), m, c, mean, rstd, (stats.data_ptr() + 8 * c), (stats.data_ptr() + 12 * c),
                                int(ctx.relu), int(ctx.training), None, ptr(dx), ptr(tg), ptr(tb), None, ptr(ws),
                                ws.numel() * 4, stream()))
        dgamma, dbeta = finish()
        return dx, dgamma, dbeta, None, None, None, None, None, None, None, None


class _BNPoolFn(torch.autograd.Function):
    """Stage 1 of BatchNormalization.se_gate: statistics + the per-image channel means of act(BN(x)) — the activated tensor is
    NOT written (embnet_affine_act_gap with y = NULL).  Its backward runs after _BNScaleFn's (the gate depends on this output)
    and computes the whole layer's dx / dgamma / dbeta from what that left in POOL_PENDING (embnet_bn_bwd_gap_sums)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, moving_mean, moving_var, eps, momentum, act, partials, token):
        x = _c(x)
        lib = _lib.lib()
        n, c = x.shape[0], x.shape[-1]
        m = x.numel() // c
        stats = torch.empty((6 if c % 4 == 0 else 4, c), device=x.device, dtype=torch.float32)   # mean, rstd, scale, shift [, bound of |act(BN(x))|, range word]
        _bn_train_fwd(x, m, c, gamma, beta, eps, momentum, act, None, stats, moving_mean, moving_var, partials)
        g = torch.empty((n, c), device=x.device, dtype=torch.float32)
        check(lib.embnet_affine_act_gap(ptr(x), n, m // n, c, (stats.data_ptr() + 8 * c), (stats.data_ptr() + 12 * c), int(act),
                                        None, ptr(g), stream()))
        ctx.relu, ctx.has_gamma, ctx.token = act, gamma is not None, token
        ctx.gamma_ref, ctx.beta_ref = gamma, beta
        ctx.save_for_backward(x, stats)
        _BN_FWD_STATS[g.data_ptr()] = (x, stats, int(act))
        return g

    @staticmethod
    def backward(ctx, dpool):
        x, stats = ctx.saved_tensors
        lib = _lib.lib()
        n, c = x.shape[0], x.shape[-1]
        m = x.numel() // c
        ent = POOL_PENDING.pop(ctx.token, None)
        if ent is None:
            raise _lib.EmbnetError("BatchNormalization.se_gate: the gated tensor's gradient has not been computed before the pooled "
                                   "branch's — the gate must be a function of the pooled means, and the gated tensor must be used")
        dg, gate, sums = ent
        dx = torch.empty_like(x)
        tg, tb, finish = _bn_grad_targets(ctx, c, x.device)
        sp = stats.data_ptr()
        check(lib.embnet_bn_bwd_gap_sums(ptr(dg), ptr(_c(dpool)), ptr(gate), ptr(sums), n, m // n, ptr(x), c, sp, sp + 4 * c, sp + 8 * c,
                                         sp + 12 * c, int(ctx.relu), ptr(dx), ptr(tg), ptr(tb), stream()))
        dgamma, dbeta = finish()
        return dx, dgamma, dbeta, None, None, None, None, None, None, None


class _BNScaleFn(torch.autograd.Function):
    """Stage 2 of BatchNormalization.se_gate: out = act(BN(x)) * s[n,c] in one pass over x (embnet_affine_act_scale).  Backward:
    one pass over (d out, x) gives the gate's gradient and the per-(image, channel) sums of embnet_se_bn_sums; the gradient
    with respect to x is produced by _BNPoolFn.backward once the pooled branch's gradient is known."""

    @staticmethod
    def forward(ctx, x, s, stats, act, token):
        n, c = x.shape[0], x.shape[-1]
        hw = x.numel() // (n * c)
        s = _c(s)
        y = torch.empty_like(x)
        sp = stats.data_ptr()
        check(_lib.lib().embnet_affine_act_scale(ptr(x), n, hw, c, sp + 8 * c, sp + 12 * c, int(act), ptr(s), ptr(y), stream()))
        if stats.shape[0] >= 6:              # |act(BN(x)) * gate| <= the BatchNorm's output bound (the gate is a sigmoid): y's range slot
            check(_lib.lib().embnet_range_from_bound(sp + 16 * c, c, 1.0, None, sp + 20 * c, stream()))
            _ACT_RANGE[y.data_ptr()] = _stats_range(stats)
        ctx.act, ctx.token = int(act), token
        ctx.save_for_backward(x, s, stats)
        return y

    @staticmethod
    def backward(ctx, dg):
        x, s, stats = ctx.saved_tensors
        n, c = x.shape[0], x.shape[-1]
        hw = x.numel() // (n * c)
        dg = _c(dg)
        sums = torch.empty((n, 5, c), device=x.device, dtype=torch.float32)
        sp = stats.data_ptr()
        check(_lib.lib().embnet_se_bn_sums(ptr(dg), ptr(x), n, hw, c, sp, sp + 4 * c, sp + 8 * c, sp + 12 * c, ctx.act, ptr(sums), stream()))
        while len(POOL_PENDING) >= 8:
            POOL_PENDING.pop(next(iter(POOL_PENDING)))
        POOL_PENDING[ctx.token] = (dg, s, sums)
        return None, sums[:, 0, :], None, None, None


class _BNDropAddFn(torch.autograd.Function):
    """BatchNormalization (no activation) -> DropConnect (per-sample) -> Add(skip): the tail of an MBConv block with an identity
    shortcut, as one forward pass (embnet_affine_drop_add) and a BatchNorm backward that applies the drop factor while it reads
    the output gradient (embnet_bn_bwd_gap with gate = factor, dpool = 0); the skip's gradient IS the output gradient."""

    @staticmethod
    def forward(ctx, x, gamma, beta, moving_mean, moving_var, eps, momentum, partials, skip, rate, seed, skip_range=None):
        """skip_range: the range slot of `skip` (layers._range_of) -> the output gets one too: |y| <= |skip| + |BN(x)| / (1 - rate)."""
        x, skip = _c(x), _c(skip)
        lib = _lib.lib()
        n, c = x.shape[0], x.shape[-1]
        m = x.numel() // c
        stats = torch.empty((6 if skip_range is not None else 4, c), device=x.device, dtype=torch.float32)   # mean, rstd, scale, shift [, bound, range word]
        _bn_train_fwd(x, m, c, gamma, beta, eps, momentum, 0, None, stats, moving_mean, moving_var, partials)
        y = torch.empty_like(x)
        factor = torch.empty((n, c), device=x.device, dtype=torch.float32)
        check(lib.embnet_affine_drop_add(ptr(x), n, m // n, c, stats.data_ptr() + 8 * c, stats.data_ptr() + 12 * c, float(rate), seed,
                                         GRAPH_TICK, ptr(skip), ptr(y), ptr(factor), stream()))
        if skip_range is not None:
            sp = stats.data_ptr()
            check(lib.embnet_range_from_bound(sp + 16 * c, c, 1.0 / (1.0 - float(rate)), _rptr(skip_range), sp + 20 * c, stream()))
            _ACT_RANGE[y.data_ptr()] = _stats_range(stats)
        ctx.has_gamma, ctx.gamma_ref, ctx.beta_ref = gamma is not None, gamma, beta
        ctx.save_for_backward(x, stats, factor)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, stats, factor = ctx.saved_tensors
        lib = _lib.lib()
        n, c = x.shape[0], x.shape[-1]
        m = x.numel() // c
        dy = _c(dy)
        dx = torch.empty_like(x)
        tg, tb, finish = _bn_grad_targets(ctx, c, x.device)
        ws = workspace(lib.embnet_bn_workspace_bytes(m, c), x.device)
        zero = torch.zeros((n, c), device=x.device, dtype=torch.float32)
        sp = stats.data_ptr()
        check(lib.embnet_bn_bwd_gap(ptr(dy), ptr(zero), ptr(factor), n, m // n, ptr(x), c, sp, sp + 4 * c, sp + 8 * c, sp + 12 * c, 0,
                                    ptr(dx), ptr(tg), ptr(tb), ptr(ws), ws.numel() * 4, stream()))
        dgamma, dbeta = finish()
        return dx, dgamma, dbeta, None, None, None, None, None, dy, None, None, None


class Deferred:
    """A BatchNormalization(+activation) output that has not been written.  `raw` is the BN's input (as an
    autograd alias whose gradient is the gradient of the BN OUTPUT), `stats` [4,C] = mean, rstd, scale, shift,
    `act` the activation code.  Conv2D consumes it directly (include/embnet.h: in_scale/in_shift/in_act);
    anything else calls materialize()."""

    def __init__(self, raw, stats, act):
        self.raw, self.stats, self.act = raw, stats, act

    @property
    def shape(self):
        return self.raw.shape

    def materialize(self):
        return _AffineActFn.apply(self.raw, self.stats, self.act)


class _AffineActFn(torch.autograd.Function):
    """act(x*scale + shift) with the gradient handed to the alias unchanged (the BN node applies the chain)."""

    @staticmethod
    def forward(ctx, raw, stats, act):
        y = torch.empty_like(raw)
        c = raw.shape[-1]
        check(_lib.lib().embnet_affine_act(ptr(raw), raw.numel() // c, c, (stats.data_ptr() + 8 * stats.shape[1]), (stats.data_ptr() + 12 * stats.shape[1]),
                                           int(act), ptr(y), stream()))
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, None, None


class _BNDeferFn(torch.autograd.Function):
    """Statistics + scale/shift only; returns an alias of x whose incoming gradient is d(BN output)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, moving_mean, moving_var, eps, momentum, act, training, partials=None):
        x = _c(x)
        lib = _lib.lib()
        c = x.shape[-1]
        m = x.numel() // c
        stats = torch.empty((4, c), device=x.device, dtype=torch.float32)   # mean, rstd, scale, shift
        if training:
            _bn_train_fwd(x, m, c, gamma, beta, eps, momentum, act, None, stats, moving_mean, moving_var, partials)
        else:
            check(lib.embnet_bn_infer_fwd(ptr(x), m, c, ptr(gamma), ptr(beta), ptr(moving_mean), ptr(moving_var), eps,
                                          int(act), None, (stats.data_ptr() + 8 * stats.shape[1]), (stats.data_ptr() + 12 * stats.shape[1]), stream()))
        ctx.relu, ctx.training, ctx.has_gamma = int(act), training, gamma is not None
        ctx.gamma_ref, ctx.beta_ref = gamma, beta
        ctx.save_for_backward(x, stats)
        ctx.mark_non_differentiable(stats)
        return x.view_as(x), stats

    @staticmethod
    def backward(ctx, dy, _dstats):
        return _BatchNormFn.backward(ctx, dy)[:10]


class BatchNormalization(nn.Module):
    """Keras BatchNormalization on the last axis; `relu=True` fuses a following Activation('relu')."""

    def __init__(self, channels, epsilon=1e-3, momentum=0.99, scale=True, relu=False, activation=None):
        super().__init__()
        # fused activation code of the kernels: 0 none, 1 relu, 2 swish
        self.eps, self.momentum = epsilon, momentum
        self.relu = {None: int(bool(relu)), "relu": 1, "swish": 2}[activation]
        self.gamma = nn.Parameter(torch.ones(channels)) if scale else None
        self.beta = nn.Parameter(torch.zeros(channels))
        self.register_buffer("moving_mean", torch.zeros(channels))
        self.register_buffer("moving_variance", torch.ones(channels))
        self.frozen = False

    def freeze(self):
        """Keras `layer.trainable = False` on a BatchNormalization (TF2): no weight updates AND inference mode —
        the layer normalises with its moving statistics and stops updating them, whatever the model's mode."""
        self.frozen = True
        for p in self.parameters():
            p.requires_grad_(False)
        self.training = False
        return self

    def train(self, mode=True):
        return super().train(mode and not self.frozen)

    def drop_add(self, x, skip, drop=None):
        """Add(skip, DropConnect(BN(x))): the tail of an MBConv block with an identity shortcut (reference backbones.py:84-98;
        drop: a layers.DropConnect or None).  In training, for a linear BatchNormalization on 4-multiple channels, one forward
        pass and no drop-connect / Add passes in backward (_BNDropAddFn); otherwise the three layers one after the other."""
        c = x.shape[-1]
        if (FUSE_DROP_ADD[0] and self.training and self.relu == 0 and torch.is_grad_enabled() and x.dim() == 4 and c % 4 == 0
                and x.numel() // 4 < 2 ** 31 - 1 and skip.shape == x.shape):
            rate, seed = 0.0, 0
            if drop is not None and drop.training and drop.enabled and drop.rate > 0:
                drop._step += 1
                rate, seed = drop.rate, (drop.seed << 32) + drop._step
            y = _BNDropAddFn.apply(x, self.gamma, self.beta, self.moving_mean, self.moving_variance, self.eps, self.momentum,
                                   _partials_of(x, True), skip, rate, seed, _range_of(skip))
            rng = _ACT_RANGE.pop(y.data_ptr(), None)
            if rng is not None:
                y._range = rng
            return y
        y = self(x)
        if drop is not None:
            y = drop(y)
        return add(y, skip)

    def se_gate(self, x, gate_fn):
        """act(BN(x)) * gate_fn(mean over the pixels of act(BN(x))): the squeeze-and-excite gating of an MBConv block
        (reference backbones.py:84-98) with THIS layer's output as its only input.  gate_fn maps the pooled means [n,c] to the
        gate [n,c] and must depend on them.  In training the activated tensor is never written: one pass over x for the
        pooled means (4 B per element), one for the gated output (8 B), and in backward one for the gate's gradient and the
        BatchNorm sums (8 B), one for dx (12 B)."""
        return _se_gate(self, x, gate_fn)

    def forward(self, x, defer=False, with_skip=False, emit_gap=False, planes_for=None, dropout=None, lazy_scale=False,
                sole=False, owns_input=False):
        """sole=True (with planes_for): the caller promises that `planes_for` is the ONLY consumer of the output; when that conv
        reads planes in all three passes (Conv2D.planes_only_gradient) the fp32 output is never written (a placeholder is returned).
        owns_input=True: the caller promises that this layer is the ONLY consumer of x; when x's producer takes its gradients
        from planes alone (`x._dy_planes_only`) backward writes dx as planes only.
        dropout=<Dropout>: the Dropout layer that consumes the output, applied in this layer's passes when it is active
        (plain path only; the caller then skips the Dropout module).
        planes_for=<Conv2D>: the conv that consumes the output; when it can run the patch kernel on it
        (Conv2D.patch_capable) the output is also written as bf16 planes (`y._planes`) in the same pass.
        emit_gap=True (C % 4 == 0): returns (bn(x), GlobalAveragePooling2D(bn(x))) from one pass over the tensor.
        defer=True (consumers are Conv2D layers): only the statistics are computed; the convs apply the
        affine + activation while gathering their input, and the normalised tensor is never written.
        with_skip=True: returns (bn(x), x) — use the second value for the identity shortcut that also consumes x,
        so that its gradient is added inside the BN backward kernel instead of by an autograd accumulation pass."""
        if emit_gap:
            if x.dim() == 4 and x.shape[-1] % 4 == 0:
                # lazy_scale (emit_gap only): see _BNGapFn.forward; granted when this layer can apply the gate in its backward
                lazy = bool(lazy_scale and self.training and torch.is_grad_enabled() and FUSE_GATE_BN[0]
                            and x.numel() // 4 < 2 ** 31 - 1)
                y, g = _BNGapFn.apply(x, self.gamma, self.beta, self.moving_mean, self.moving_variance, self.eps,
                                      self.momentum, self.relu, self.training, _partials_of(x, self.training), lazy)
                if lazy:
                    y._lazy_scale_ok = _BN_FWD_STATS.pop(y.data_ptr())
                return y, g
            y = self.forward(x)
            return y, _GapFn.apply(y)
        want_dx_planes = bool(getattr(x, "_wants_dy_planes", False)) and torch.is_grad_enabled()
        want_dx_range = bool(getattr(x, "_wants_dy_range", False)) and torch.is_grad_enabled()
        dx_only = bool(want_dx_planes and owns_input and PLANES_ONLY[0] and getattr(x, "_dy_planes_only", False)
                       and x.shape[-1] % 16 == 0)

        def tag(y):          # (x, stats, act) for a conv behind this layer whose data gradient can emit the backward sums (BN_SUMS)
            src = _BN_FWD_STATS.pop(y.data_ptr(), None)
            if src is not None and torch.is_grad_enabled():
                y._bn_src = src
            rng = _ACT_RANGE.pop(y.data_ptr(), None)       # the range slot of y (an upper bound of max |y|) for the convs that read it
            if rng is not None:
                y._range = rng

        if planes_for is not None and not defer and planes_for.patch_capable(x.shape):
            y_only = bool(sole and PLANES_ONLY[0] and planes_for.planes_only_input(x.shape))
            out = _BatchNormFn.apply(x, self.gamma, self.beta, self.moving_mean, self.moving_variance, self.eps,
                                     self.momentum, self.relu, self.training, _partials_of(x, self.training), with_skip,
                                     True, want_dx_planes, None, None, y_only, dx_only, want_dx_range)
            y = out[0] if with_skip else out
            y._planes = _ACT_PLANES.pop(y.data_ptr())
            tag(y)
            return out
        if with_skip and not defer:
            out = _BatchNormFn.apply(x, self.gamma, self.beta, self.moving_mean, self.moving_variance, self.eps,
                                     self.momentum, self.relu, self.training, _partials_of(x, self.training), True,
                                     False, want_dx_planes, None, None, False, False, want_dx_range)
            tag(out[0])
            return out
        if with_skip:
            return self.forward(x, defer=True), x
        if defer and x.shape[-1] % 4 == 0:
            raw, stats = _BNDeferFn.apply(x, self.gamma, self.beta, self.moving_mean, self.moving_variance, self.eps,
                                          self.momentum, self.relu, self.training, _partials_of(x, self.training))
            return Deferred(raw, stats, self.relu)
        in_relu_bias = getattr(x, "_relu_conv", None) if (self.training and torch.is_grad_enabled() and FUSE_RELU_BN[0]) else None
        y = _BatchNormFn.apply(x, self.gamma, self.beta, self.moving_mean, self.moving_variance, self.eps,
                               self.momentum, self.relu, self.training, _partials_of(x, self.training), False,
                               False, want_dx_planes, in_relu_bias, dropout.take() if dropout is not None else None,
                               False, dx_only and in_relu_bias is None and dropout is None, want_dx_range)
        tag(y)
        return y


def _se_gate(bn, x, gate_fn):
    """BatchNormalization.se_gate (below)."""
    c = x.shape[-1]
    if (SE_TWO_STAGE[0] and SE_BN_SUMS[0] and FUSE_GATE_BN[0] and bn.training and torch.is_grad_enabled() and x.dim() == 4
            and c % 4 == 0 and x.numel() // 4 < 2 ** 31 - 1 and x.requires_grad):
        token = object()
        pooled = _BNPoolFn.apply(x, bn.gamma, bn.beta, bn.moving_mean, bn.moving_variance, bn.eps, bn.momentum, bn.relu,
                                 _partials_of(x, True), token)
        bn_x, stats, act = _BN_FWD_STATS.pop(pooled.data_ptr())
        s = gate_fn(pooled)
        y = _BNScaleFn.apply(bn_x, s.reshape(x.shape[0], c), stats, act, token)
        rng = _ACT_RANGE.pop(y.data_ptr(), None)
        if rng is not None:
            y._range = rng
        return y
    y, g = bn(x, emit_gap=True, lazy_scale=True)
    return channel_scale(y, gate_fn(g), lazy=True)


class _InputBNConvFn(torch.autograd.Function):
    """BN(scale=False) on the raw image followed by a conv (the zoo ResNet stem: bn_data -> conv0).
    The image needs no gradient and the only trainable BN weight is beta, whose gradient is
    sum(d conv-input) = contract(kernel, per-tap sums of dy); the per-tap sums are a weight-gradient
    of an all-ones one-channel image.  This replaces a 3-channel data-gradient conv over the full
    224x224 map (N=3 of a 32-wide MFMA tile, 3/4 of the taps structurally zero at stride 2)."""

    _ones = {}

    @staticmethod
    def forward(ctx, x, beta, moving_mean, moving_var, w, eps, momentum, geom, out_stats=None, zero_sum_dy=False, w_range=None,
                stem=False):
        """w_range: the kernel's range slot (_Conv2dFn.forward); the channel-padded copy has the same range.
        stem: the forward runs csrc/conv_stem.hip (input_bn_conv decided: 7x7 / 2, 4 -> 64 channels, both ranges known)."""
        x, w = _c(x), _c(w)
        lib = _lib.lib()
        n, h, wd, c = x.shape
        r, s, _, k = w.shape
        stride, pt, pl, oh, ow = geom
        m = n * h * wd
        cp = (c + 3) // 4 * 4                      # widen 3 -> 4 channels: 16-byte gathers in the stem conv
        if cp != c:
            xp = torch.empty((n, h, wd, cp), device=x.device, dtype=torch.float32)
            check(lib.embnet_pad_channels(ptr(x), m, c, cp, ptr(xp), stream()))
            # persistent padded copies of the three BN vectors and of the kernel (zero taps / zero offset for the pad
            # channel), refreshed by plain copies: no cat / pad / fill launches per step
            # (kept on the kernel Parameter object, like its planes: an address-keyed cache would hand a new model the pads
            # of a freed one whose storage it inherited, and never evict)
            pads = getattr(w, "_embnet_stem_pads", None)
            if pads is None or pads[3].shape != (r, s, cp, k) or pads[3].device != x.device:
                pads = (torch.zeros(cp, device=x.device), torch.zeros(cp, device=x.device),
                        torch.ones(cp, device=x.device), torch.zeros((r, s, cp, k), device=x.device))
                try:
                    w._embnet_stem_pads = pads
                except AttributeError:
                    pass
            beta_p, mm_p, mv_p, w_p = pads
            beta_p[:c].copy_(beta.detach()); mm_p[:c].copy_(moving_mean); mv_p[:c].copy_(moving_var)
            w_p[:, :, :c, :].copy_(w.detach())
        else:
            xp, beta_p, mm_p, mv_p, w_p = x, beta, moving_mean, moving_var, w
        a = torch.empty_like(xp)
        ranged = w_range is not None and cp % 4 == 0
        stats = torch.empty((6 if ranged else 4, cp), device=x.device, dtype=torch.float32)
        _bn_train_fwd(xp, m, cp, None, beta_p, eps, momentum, 0, a, stats, mm_p, mv_p, with_range=ranged)
        a_range = _stats_range(stats) if ranged else None
        if cp != c:                                # the pad channel is identically 0 after BN (x=0, beta=0)
            moving_mean.copy_(mm_p[:c])
            moving_var.copy_(mv_p[:c])
        y = torch.empty((n, oh, ow, k), device=x.device, dtype=torch.float32)
        if stem:
            if a_range is None:
                raise _lib.EmbnetError("input_bn_conv: the stem kernel needs the ranges of both operands")
            check(lib.embnet_conv2d_stem_f32(ptr(a), ptr(w_p), ptr(y), n, h, wd, pt, pl, oh, ow, ptr(out_stats), _rptr(a_range),
                                             _rptr(w_range), stream()))
        else:
            cws = workspace(lib.embnet_conv2d_fwd_workspace_bytes(n, cp, r, s, k, oh, ow), x.device)
            check(lib.embnet_conv2d_fwd_f32_ex(
                ptr(a), ptr(w_p), None, ptr(y), n, h, wd, cp, r, s, k, stride, pt, pl, oh, ow, 0, None, None, None, 0,
                ptr(out_stats), ptr(cws), cws.numel() * 4, _rptr(a_range), _rptr(w_range) if a_range is not None else None, stream()))
        ctx.w_range = w_range
        ctx.a_range = a_range
        ctx.geom, ctx.c, ctx.zero_sum_dy = geom, c, bool(zero_sum_dy)
        ctx.beta_ref = beta
        ctx.save_for_backward(a, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        a, w = ctx.saved_tensors
        lib = _lib.lib()
        n, h, wd, cp = a.shape
        r, s, c, k = w.shape
        stride, pt, pl, oh, ow = ctx.geom
        dy = _c(dy)
        dw_p = torch.empty((r, s, cp, k), device=a.device, dtype=torch.float32)
        ws = workspace(lib.embnet_conv2d_wgrad_workspace_bytes(n, cp, r, s, k, oh, ow), a.device)
        a_range = getattr(ctx, "a_range", None)
        dy_range = _take_dy_range(dy) if (getattr(ctx, "w_range", None) is not None and a_range is not None) else None
        check(lib.embnet_conv2d_wgrad_f32_ex(
            ptr(a), ptr(dy), ptr(dw_p), ptr(ws), ws.numel() * 4, n, h, wd, cp, r, s, k, stride, pt, pl, oh, ow,
            None, None, 0, _rptr(a_range) if dy_range is not None else None, _rptr(dy_range), stream()))
        if cp == c:
            dw = dw_p
        else:
            dw, dw_note = _sink(w)
            dw.copy_(dw_p[:, :, :c, :])
            dw = _done(dw, dw_note)
        taps = torch.empty((r, s, 1, k), device=a.device, dtype=torch.float32)
        if ctx.zero_sum_dy:
            # dy is the data gradient of a training-mode BatchNormalization (sums to zero per channel): the per-tap sums
            # are minus the border strips' sums — 6 % of the tensor instead of a pass over all of it
            ws = workspace(max(lib.embnet_tap_border_sums_workspace_bytes(n, oh, ow, k, r, s, stride, pt, pl, h, wd), 16), a.device)
            check(lib.embnet_tap_border_sums(ptr(dy), n, oh, ow, k, r, s, stride, pt, pl, h, wd, ptr(taps), ptr(ws),
                                             ws.numel() * 4, stream()))
        else:
            key = (a.device, n, h, wd)
            ones = _InputBNConvFn._ones.get(key)
            if ones is None:
                ones = _InputBNConvFn._ones[key] = torch.ones((n, h, wd, 1), device=a.device, dtype=torch.float32)
            ws = workspace(lib.embnet_conv2d_wgrad_workspace_bytes(n, 1, r, s, k, oh, ow), a.device)
            check(lib.embnet_conv2d_wgrad_f32(
                ptr(ones), ptr(dy), ptr(taps), ptr(ws), ws.numel() * 4, n, h, wd, 1, r, s, k, stride, pt, pl, oh, ow,
                None, None, 0, stream()))
        dbeta, db_note = _sink(ctx.beta_ref)
        check(lib.embnet_tap_contract(ptr(w), ptr(taps), r * s, c, k, ptr(dbeta), stream()))
        return None, _done(dbeta, db_note), None, None, dw, None, None, None, None, None, None, None


def input_bn_conv(x, bn, conv, emit_stats=False, zero_sum_dy=False):
    """bn (scale=False, no relu) then conv (no bias / activation) on an image batch.
    emit_stats: as Conv2D.forward (the BatchNormalization behind the conv gets its sums from the epilogue).
    zero_sum_dy: the caller vouches that the conv output's ONLY consumer is a BatchNormalization in training mode, whose
    data gradient sums to zero per channel — bn's beta gradient then needs the border strips of dy only."""
    fusable = (bn.training and not x.requires_grad and bn.gamma is None and not bn.relu and conv.bias is None
               and not conv.relu and torch.is_grad_enabled())
    if not fusable:
        return conv(bn(x), emit_stats=emit_stats)
    geom = conv.geometry(x.shape[1], x.shape[2])
    r, s, c, k = conv.kernel.shape
    cp = c if c % 4 == 0 else c + 4 - c % 4                         # the fused stem pads the image channels
    w_range = None
    if CONV_F16[0] and getattr(conv, "f16", False) and conv.kernel.shape[3] % 4 == 0 and not _BN_SCALAR:
        w_range = weight_range(conv.kernel)
    # the zoo ResNets' 7x7 / 2 stem on its own kernel (csrc/conv_stem.hip: every input pixel once per tile instead of 49 gathers per
    # output pixel) when both operands carry a range
    stem = bool(STEM_CONV[0] and w_range is not None
                and _lib.lib().embnet_conv2d_stem_supported(x.shape[0], x.shape[1], x.shape[2], cp, r, s, k, geom[0], geom[1], geom[2],
                                                            geom[3], geom[4]))
    out_stats = None
    if emit_stats:
        if stem:
            rows = _lib.lib().embnet_conv2d_stem_stats_rows(x.shape[0], geom[3], geom[4])
        else:
            rows = _lib.lib().embnet_conv2d_fwd_stats_rows(x.shape[0], cp, r, s, k, geom[3], geom[4])
        if rows > 0:
            out_stats = torch.empty((2, k, rows), device=x.device, dtype=torch.float32)
    y = _InputBNConvFn.apply(x, bn.beta, bn.moving_mean, bn.moving_variance, conv.kernel, bn.eps, bn.momentum, geom,
                             out_stats, zero_sum_dy, w_range, stem)
    if out_stats is not None:
        y._bn_partials = out_stats
    if w_range is not None:
        y._wants_dy_range = True
    return y


# ----------------------------------------------------------------------------- pooling
class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, stride, pad, relu_bias=None, y_range=None, want_dz_range=False):
        """relu_bias: the bias of the Conv2D (fused ReLU) whose output x is, when that conv's ReLU backward and bias gradient
        are to ride on this layer's backward (FUSE_RELU_POOL) — x is then kept for backward (the conv keeps it anyway).
        y_range: a range slot that receives the exact max |y| (MaxPool2D.emit_range); want_dz_range (with relu_bias): backward
        leaves the range of dz in DY_RANGE for the conv in front (it tagged x `_wants_dy_range`)."""
        x = _c(x)
        n, h, w, c = x.shape
        oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        if oh <= 0 or ow <= 0:
            raise _lib.EmbnetError(f"MaxPool {k}x{k}/{stride} does not fit a {h}x{w} input")
        y = torch.empty((n, oh, ow, c), device=x.device, dtype=torch.float32)
        arg = torch.empty((n, oh, ow, c), device=x.device, dtype=torch.uint8)
        check(_lib.lib().embnet_maxpool_fwd_ex(ptr(x), n, h, w, c, k, stride, pad, oh, ow, ptr(y), ptr(arg), ptr(y_range), stream()))
        ctx.cfg = (n, h, w, c, k, stride, pad, oh, ow)
        ctx.relu_bias = relu_bias
        ctx.want_dz_range = bool(want_dz_range) and relu_bias is not None
        if relu_bias is not None:
            ctx.save_for_backward(arg, x)
        else:
            ctx.save_for_backward(arg)
        return y

    @staticmethod
    def backward(ctx, dy):
        arg = ctx.saved_tensors[0]
        n, h, w, c, k, stride, pad, oh, ow = ctx.cfg
        dy = _c(dy)
        dx = torch.empty((n, h, w, c), device=dy.device, dtype=torch.float32)
        lib = _lib.lib()
        if ctx.relu_bias is not None:
            # x is the ReLU output of the conv in front: the mask and the bias gradient ride on this pass and the conv's
            # backward finds its incoming gradient in RELU_DONE (as behind a BatchNormalization)
            x = ctx.saved_tensors[1]
            db, db_note = _sink(ctx.relu_bias)
            ws = workspace(lib.embnet_bn_workspace_bytes(n * h * w, c), dy.device)
            dzr = _emit_dx_range(dx) if getattr(ctx, "want_dz_range", False) else None
            check(lib.embnet_maxpool_relu_bwd_colsum_ex(ptr(dy), ptr(arg), ptr(x), n, h, w, c, k, stride, pad, oh, ow, ptr(dx), ptr(db),
                                                        ptr(ws), ws.numel() * 4, ptr(dzr), stream()))
            if len(RELU_DONE) > 64:
                RELU_DONE.clear()
            RELU_DONE[dx.data_ptr()] = (dx.detach(), db, db_note)
            return (dx,) + (None,) * 6
        check(lib.embnet_maxpool_bwd(ptr(dy), ptr(arg), n, h, w, c, k, stride, pad, oh, ow, ptr(dx), stream()))
        return (dx,) + (None,) * 6


class MaxPool2D(nn.Module):
    """Keras MaxPool2D(); zero_pad=p reproduces ZeroPadding2D(p) + 'valid' pooling (pads with 0)."""

    def __init__(self, pool_size=2, strides=None, zero_pad=0):
        super().__init__()
        self.k, self.s, self.p = pool_size, strides or pool_size, zero_pad
        # True: the output carries its exact range (`_range`) for a three-product conv behind it — set by the model builder where
        # no BatchNormalization bounds the activation (the `simple` backbone's conv -> ReLU -> pool blocks)
        self.emit_range = False

    def forward(self, x):
        rc = getattr(x, "_relu_conv", None) if (FUSE_RELU_POOL[0] and torch.is_grad_enabled() and x.requires_grad) else None
        fused = rc is not None and x.shape[3] % 4 == 0 and x.shape[0] * x.shape[1] * x.shape[2] < 2 ** 31
        slot = None
        if _range_of(x) is None and self.emit_range and CONV_F16[0] and x.shape[3] % 4 == 0 and not _BN_SCALAR:
            slot = _new_range_slot(x.device)
        want = bool(fused and getattr(x, "_wants_dy_range", False))
        y = _MaxPoolFn.apply(x, self.k, self.s, self.p, rc[0] if fused else None, slot, want)
        # (zero padding: a window's maximum is one of x's values or a padding zero, so max |y| <= max |x|)
        rng = slot if slot is not None else _range_of(x)
        if rng is not None:
            y._range = rng
        return y


class _BNActMaxPoolFn(torch.autograd.Function):
    """pool(act(bn(x))) without materialising the activation or its gradient (include/embnet.h:
    embnet_bn_act_maxpool_fwd/bwd) — the zoo ResNet stem bn0 -> relu -> ZeroPadding2D(1) -> MaxPool(3,2)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, moving_mean, moving_var, eps, momentum, act, training, k, stride, pad,
                partials=None, emit_dx_range=False):
        x = _c(x)
        lib = _lib.lib()
        n, h, w, c = x.shape
        m = n * h * w
        ctx.emit_dx_range = bool(emit_dx_range) and training
        oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
        if oh <= 0 or ow <= 0:
            raise _lib.EmbnetError(f"MaxPool {k}x{k}/{stride} does not fit a {h}x{w} input")
        stats = torch.empty((4, c), device=x.device, dtype=torch.float32)   # mean, rstd, scale, shift
        if training:
            _bn_train_fwd(x, m, c, gamma, beta, eps, momentum, act, None, stats, moving_mean, moving_var, partials)
        else:
            check(lib.embnet_bn_infer_fwd(ptr(x), m, c, ptr(gamma), ptr(beta), ptr(moving_mean), ptr(moving_var), eps,
                                          int(act), None, (stats.data_ptr() + 8 * stats.shape[1]), (stats.data_ptr() + 12 * stats.shape[1]), stream()))
        y = torch.empty((n, oh, ow, c), device=x.device, dtype=torch.float32)
        arg = torch.empty((n, oh, ow, c), device=x.device, dtype=torch.uint8)
        # training: keep the BN input at every window's winner, so backward's dgamma/dbeta sums stream instead of gathering
        xwin = torch.empty_like(y) if training else None
        check(lib.embnet_bn_act_maxpool_fwd(ptr(x), n, h, w, c, (stats.data_ptr() + 8 * stats.shape[1]), (stats.data_ptr() + 12 * stats.shape[1]), int(act), k,
                                            stride, pad, oh, ow, ptr(y), ptr(arg), ptr(xwin) if training else None, stream()))
        ctx.cfg = (n, h, w, c, k, stride, pad, oh, ow, int(act), training, gamma is not None)
        ctx.has_gamma, ctx.gamma_ref, ctx.beta_ref = gamma is not None, gamma, beta
        if training:
            ctx.save_for_backward(x, stats, arg, xwin)
        else:
            ctx.save_for_backward(x, stats, arg)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, h, w, c, k, stride, pad, oh, ow, act, training, has_gamma = ctx.cfg
        x, stats, arg = ctx.saved_tensors[:3]
        xwin = ctx.saved_tensors[3] if training else None
        lib = _lib.lib()
        dy = _c(dy)
        dx = torch.empty_like(x)
        tg, tb, finish = _bn_grad_targets(ctx, c, x.device)
        ws = workspace(lib.embnet_bn_act_maxpool_bwd_workspace_bytes(n, oh, ow, c), x.device)
        mean = stats.data_ptr() if training else None
        rstd = (stats.data_ptr() + 4 * stats.shape[1]) if training else None
        dxr = _emit_dx_range(dx) if getattr(ctx, "emit_dx_range", False) else None
        check(lib.embnet_bn_act_maxpool_bwd_ex(ptr(dy), ptr(arg), ptr(x), n, h, w, c, k, stride, pad, oh, ow, mean, rstd,
                                               (stats.data_ptr() + 8 * stats.shape[1]), (stats.data_ptr() + 12 * stats.shape[1]), act, int(training),
                                               ptr(xwin) if training else None, ptr(dx),
                                               ptr(tg), ptr(tb), ptr(ws), ws.numel() * 4, ptr(dxr), stream()))
        dgamma, dbeta = finish()
        return (dx, dgamma, dbeta) + (None,) * 11


def bn_act_maxpool(x, bn, pool):
    """pool(bn(x)) for a BatchNormalization (with its fused activation) followed by a MaxPool2D; one fused
    pass when the channel count allows 16-byte lanes, the two layers otherwise."""
    if x.shape[-1] % 4:
        return pool(bn(x))
    return _BNActMaxPoolFn.apply(x, bn.gamma, bn.beta, bn.moving_mean, bn.moving_variance, bn.eps, bn.momentum,
                                 bn.relu, bn.training, pool.k, pool.s, pool.p, _partials_of(x, bn.training),
                                 bool(getattr(x, "_wants_dy_range", False)) and torch.is_grad_enabled())


class _GapFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, with_skip=False):
        x = _c(x)
        n, h, w, c = x.shape
        y = torch.empty((n, c), device=x.device, dtype=torch.float32)
        check(_lib.lib().embnet_gap_fwd(ptr(x), n, h * w, c, ptr(y), stream()))
        ctx.shape = (n, h, w, c)
        if with_skip:                       # second output: x itself, for the tensor's other consumer (see backward)
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dskip=None):
        """dskip: gradient of the pass-through copy of x (with_skip) — added while the pooled gradient is broadcast,
        instead of a broadcast tensor plus an autograd accumulation pass."""
        n, h, w, c = ctx.shape
        dy = _c(dy)
        dskip = _c(dskip) if (dskip is not None and c % 4 == 0) else dskip
        dx = torch.empty(ctx.shape, device=dy.device, dtype=torch.float32)
        fold = dskip is not None and c % 4 == 0
        check(_lib.lib().embnet_gap_bwd(ptr(dy), n, h * w, c, ptr(dskip) if fold else None, ptr(dx), stream()))
        if dskip is not None and not fold:
            dx = dx + dskip
        return dx, None


class GlobalAveragePooling2D(nn.Module):
    def forward(self, x, with_skip=False):
        """with_skip=True: returns (gap(x), x) — hand the second value to x's other consumer (squeeze-and-excite: the
        channel scaling), so that its gradient is added inside the pooling's backward kernel."""
        return _GapFn.apply(x, with_skip)


class Flatten(nn.Module):
    """(h, w, c) order — a view on NHWC, no kernel."""

    def forward(self, x):
        return x.reshape(x.shape[0], -1)


# ----------------------------------------------------------------------------- elementwise
class _AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        y = torch.empty_like(a)
        check(_lib.lib().embnet_add(ptr(a), ptr(b), a.numel(), ptr(y), stream()))
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add(a, b):
    if a.shape != b.shape:
        raise _lib.EmbnetError(f"Add: shapes differ {tuple(a.shape)} vs {tuple(b.shape)}")
    return _AddFn.apply(a, b)


# Device address of a uint64 "steps since capture" counter while a training step is being captured into a HIP graph
# (train_step.TripletTrainer sets it): the dropout kernels add it to their seed, so a replay draws the mask an eager
# step would have drawn at that step.  None outside a capture.
GRAPH_TICK = None


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rate, seed):
        x = _c(x)
        y = torch.empty_like(x)
        ctx.rate, ctx.seed, ctx.tick = rate, seed, GRAPH_TICK
        check(_lib.lib().embnet_dropout(ptr(x), x.numel(), rate, seed, ctx.tick, ptr(y), stream()))
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        dx = torch.empty_like(dy)
        check(_lib.lib().embnet_dropout(ptr(dy), dy.numel(), ctx.rate, ctx.seed, ctx.tick, ptr(dx), stream()))
        return dx, None, None


class Dropout(nn.Module):
    """Keras Dropout (inverted scaling, training only).  `enabled=False` turns it off for parity runs."""

    def __init__(self, rate, seed=0):
        super().__init__()
        self.rate, self.seed, self.enabled, self._step = rate, seed, True, 0

    def active(self):
        return bool(self.training and self.enabled and self.rate > 0)

    def take(self):
        """(rate, seed) of this call's mask for a layer that applies the dropout in its own passes (BatchNormalization
        (dropout=...)); None when the layer is inactive.  Advances the mask counter exactly as forward() does."""
        if not self.active():
            return None
        self._step += 1
        return self.rate, (self.seed << 32) + self._step

    def forward(self, x):
        if not self.active():
            return x
        self._step += 1
        y = _DropoutFn.apply(x, self.rate, (self.seed << 32) + self._step)
        r = _range_of(x)
        if r is not None:          # |dropout(x)| <= |x| / (1 - rate): the input's range, scaled (as a BatchNormalization with a fused Dropout does)
            slot = _new_range_slot(x.device)
            check(_lib.lib().embnet_range_from_bound(_rptr(r), 1, 1.0 / (1.0 - float(self.rate)), None, ptr(slot), stream()))
            y._range = slot
        return y


# ----------------------------------------------------------------------------- MBConv pieces
class _DepthwiseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, geom, out_stats=None, bn_src=None):
        """out_stats [2,C,P] (P = embnet_dwconv2d_fwd_stats_rows): the kernel also writes the per-channel sums of y, the
        statistics partials of the BatchNormalization that follows (DepthwiseConv2D(emit_stats=True)).
        bn_src = (bn_x, bn_stats, bn_act): x is act(BatchNorm(bn_x)) — the stride-1 data gradient also emits that layer's
        backward sums (BN_SUMS, as _Conv2dFn)."""
        x, w = _c(x), _c(w)
        n, h, wd, c = x.shape
        r, s = w.shape[0], w.shape[1]
        stride, pt, pl, oh, ow = geom
        y = torch.empty((n, oh, ow, c), device=x.device, dtype=torch.float32)
        if out_stats is not None:
            check(_lib.lib().embnet_dwconv2d_fwd_stats_f32(ptr(x), ptr(w), ptr(y), n, h, wd, c, r, s, stride, pt, pl, oh, ow,
                                                           ptr(out_stats), stream()))
        else:
            check(_lib.lib().embnet_dwconv2d_fwd_f32(ptr(x), ptr(w), ptr(y), n, h, wd, c, r, s, stride, pt, pl, oh, ow,
                                                     stream()))
        ctx.geom, ctx.bn_src = geom, bn_src
        ctx.save_for_backward(x, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        lib = _lib.lib()
        n, h, wd, c = x.shape
        r, s = w.shape[0], w.shape[1]
        stride, pt, pl, oh, ow = ctx.geom
        dy = _c(dy)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(x)
            bn_src = getattr(ctx, "bn_src", None)
            rows = lib.embnet_dwconv2d_dgrad_bnsums_rows(n, h, wd, c, r, s, stride) if bn_src is not None else 0
            if rows > 0:
                bn_x, bn_stats, bn_act = bn_src
                partial = (torch.zeros if c // 4 > 256 else torch.empty)((2, c, rows), device=x.device, dtype=torch.float32)
                sp = bn_stats.data_ptr()
                check(lib.embnet_dwconv2d_dgrad_bnsums_f32(ptr(dy), ptr(w), ptr(dx), n, h, wd, c, r, s, stride, pt, pl, oh, ow,
                                                           ptr(bn_x), sp + 8 * c, sp + 12 * c, sp, sp + 4 * c, int(bn_act),
                                                           ptr(partial), rows, stream()))
                while len(BN_SUMS) >= 8:
                    BN_SUMS.pop(next(iter(BN_SUMS)))
                BN_SUMS[dx.data_ptr()] = (partial, rows, bn_x.data_ptr(), dx.detach())
            else:
                check(lib.embnet_dwconv2d_dgrad_f32(ptr(dy), ptr(w), ptr(dx), n, h, wd, c, r, s, stride, pt, pl, oh, ow,
                                                    stream()))
        if ctx.needs_input_grad[1]:
            dw, note = _sink(w)
            ws = workspace(lib.embnet_dwconv2d_wgrad_workspace_bytes(n, c, r, s, oh, ow), x.device)
            check(lib.embnet_dwconv2d_wgrad_f32(ptr(x), ptr(dy), ptr(dw), ptr(ws), ws.numel() * 4, n, h, wd, c, r, s,
                                                stride, pt, pl, oh, ow, stream()))
            dw = _done(dw, note)
        return dx, dw, None, None, None


def conv_normal_(t, gen):
    """efficientnet's CONV_KERNEL_INITIALIZER: VarianceScaling(scale=2, mode='fan_out', normal)."""
    shape = t.shape
    fan_out = shape[-1] * int(math.prod(shape[:-2]))
    return t.normal_(0.0, math.sqrt(2.0 / fan_out), generator=gen)


class DepthwiseConv2D(nn.Module):
    """Keras DepthwiseConv2D(padding='same', use_bias=False); depthwise_kernel [k,k,C,1]."""

    def __init__(self, channels, kernel_size, strides=1, gen=None):
        super().__init__()
        self.k, self.stride = kernel_size, strides
        w = torch.empty(kernel_size, kernel_size, channels, 1)
        # fan_out of a depthwise kernel in Keras' VarianceScaling: k*k*depth_multiplier(=1)... uses shape[-1]*rf
        self.depthwise_kernel = nn.Parameter(conv_normal_(w, gen))

    def forward(self, x, emit_stats=False):
        """emit_stats: a training-mode BatchNormalization reads this output next — the kernel also produces its per-channel
        sums (attached to the result as `_bn_partials`, as Conv2D does), so that layer skips its statistics pass."""
        h, w = x.shape[1], x.shape[2]
        oh, pt = same_pad(h, self.k, self.stride)
        ow, pl = same_pad(w, self.k, self.stride)
        out_stats = None
        if emit_stats and DW_EMIT_STATS[0]:
            n, c = x.shape[0], x.shape[-1]
            rows = _lib.lib().embnet_dwconv2d_fwd_stats_rows(n, c, self.k, self.k, self.stride, oh, ow)
            if rows > 0:               # (a workgroup covers 256 channel quads: wider layers start from zeros)
                out_stats = (torch.zeros if c // 4 > 256 else torch.empty)((2, c, rows), device=x.device, dtype=torch.float32)
        bn_src = getattr(x, "_bn_src", None)
        if not (DW_BN_SUMS[0] and FUSE_BN_SUMS[0] and bn_src is not None and self.stride in (1, 2) and torch.is_grad_enabled()
                and x.requires_grad and x.shape[-1] % 4 == 0):
            bn_src = None
        y = _DepthwiseFn.apply(x, self.depthwise_kernel, (self.stride, pt, pl, oh, ow), out_stats, bn_src)
        if out_stats is not None:
            y._bn_partials = out_stats
        return y


class _ActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, kind):
        x = _c(x)
        y = torch.empty_like(x)
        check(_lib.lib().embnet_activation_fwd(ptr(x), x.numel(), kind, ptr(y), stream()))
        ctx.kind = kind
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        check(_lib.lib().embnet_activation_bwd(ptr(x), ptr(_c(dy)), x.numel(), ctx.kind, ptr(dx), stream()))
        return dx, None


def sigmoid(x):
    return _ActFn.apply(x, 0)


def swish(x):
    return _ActFn.apply(x, 1)


class Swish(nn.Module):
    def forward(self, x):
        return swish(x)


class _ChannelScaleFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, s, lazy=None):
        """lazy = (bn_x, bn_stats, bn_act) of the BatchNormalization that produced x and granted lazy_scale, or None."""
        x, s = _c(x), _c(s)
        n, h, w, c = x.shape
        y = torch.empty_like(x)
        check(_lib.lib().embnet_channel_scale_fwd(ptr(x), ptr(s), n, h * w, c, ptr(y), stream()))
        ctx.lazy = lazy
        ctx.save_for_backward(x, s)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, s = ctx.saved_tensors
        n, h, w, c = x.shape
        dy = _c(dy)
        ds = torch.empty_like(s)
        if ctx.lazy is not None:
            # x's producer (a BatchNormalization that granted lazy_scale) multiplies by s inside its backward passes: only the
            # gate's gradient is computed here and dy travels on UNSCALED, announced in GATE_PENDING (with an alias, so that
            # autograd cannot add into it in place) — that layer raises if it receives anything else
            sums = None
            if SE_BN_SUMS[0]:
                # ... and the same pass over (dy, the BatchNormalization's input) leaves the per-(image, channel) sums from which
                # that layer's dbeta / dgamma follow: it then skips its reduction pass (embnet_se_bn_sums)
                bn_x, bn_stats, bn_act = ctx.lazy
                sums = torch.empty((n, 5, c), device=x.device, dtype=torch.float32)
                sp = bn_stats.data_ptr()
                check(_lib.lib().embnet_se_bn_sums(ptr(dy), ptr(bn_x), n, h * w, c, sp, sp + 4 * c, sp + 8 * c, sp + 12 * c, int(bn_act),
                                                   ptr(sums), stream()))
                ds = sums[:, 0, :]
            else:
                check(_lib.lib().embnet_channel_scale_dgate(ptr(x), ptr(dy), n, h * w, c, ptr(ds), stream()))
            while len(GATE_PENDING) >= 8:
                GATE_PENDING.pop(next(iter(GATE_PENDING)))
            GATE_PENDING[dy.data_ptr()] = (s, dy.detach(), sums)
            return dy, ds, None
        dx = torch.empty_like(x)
        check(_lib.lib().embnet_channel_scale_bwd(ptr(x), ptr(s), ptr(dy), n, h * w, c, ptr(dx), ptr(ds), stream()))
        return dx, ds, None


def channel_scale(x, s, lazy=False):
    """x[n,h,w,c] * s[n,c] (the squeeze-excite multiply).  lazy=True: x is the output of BatchNormalization(emit_gap=True,
    lazy_scale=True) and THIS is its only consumer — the multiply's backward is left to that layer (see _ChannelScaleFn)."""
    src = getattr(x, "_lazy_scale_ok", None) if (lazy and x.shape[-1] % 4 == 0) else None
    return _ChannelScaleFn.apply(x, s.reshape(x.shape[0], x.shape[-1]), src)


class _SampleDropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, rate, seed):
        x = _c(x)
        y = torch.empty_like(x)
        per = x.numel() // x.shape[0]
        tick = GRAPH_TICK
        check(_lib.lib().embnet_sample_dropout(ptr(x), x.numel(), per, rate, seed, tick, ptr(y), stream()))
        ctx.cfg = (rate, seed, per, tick)
        return y

    @staticmethod
    def backward(ctx, dy):
        rate, seed, per, tick = ctx.cfg
        dy = _c(dy)
        dx = torch.empty_like(dy)
        check(_lib.lib().embnet_sample_dropout(ptr(dy), dy.numel(), per, rate, seed, tick, ptr(dx), stream()))
        return dx, None, None


class DropConnect(nn.Module):
    """efficientnet's FixedDropout(noise_shape=(None,1,1,1)): drops whole samples of the residual branch."""

    def __init__(self, rate, seed=0):
        super().__init__()
        self.rate, self.seed, self.enabled, self._step = rate, seed, True, 0

    def forward(self, x):
        if not (self.training and self.enabled and self.rate > 0):
            return x
        self._step += 1
        return _SampleDropoutFn.apply(x, self.rate, (self.seed << 32) + self._step)


class _AbsDiffFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        y = torch.empty_like(a)
        check(_lib.lib().embnet_absdiff_fwd(ptr(a), ptr(b), a.numel(), ptr(y), stream()))
        ctx.save_for_backward(a, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        a, b = ctx.saved_tensors
        da, db = torch.empty_like(a), torch.empty_like(b)
        check(_lib.lib().embnet_absdiff_bwd(ptr(a), ptr(b), ptr(_c(dy)), a.numel(), ptr(da), ptr(db), stream()))
        return da, db


def abs_diff(a, b):
    return _AbsDiffFn.apply(a, b)


class _L2PenaltyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, w, lam):
        w = _c(w)
        lib = _lib.lib()
        out = torch.empty((), device=w.device, dtype=torch.float32)
        ws = workspace(lib.embnet_sumsq_workspace_bytes(), w.device)
        check(lib.embnet_sumsq(ptr(w), w.numel(), lam, ptr(out), ptr(ws), ws.numel() * 4, stream()))
        ctx.lam = lam
        ctx.save_for_backward(w)
        return out

    @staticmethod
    def backward(ctx, dout):
        (w,) = ctx.saved_tensors
        dw = torch.empty_like(w)
        check(_lib.lib().embnet_scale(ptr(w), w.numel(), 2.0 * ctx.lam, ptr(_c(dout)), ptr(dw), stream()))
        return dw, None


def l2_penalty(w, lam):
    """Keras regularizers.l2(lam)(w) = lam * sum(w^2)."""
    return _L2PenaltyFn.apply(w, float(lam))


def regularized_kernels(module):
    """[(kernel parameter, lambda)] of the Conv2D / Dense layers that declare kernel_regularizer=l2(lambda)."""
    return [(m.kernel, float(m.l2)) for m in module.modules()
            if isinstance(m, (Conv2D, Dense)) and getattr(m, "l2", 0.0) and m.kernel.requires_grad]


class _L2MultiFn(torch.autograd.Function):
    """sum_t lambda_t * sum(w_t^2) over all regularised kernels in ONE launch pair (embnet_sumsq_multi).
    with_grad=False: the value only — the caller's optimizer adds 2*lambda*w to the gradients itself
    (KerasOptimizer.set_l2: folded into embnet_optimizer_step), so backward returns nothing."""

    @staticmethod
    def forward(ctx, with_grad, plan, *kernels):
        lib = _lib.lib()
        out = torch.empty((), device=kernels[0].device, dtype=torch.float32)
        ws = workspace(4 * plan["chunks"].shape[0], out.device)
        check(lib.embnet_sumsq_multi(plan["table"].data_ptr(), len(kernels), plan["chunks"].data_ptr(), plan["chunks"].shape[0],
                                     ptr(out), ptr(ws), ws.numel() * 4, stream()))
        ctx.with_grad, ctx.lams = with_grad, plan["lams"]
        if with_grad:
            ctx.save_for_backward(*kernels)
        else:
            ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, dout):
        if not ctx.with_grad:
            return (None, None) + (None,) * len(ctx.lams)
        grads = []
        for w, lam in zip(ctx.saved_tensors, ctx.lams):
            dw = torch.empty_like(w)
            check(_lib.lib().embnet_scale(ptr(w), w.numel(), 2.0 * lam, ptr(_c(dout)), ptr(dw), stream()))
            grads.append(dw)
        return (None, None) + tuple(grads)


def regularization_loss(module, with_grad=True):
    """Sum of the l2 kernel regularisers declared on Conv2D/Dense layers (Keras adds it to the loss).
    with_grad=False: value only (see _L2MultiFn)."""
    import numpy as np
    ks = regularized_kernels(module)
    if not ks:
        return None
    key = tuple((k.data_ptr(), k.numel(), lam) for k, lam in ks)
    plan = getattr(module, "_l2_plan", None)
    if plan is None or plan["key"] != key:               # descriptor table + chunk list, rebuilt only if storage moved
        dev = ks[0][0].device
        ce = _lib.lib().embnet_sumsq_chunk_elems()
        rows = np.zeros((len(ks), 3), dtype=np.int64)
        for i, (k, lam) in enumerate(ks):
            rows[i, 0], rows[i, 1] = k.data_ptr(), k.numel()
            rows[i, 2] = int(np.float32(lam).view(np.uint32))               # {float alpha; int32 pad} in one int64
        chunks = [(i, c) for i, (k, _) in enumerate(ks) for c in range(-(-k.numel() // ce))]
        plan = dict(key=key, table=torch.from_numpy(rows).to(dev), lams=[lam for _, lam in ks],
                    chunks=torch.tensor(chunks, dtype=torch.int32, device=dev))
        module._l2_plan = plan
    return _L2MultiFn.apply(bool(with_grad), plan, *[k for k, _ in ks])
