"""The optimizers the reference builds in embedding_net/utils.py:143-153 — `optimizers.Adam(lr)`,
`optimizers.RMSprop(lr)`, `keras_radam.RAdam(lr)`, else `optimizers.SGD(lr)` — with the Keras update rules and
defaults (Adam/RAdam b1 .9, b2 .999, eps 1e-7; RMSprop rho .9, eps 1e-7, no momentum; plain SGD).

torch.optim's Adam/RAdam fold epsilon into the bias-corrected denominator differently (eps/sqrt(1-b2^t) instead of
eps), so they are not the reference's rules; these are.  Each `step()` is ONE launch of `embnet_optimizer_step`
(csrc/optimizer.hip) over every parameter tensor: a device table of {w, g, slot1, slot2, n} descriptors plus a
static chunk list.  The table is re-uploaded only when a gradient pointer changed since the last step (with the
caching allocator they normally do not).  A parameter without a gradient is skipped, as Keras does.
The classes are torch.optim.Optimizer subclasses, so param_groups[...]['lr'] scheduling and state_dict() work
as usual.  GPU tensors only (the library has no CPU path).
"""
import math

import numpy as np
import torch

from . import _lib
from ._lib import check
from . import layers as L

RULE = {"sgd": 0, "rms_prop": 1, "adam": 2, "radam": 3}


class KerasOptimizer(torch.optim.Optimizer):
    """rule: 'sgd' | 'rms_prop' | 'adam' | 'radam' (the names utils.get_optimizer dispatches on)."""

    def __init__(self, params, rule, lr, beta_1=0.9, beta_2=0.999, rho=0.9, epsilon=1e-7):
        if rule not in RULE:
            raise KeyError(rule)
        super().__init__(params, dict(lr=float(lr)))
        self.rule, self.b1, self.b2, self.rho, self.eps = rule, beta_1, beta_2, rho, epsilon
        self.iterations = 0
        self._tensors = [p for g in self.param_groups for p in g["params"] if p.requires_grad]
        if not self._tensors:
            raise ValueError("optimizer got no trainable parameters")
        if len(self.param_groups) != 1:
            raise ValueError("one parameter group (the reference sets a single learning rate)")
        self._ptrs, self._table, self._chunks, self._host, self._copied = None, None, None, None, None
        self._l2 = {}                # parameter -> lambda of its kernel_regularizer (set_l2): 2*lambda*w joins the gradient in the kernel
        self.coef_dev = None         # device float[6] the kernel reads its scalars from (set by a graph-capturing trainer)

    # -- state ----------------------------------------------------------------------------
    def _slots(self, p):
        st = self.state[p]
        n = {"sgd": 0, "rms_prop": 1, "adam": 2, "radam": 2}[self.rule]
        for i in range(n):
            if f"slot{i + 1}" not in st:
                st[f"slot{i + 1}"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
        return st.get("slot1"), st.get("slot2")

    def set_l2(self, kernels):
        """[(parameter, lambda)]: fold the gradient of kernel_regularizer=l2(lambda) — 2*lambda*w — into the update launch
        (the caller then adds only the regularisers' VALUE to the loss: layers.regularization_loss(..., with_grad=False))."""
        self._l2 = {id(p): float(lam) for p, lam in kernels}
        self._ptrs = None

    def state_dict(self):
        d = super().state_dict()
        d["iterations"] = self.iterations
        return d

    def load_state_dict(self, d):
        d = dict(d)
        self.iterations = int(d.pop("iterations", 0))
        super().load_state_dict(d)
        self._ptrs = None

    # -- one launch -----------------------------------------------------------------------
    def _coefficients(self, lr, t):
        """-> (kernel rule, b1, b2, c1, c2), scalars in double (oracle/optimizers.py states the rules)."""
        if self.rule == "sgd":
            return 0, 0.0, 0.0, 0.0, 0.0
        if self.rule == "rms_prop":
            return 1, self.rho, 0.0, 0.0, 0.0
        b1, b2 = self.b1, self.b2
        if self.rule == "adam":
            return 2, b1, b2, lr * math.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t), 0.0
        sma_inf = 2.0 / (1.0 - b2) - 1.0
        sma_t = sma_inf - 2.0 * t * b2 ** t / (1.0 - b2 ** t)
        if sma_t >= 5.0:
            r_t = math.sqrt((sma_t - 4.0) / (sma_inf - 4.0) * (sma_t - 2.0) / (sma_inf - 2.0) * sma_inf / sma_t)
            return 3, b1, b2, lr * r_t / (1.0 - b1 ** t), 1.0 / (1.0 - b2 ** t)
        return 4, b1, b2, lr / (1.0 - b1 ** t), 0.0

    def _build_table(self):
        lib = _lib.lib()
        dev = self._tensors[0].device
        rows, ptrs = [], []
        for p in self._tensors:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                raise _lib.EmbnetError("KerasOptimizer: parameters must be contiguous fp32 GPU tensors")
            g = p.grad
            if g is not None and (g.dtype != torch.float32 or not g.is_contiguous() or g.shape != p.shape):
                raise _lib.EmbnetError("KerasOptimizer: gradients must be dense contiguous fp32")
            s1, s2 = self._slots(p)
            gp = g.data_ptr() if g is not None else 0
            l2x2 = int(np.float32(2.0 * self._l2.get(id(p), 0.0)).view(np.uint32))      # {float l2x2; int32 pad} as one int64
            rows.append((p.data_ptr(), gp, s1.data_ptr() if s1 is not None else 0,
                         s2.data_ptr() if s2 is not None else 0, p.numel(), l2x2))
            ptrs.append(gp)
        if self._chunks is None:
            ce = lib.embnet_optimizer_chunk_elems()
            ck = [(i, c) for i, p in enumerate(self._tensors) for c in range(-(-p.numel() // ce))]
            self._chunks = torch.tensor(ck, dtype=torch.int32, device=dev)
        if torch.cuda.is_current_stream_capturing():
            # a captured step keeps a table (and its pinned source) of its own: replays re-run the upload node, and eager
            # steps in between must not rewrite what it copies from; no host wait is legal inside a capture either
            host = getattr(self, "_graph_host", None)     # pinned memory cannot be allocated while capturing: prepare_capture()
            if host is None or host.shape[0] != len(rows):
                raise _lib.EmbnetError("KerasOptimizer: call prepare_capture() before capturing a step")
            host.copy_(torch.from_numpy(np.asarray(rows, dtype=np.uint64).view(np.int64)))
            self._table = torch.empty((len(rows), 6), dtype=torch.int64, device=dev)
            self._table.copy_(host, non_blocking=True)
            self._ptrs = ptrs
            return
        # Autograd hands out fresh gradient buffers every step, so this runs every step: a ring of four staging / device
        # tables, so that the host never waits for the upload of the step before (one table + an event wait kept the host
        # at most one step ahead of the GPU: 5 ms of host time per ResNet18 step went into that wait)
        if not hasattr(self, "_ring"):
            self._ring = [[torch.empty((len(rows), 6), dtype=torch.int64).pin_memory(),
                           torch.empty((len(rows), 6), dtype=torch.int64, device=dev), None] for _ in range(4)]
            self._ring_i = 0
        self._ring_i = (self._ring_i + 1) % len(self._ring)
        slot = self._ring[self._ring_i]
        if slot[2] is not None:
            slot[2].synchronize()                         # the upload issued four rebuilds ago
        slot[0].copy_(torch.from_numpy(np.asarray(rows, dtype=np.uint64).view(np.int64)))
        slot[1].copy_(slot[0], non_blocking=True)
        slot[2] = torch.cuda.Event()
        slot[2].record()
        self._table = slot[1]
        self._ptrs = ptrs

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        cur = [p.grad.data_ptr() if p.grad is not None else 0 for p in self._tensors]
        if cur != self._ptrs:
            self._build_table()
        self.iterations += 1
        L.WEIGHT_EPOCH[0] += 1                             # cached bf16 planes of conv kernels are stale from here on
        lr = float(self.param_groups[0]["lr"])
        rule, b1, b2, c1, c2 = self._coefficients(lr, self.iterations)
        check(_lib.lib().embnet_optimizer_step(rule, self._table.data_ptr(), len(self._tensors), self._chunks.data_ptr(),
                                               self._chunks.shape[0], lr, b1, b2, self.eps, c1, c2,
                                               self.coef_dev.data_ptr() if self.coef_dev is not None else None,
                                               _lib.stream()))
        # planes and ranges of the conv kernels among the parameters, current again in one launch each (layers.refresh_tensors)
        L.refresh_tensors(self._tensors, self)
        return loss

    def prepare_capture(self):
        """Allocate what a captured step() needs that cannot be allocated during stream capture."""
        self._graph_host = torch.empty((len(self._tensors), 6), dtype=torch.int64).pin_memory()
        self._ptrs = None                                  # the captured step builds (and keeps) its own table

    def scalars(self, t):
        """-> (kernel rule, [lr, b1, b2, eps, c1, c2]) of step t (1-based): what step() passes, for a caller that keeps
        them in device memory (`coef_dev`) across graph replays."""
        lr = float(self.param_groups[0]["lr"])
        rule, b1, b2, c1, c2 = self._coefficients(lr, t)
        return rule, [lr, b1, b2, self.eps, c1, c2]


def save_optimizer_state(path, opt, named_params, extra=None):
    """Optimizer slots keyed by the Keras weight names (backbones.keras_weights order), `iterations`, and `extra` scalars
    (e.g. the epoch) -> an .npz next to a weights checkpoint, so that --resume_from continues Adam / RAdam / RMSprop where
    they stopped (the reference's ModelCheckpoint saves the optimizer with the model)."""
    data = {"__iterations__": np.asarray(opt.iterations), "__rule__": np.asarray(opt.rule)}
    for k, v in (extra or {}).items():
        data[f"__extra__{k}"] = np.asarray(v)
    index = {id(p): name for name, p in named_params.items()}
    for p, st in opt.state.items():
        name = index.get(id(p))
        if name is None:
            continue
        for slot in ("slot1", "slot2"):
            if slot in st:
                data[f"{name}::{slot}"] = st[slot].detach().cpu().numpy()
    np.savez(path, **data)


def load_optimizer_state(path, opt, named_params):
    """Inverse of save_optimizer_state; returns the `extra` dict.  Slots of parameters the file does not hold stay zero."""
    d = np.load(path, allow_pickle=False)
    if str(d["__rule__"]) != opt.rule:
        raise _lib.EmbnetError(f"optimizer state is for '{d['__rule__']}', the config builds '{opt.rule}'")
    opt.iterations = int(d["__iterations__"])
    mine = {id(q) for g in opt.param_groups for q in g["params"]}
    for name, p in named_params.items():
        if id(p) not in mine:                   # frozen / foreign parameters: no slots, no keys in opt.state
            continue
        s1, s2 = opt._slots(p)
        for slot, t in (("slot1", s1), ("slot2", s2)):
            if t is not None and f"{name}::{slot}" in d:
                t.copy_(torch.as_tensor(d[f"{name}::{slot}"]).to(t.device).view_as(t))
    opt._ptrs = None
    return {k[len("__extra__"):]: d[k] for k in d.files if k.startswith("__extra__")}


def SGD(params, lr):
    return KerasOptimizer(params, "sgd", lr)


def Adam(params, lr):
    return KerasOptimizer(params, "adam", lr)


def RMSprop(params, lr):
    return KerasOptimizer(params, "rms_prop", lr)


def RAdam(params, lr):
    return KerasOptimizer(params, "radam", lr)
