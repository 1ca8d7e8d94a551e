"""Data-parallel plumbing: one process per GPU, RCCL (torch.distributed 'nccl' on ROCm) over xGMI.

The reference has no gradient exchange at all (its multi-GPU switch, tools/train.py:121-140, only
replicates the mining predict()), so this is build-defined: every rank runs the fused
step on its own whole classes (mining stays local: no cross-GPU negatives), and the ONLY collective
is the all-reduce of the flat fp32 gradient, issued bucket by bucket from autograd hooks so it
overlaps the rest of backward.  Buckets are ~16 MB slices of one contiguous buffer plus a small
final one (xGMI is point-to-point, 7 links x ~153 GB/s: a ring step is bound by one link, so few
large messages beat many small ones, and the last, exposed message should be latency-sized);
ResNet18's 45 MB gradient goes out as 16 + 16 + 12 + 0.7 MB.  Every slot starts on a 16-byte
boundary (the kernels that write gradients in place use 16-byte stores).
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None, timeout_s=None):
    """Initialise from torchrun's env (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*).  Returns (rank, world, local_rank).
    timeout_s (or EMBNET_DIST_TIMEOUT_S): collective timeout — the default (10 min for RCCL) is the watchdog that aborts a
    rank waiting in a collective while another rank does something long on its own."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # EMBNET_DIST_BACKEND=gloo: debugging aid — runs the multi-rank path (broadcasts, bucketed gradient all-reduce) on
        # GPU tensors without RCCL, e.g. two ranks sharing the one GPU of a test box
        backend = backend or os.environ.get("EMBNET_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        timeout_s = timeout_s or os.environ.get("EMBNET_DIST_TIMEOUT_S")
        kw = {}
        if timeout_s:
            import datetime
            kw["timeout"] = datetime.timedelta(seconds=float(timeout_s))
        try:
            dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
        except Exception as e:          # noqa: BLE001 — whatever the store raises when the port is taken
            # a self-spawned world (embeddingnet_amd/launch.py) picked its port by bind-and-close: if somebody took it before
            # rank 0 bound it, say so with the exit code the parent restarts the world for
            if os.environ.get("EMBNET_SPAWNED") == "1" and any(t in str(e).lower() for t in ("address already in use", "eaddrinuse")):
                import sys
                from .launch import EXIT_PORT_IN_USE
                print(f"[rank {rank}] rendezvous port {os.environ.get('MASTER_PORT')} is in use: {e}", file=sys.stderr, flush=True)
                sys.exit(EXIT_PORT_IN_USE)
            raise
    return rank, world, local


def shard_classes(k_classes_global, world, rank):
    """Contiguous block of whole classes per rank, so every anchor keeps its K-1 local positives.
    Returns (first_class, n_local_classes)."""
    if k_classes_global % world:
        raise ValueError(f"k_classes={k_classes_global} is not divisible by world size {world}")
    per = k_classes_global // world
    if per < 2:
        raise ValueError("each rank needs at least 2 classes to have negatives")
    return rank * per, per


def broadcast_model(module, src=0, process_group=None):
    """Make every rank start from rank `src`'s parameters AND buffers (BatchNorm moving statistics): one flat
    broadcast.  Call after construction, after --resume_from, and after any rank-local pre-training."""
    if not (dist.is_initialized() and dist.get_world_size(process_group) > 1):
        return
    tensors = [t.data for t in list(module.parameters()) + list(module.buffers()) if t.is_floating_point()]
    if not tensors:
        return
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.broadcast(flat, src=src, group=process_group)
    off = 0
    for t in tensors:
        t.copy_(flat[off:off + t.numel()].view_as(t))
        off += t.numel()
    from . import layers as L
    L.WEIGHT_EPOCH[0] += 1               # (written through .data: the parameters' version counters did not move)


def average_buffers(module, process_group=None):
    """Mean of the floating-point buffers (BatchNorm moving statistics) over the ranks — each rank normalises its own
    local batches (no SyncBN, as in the reference), so a checkpoint should hold the average, not rank 0's copy."""
    if not (dist.is_initialized() and dist.get_world_size(process_group) > 1):
        return
    bufs = [b.data for b in module.buffers() if b.is_floating_point()]
    if not bufs:
        return
    flat = torch.cat([b.reshape(-1) for b in bufs])
    dist.all_reduce(flat, group=process_group)
    flat /= dist.get_world_size(process_group)
    off = 0
    for b in bufs:
        b.copy_(flat[off:off + b.numel()].view_as(b))
        off += b.numel()


def all_reduce_mean(value, device=None, process_group=None):
    """Mean over ranks of a Python float / 0-d tensor (the scalar loss all-reduce SURVEY §8e asks for: logging, and
    the plateau / early-stopping decisions, which must be identical on every rank).  Returns a float."""
    if not (dist.is_initialized() and dist.get_world_size(process_group) > 1):
        return float(value)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(process_group) == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, group=process_group)
    return float(t.item()) / dist.get_world_size(process_group)


class GradReducer:
    """Flat gradient buffer + bucketed asynchronous all-reduce (mean over ranks).

    Contract: the reducer owns the parameters' `.grad` (views of one flat buffer).  Start a step with `zero()`,
    never `optimizer.zero_grad(set_to_none=True)`; a gradient found outside the buffer (someone dropped the view)
    is copied back in and re-bound by the hook, so the all-reduce never silently misses it.  A parameter that took
    no part in a step contributes zeros (its bucket is still reduced, so the ranks stay in step).
    Bucket order: parameters are laid out in the order the FIRST backward produced their gradients (recorded by the
    hooks, rank 0's order broadcast so every rank agrees), so from step 2 on each bucket closes as early as possible.
    A parameter counts ONCE per step however often its hook fires; for gradient accumulation over several backwards use
    hold(True) ... hold(False); reduce_all() (a bucket reduced after the first backward would miss the later ones).
    `direct(True)`: the weight-gradient / BatchNorm / bias kernels write straight into the flat-buffer views
    (layers.GRAD_SINKS) and report to the reducer themselves, so autograd launches no AccumulateGrad `add_` kernel per
    parameter — only valid while each parameter receives ONE gradient per step (the fused TripletTrainer step).
    `hold(True)`: count, but launch no collective (a step being captured into a HIP graph); `reduce_all()` then reduces
    every bucket in order.
    Bucket sizes: ~bucket_bytes each (xGMI is point-to-point: few, large messages), except that the gradients produced LAST
    in backward (the first layers' — tail_bytes of them) get a bucket of their own: every earlier bucket's all-reduce overlaps
    the rest of backward, but the final bucket's is fully exposed in front of the optimizer, so it should be a small,
    latency-sized message rather than whatever remainder the cutting left (ResNet18: 45 MB = 16 + 16 + 12 + 0.7 MB instead
    of 32 + 13).  On RCCL the mean is taken by the collective itself (ReduceOp.AVG); other backends sum and scale once.
    """

    def __init__(self, params, bucket_bytes=16 << 20, process_group=None, always_reduce=False, tail_bytes=1 << 20):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("GradReducer: no trainable parameters")
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self._reduce = self.world > 1 or (always_reduce and dist.is_initialized())   # always_reduce: test hook
        dev, dtype = self.params[0].device, self.params[0].dtype
        self._bucket_bytes, self._tail_bytes = bucket_bytes, tail_bytes
        # every slot starts on a 16-byte boundary: the weight-gradient / BatchNorm kernels that write the views in place
        # (direct) store and load them as float4; the zero padding rides along in the all-reduce
        total = sum(self._pad(p.numel()) for p in self.params)
        self.flat = torch.zeros(total, device=dev, dtype=dtype)
        self._works, self._hold, self._direct = [], False, False
        self._layout(list(reversed(self.params)))           # first guess: reverse definition order
        self._seen, self._ordered = [], False               # hook order of the first backward
        self._fired = set()                                 # parameters already counted in this step
        # default: SUM + one in-place scale of the flat buffer (works on every backend); on RCCL the mean is taken inside
        # the collective (ReduceOp.AVG is NCCL/RCCL-only), checked once against sum-and-scale on the first real exchange
        # EMBNET_DP_AVG=1 opts into the in-collective mean; the default stays SUM + scale (one ~15 us pass over the flat buffer)
        # until a multi-GPU run has validated AVG on this RCCL (N > 1 has never run on hardware).  Constructing a
        # reducer with EMBNET_DP_AVG=1 at world size > 1 issues three small collectives (the self-check below): every rank must
        # construct its reducer at the same point.
        self._avg = dist.ReduceOp.SUM
        self.mean_mode = "sum + scale"
        if self._reduce and dist.get_backend(process_group) == "nccl" and os.environ.get("EMBNET_DP_AVG", "0") == "1":
            if self.world == 1 or self._avg_matches_sum_and_scale(dev):
                self._avg = dist.ReduceOp.AVG               # RCCL averages inside the collective: no scaling pass over the buffer
                self.mean_mode = "ReduceOp.AVG" + (" (self-check passed)" if self.world > 1 else "")
            else:
                import warnings
                warnings.warn("GradReducer: ReduceOp.AVG unavailable or disagreeing with sum-and-scale on this RCCL; using SUM + scale")
                self.mean_mode = "sum + scale (ReduceOp.AVG failed its self-check)"
        self._handles = [p.register_post_accumulate_grad_hook(self._hook) for p in self.params]

    def _avg_matches_sum_and_scale(self, dev):
        """One-time self-check of the in-collective mean (a collective: every rank constructs its reducer at the same
        point): a rank-dependent vector averaged by ReduceOp.AVG against the same vector summed and scaled.  Both results
        are identical on all ranks, so all ranks take the same decision."""
        rank = dist.get_rank(self.group)
        t = torch.arange(1024, device=dev, dtype=torch.float32) * (0.37 + rank) - 11.0 * rank
        a, b = t.clone(), t.clone()
        ok = True
        try:
            dist.all_reduce(a, op=dist.ReduceOp.AVG, group=self.group)
        except Exception:                   # a build without AVG raises in the argument check, before any communication
            ok = False
        dist.all_reduce(b, op=dist.ReduceOp.SUM, group=self.group)
        ok = ok and bool(torch.allclose(a, b / self.world, rtol=1e-6, atol=1e-6))
        # the verdict is itself agreed on (MIN over ranks): no rank may average in the collective while another sums and scales
        flag = torch.tensor([1.0 if ok else 0.0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(flag.item() > 0.5)

    @staticmethod
    def _pad(n, quantum=4):
        return (n + quantum - 1) // quantum * quantum

    def close(self):
        """Detach from the parameters (hooks, gradient sinks, .grad views): another reducer may take them over."""
        self.direct(False)
        for h in self._handles:
            h.remove()
        self._handles = []
        for p in self.params:
            p.grad = None

    def _layout(self, order, keep=False):
        """Assign flat-buffer slots and buckets in `order`; keep=True carries the current gradients over."""
        old = {p: p.grad.clone() for p in order} if keep else None
        self.buckets, self._bucket_of, self._slot = [], {}, {}
        off, start, pending = 0, 0, 0
        per_bucket = max(self._bucket_bytes // self.flat.element_size(), 1)
        # where the final (exposed) bucket starts: the trailing parameters that together stay within tail_bytes
        tail_from, acc_tail = len(order), 0
        for i in range(len(order) - 1, 0, -1):
            acc_tail += order[i].numel() * self.flat.element_size()
            if acc_tail > self._tail_bytes:
                break
            tail_from = i
        for i, p in enumerate(order):
            if i == tail_from and pending:                  # close the running bucket in front of the tail
                self.buckets.append([start, off, pending])
                start, pending = off, 0
            n = p.numel()
            self._slot[p] = (off, n)
            p.grad = self.flat[off:off + n].view_as(p)
            if keep:
                p.grad.copy_(old[p])
            self._bucket_of[p] = len(self.buckets)
            off += self._pad(n)
            pending += 1
            if off - start >= per_bucket:
                self.buckets.append([start, off, pending])
                start, pending = off, 0
        if pending:
            self.buckets.append([start, off, pending])
        self.order = list(order)
        self._left = [b[2] for b in self.buckets]
        if self._direct:
            self.direct(True)                               # the views moved

    def direct(self, on):
        from . import layers as L
        self._direct = bool(on)
        for p in self.params:
            L.GRAD_SINKS.pop(p.data_ptr(), None)
            if on:
                L.GRAD_SINKS[p.data_ptr()] = (p.grad, (lambda q=p: self._count(q)))

    def hold(self, on):
        self._hold = bool(on)

    def _hook(self, p):
        off, n = self._slot[p]
        if p.grad.data_ptr() != self.flat.data_ptr() + off * self.flat.element_size():
            view = self.flat[off:off + n].view_as(p)        # the gradient was re-allocated outside the buffer
            view.copy_(p.grad)
            p.grad = view
        self._count(p)

    def _count(self, p):
        if p in self._fired:                                # a second gradient for the same parameter in this step
            return
        self._fired.add(p)
        if not self._ordered:
            self._seen.append(p)
        b = self._bucket_of[p]
        self._left[b] -= 1
        if self._left[b] == 0 and self._reduce and not self._hold:
            from . import layers as L
            L.flush_slab_reduces()                          # queued weight-gradient sums must land before the bucket leaves
            s, e, _ = self.buckets[b]
            self._works.append(dist.all_reduce(self.flat[s:e], op=self._avg, group=self.group, async_op=True))

    def zero(self):
        self.flat.zero_()
        self._left = [b[2] for b in self.buckets]
        self._fired = set()

    def zero_counts(self):
        """Re-arm the per-step bookkeeping without touching the buffer (a replayed graph zeroes it itself)."""
        self._left = [b[2] for b in self.buckets]
        self._fired = set()

    def reduce_all(self):
        """All buckets, in order, then wait and average: the gradient exchange of a step whose backward ran with the
        collectives held back (a replayed HIP graph)."""
        if self._reduce:
            for s, e, _ in self.buckets:
                self._works.append(dist.all_reduce(self.flat[s:e], op=self._avg, group=self.group, async_op=True))
        for w in self._works:
            w.wait()
        self._works = []
        if self.world > 1 and self._avg == dist.ReduceOp.SUM:
            self.flat.div_(self.world)

    def finish(self):
        """Wait for the outstanding bucket reductions (call after backward, before optimizer.step)."""
        if self._hold:
            return
        if self._reduce:
            # parameters whose hook never fired (unused in this step) keep their bucket open; reduce those too
            for b, left in enumerate(self._left):
                if left:
                    s, e, _ = self.buckets[b]
                    self._works.append(dist.all_reduce(self.flat[s:e], op=self._avg, group=self.group,
                                                       async_op=True))
        for w in self._works:
            w.wait()
        self._works = []
        if self.world > 1 and self._avg == dist.ReduceOp.SUM:
            self.flat.div_(self.world)
        if not self._ordered:
            self._adopt_first_backward_order()

    def _adopt_first_backward_order(self):
        """After the first step: re-lay the buffer in the order the hooks fired (rank 0's, so all ranks agree)."""
        self._ordered = True
        index = {p: i for i, p in enumerate(self.params)}
        seen = [index[p] for p in self._seen]
        seen += [i for i in range(len(self.params)) if i not in set(seen)]      # parameters that never fired: last
        self._seen = []
        order = torch.tensor(seen, dtype=torch.int64, device=self.flat.device if self._reduce and
                             dist.get_backend(self.group) == "nccl" else "cpu")
        if self._reduce:
            dist.broadcast(order, src=0, group=self.group)
        new = [self.params[i] for i in order.tolist()]
        if any(a is not b for a, b in zip(new, self.order)):
            self._layout(new, keep=True)
