"""Data-parallel plumbing: one process per GPU, RCCL (torch.distributed 'nccl' on ROCm) over xGMI.

The reference has no gradient exchange at all (its multi-GPU switch, tools/train.py:121-140, only
replicates the mining predict()), so this is build-defined: every rank runs the fused
step on its own whole classes (mining stays local: no cross-GPU negatives), and the ONLY collective
is the all-reduce of the flat fp32 gradient, issued bucket by bucket from autograd hooks so it
overlaps the rest of backward.  Buckets are ~32 MB slices of one contiguous buffer: xGMI is
point-to-point (7 links x ~153 GB/s), a ring step is bound by one link, so few large messages beat
many small ones; ResNet18's 45 MB gradient goes out as two.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise from torchrun's env (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
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


class GradReducer:
    """Flat gradient buffer + bucketed asynchronous all-reduce (mean over ranks)."""

    def __init__(self, params, bucket_bytes=32 << 20, process_group=None, always_reduce=False):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("GradReducer: no trainable parameters")
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self._reduce = self.world > 1 or (always_reduce and dist.is_initialized())   # always_reduce: test hook
        dev, dtype = self.params[0].device, self.params[0].dtype
        order = list(reversed(self.params))                 # roughly the order backward produces them
        total = sum(p.numel() for p in order)
        self.flat = torch.zeros(total, device=dev, dtype=dtype)
        self.buckets, self._bucket_of = [], {}
        off, start, pending = 0, 0, 0
        per_bucket = max(bucket_bytes // self.flat.element_size(), 1)
        for p in order:
            n = p.numel()
            p.grad = self.flat[off:off + n].view_as(p)
            self._bucket_of[p] = len(self.buckets)
            off += n
            pending += 1
            if off - start >= per_bucket:
                self.buckets.append([start, off, pending])
                start, pending = off, 0
        if pending:
            self.buckets.append([start, off, pending])
        self._left = [b[2] for b in self.buckets]
        self._works = []
        # SUM + one in-place scale of the flat buffer (works on every backend; ReduceOp.AVG is
        # NCCL-only and could not be exercised on the single-GPU development box)
        self._avg = dist.ReduceOp.SUM
        for p in self.params:
            p.register_post_accumulate_grad_hook(self._hook)

    def _hook(self, p):
        b = self._bucket_of[p]
        self._left[b] -= 1
        if self._left[b] == 0 and self._reduce:
            s, e, _ = self.buckets[b]
            self._works.append(dist.all_reduce(self.flat[s:e], op=self._avg, group=self.group, async_op=True))

    def zero(self):
        self.flat.zero_()
        self._left = [b[2] for b in self.buckets]

    def finish(self):
        """Wait for the outstanding bucket reductions (call after backward, before optimizer.step)."""
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
