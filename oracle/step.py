"""Oracle (test infrastructure / timed CPU baseline): ONE training step with the reference's
own step structure, on PyTorch-CPU (float32 for the timed baseline, float64 for the parity curves).
PARITY UNPINNED for the backbone and optimizer parts (see oracle/__init__.py); the mining and loss
parts are the pinned restatements.

Per step, as /root/reference does it:
  1. for each of the P sampled classes: predict() on its K images — inference-mode forward
     (datagenerators.py:211-215);
  2. all-pairs Euclidean matrix with f64 accumulation (datagenerators.py:219);
  3. Python mining loop over the P*C(K,2) ordered positive pairs (datagenerators.py:225-250);
  4. three-branch training forward/backward on the T mined triplets, squared-L2 hinge, mean over T
     plus the kernel regularisers, optimizer update with the Keras rule (models.py:181-185,
     losses_and_accuracies.py:26-42, train.py:160-177, utils.py:143-153 -> oracle/optimizers.py).
"""
import numpy as np
import torch

from . import backbones as OB
from . import mining
from . import optimizers as OO


class ReferenceStep:
    def __init__(self, backbone_name, input_shape, encodings_len, k_classes, k_samples, margin, mode,
                 lr=1e-3, seed=0, params=None, optimizer="adam", dtype=torch.float32):
        self.kw = dict(backbone_name=backbone_name, encodings_len=encodings_len)
        self.p, self.k, self.margin, self.mode = k_classes, k_samples, margin, mode
        self.shape, self.dtype = tuple(input_shape), dtype
        ctx = OB.Ctx(params, training=False, seed=seed)
        if params is None:
            with torch.no_grad():                              # materialise the weights (float32 initialisers)
                OB.base_model(ctx, torch.zeros((2,) + self.shape), **self.kw)
        self.params = {k: v.detach().clone().to(dtype) for k, v in ctx.params.items()}      # own copies
        self.names = [k for k in self.params if "moving_" not in k]
        for k in self.names:
            self.params[k].requires_grad_(True)
        self.opt = OO.get_optimizer(optimizer, lr)

    def embed(self, x):
        """predict(): inference-mode embeddings, one call per class block as the reference does."""
        with torch.no_grad():
            ctx = OB.Ctx(self.params, training=False)
            return torch.cat([OB.base_model(ctx, x[c * self.k:(c + 1) * self.k], **self.kw) for c in range(self.p)])

    def mine(self, images, rng=None):
        """Steps 1-3 on a class-contiguous batch: -> oracle/mining.py's result dict (triplets, loss_values, ...)."""
        x = torch.as_tensor(images, dtype=self.dtype)
        emb = self.embed(x)                                     # 1.
        return mining.mine_from_embeddings(emb.numpy().astype(np.float32), self.p, self.k, self.margin,
                                           self.mode, rng)      # 2. + 3.

    def step(self, images, rng=None, triplets=None):
        """images: [P*K, H, W, 3] class-contiguous.  Returns (triplet loss, T, total loss incl. regularisers).
        `triplets` [T,3] replaces the oracle's own mining for the training part (parity tests pass the device's
        triplets once they have checked them against mine(), so that an fp32-borderline tie in the mining does
        not end the comparison of the curves)."""
        x = torch.as_tensor(images, dtype=self.dtype)
        if triplets is None:
            triplets = self.mine(images, rng)["triplets"]
        t = torch.as_tensor(np.asarray(triplets), dtype=torch.long)
        ctx = OB.Ctx(self.params, training=True)                # 4. three-branch train step
        y = OB.triplet_model(ctx, x[t[:, 0]], x[t[:, 1]], x[t[:, 2]], **self.kw)
        e = y.shape[1] // 3
        pos = ((y[:, :e] - y[:, e:2 * e]) ** 2).sum(1)
        neg = ((y[:, :e] - y[:, 2 * e:]) ** 2).sum(1)
        loss = torch.clamp(pos - neg + self.margin, min=0).mean()
        total = loss + OB.regularisation(ctx)
        ws = [self.params[k] for k in self.names]
        grads = torch.autograd.grad(total, ws, allow_unused=True)
        self.last_grads = {k: g for k, g in zip(self.names, grads) if g is not None}      # kept for the parity tests
        with torch.no_grad():
            arrs = [w.detach().numpy() for w in ws]            # views: the update lands in the tensors
            if self.dtype == torch.float64:
                self.opt.step(arrs, [None if g is None else g.numpy() for g in grads])
            else:                                               # float32 parameters: update in f64, round once
                up = [a.astype(np.float64) for a in arrs]
                self.opt.step(up, [None if g is None else g.numpy() for g in grads])
                for a, u in zip(arrs, up):
                    a[...] = u
            for kname, v in ctx.new_stats.items():
                self.params[kname].copy_(v)
        return float(loss.detach()), int(len(t)), float(total.detach())


class SiameseReferenceStep:
    """The reference's siamese training step (tools/train.py:108-119, models.py:203-230, 'l2' distance head):
    two forwards of the shared base model (two BatchNorm batches), d = sqrt(max(sum (e1-e2)^2, eps)),
    contrastive_loss (losses_and_accuracies.py:4-11), backward, Keras-rule optimizer.  PARITY UNPINNED like the rest
    of this file (Keras layers)."""

    def __init__(self, backbone_name, input_shape, encodings_len, lr=1e-3, seed=0, params=None, optimizer="adam",
                 dtype=torch.float32):
        self.kw = dict(backbone_name=backbone_name, encodings_len=encodings_len)
        self.shape, self.dtype = tuple(input_shape), dtype
        ctx = OB.Ctx(params, training=False, seed=seed)
        if params is None:
            with torch.no_grad():
                OB.base_model(ctx, torch.zeros((2,) + self.shape), **self.kw)
        self.params = {k: v.detach().clone().to(dtype) for k, v in ctx.params.items()}      # own copies
        self.names = [k for k in self.params if "moving_" not in k]
        for k in self.names:
            self.params[k].requires_grad_(True)
        self.opt = OO.get_optimizer(optimizer, lr)

    def step(self, x1, x2, y):
        """x1, x2: [B,H,W,3]; y: [B] or [B,1], 1 = same class.  Returns the contrastive loss."""
        x1, x2 = torch.as_tensor(x1, dtype=self.dtype), torch.as_tensor(x2, dtype=self.dtype)
        yt = torch.as_tensor(np.asarray(y), dtype=self.dtype).reshape(-1, 1)
        ctx = OB.Ctx(self.params, training=True)
        d = OB.siamese_l2_distance(OB.base_model(ctx, x1, **self.kw), OB.base_model(ctx, x2, **self.kw))
        loss = (yt * d ** 2 + (1 - yt) * torch.clamp(1 - d, min=0) ** 2).mean()
        ws = [self.params[k] for k in self.names]
        grads = torch.autograd.grad(loss + OB.regularisation(ctx), ws, allow_unused=True)
        with torch.no_grad():
            up = [w.detach().numpy().astype(np.float64) for w in ws]
            self.opt.step(up, [None if g is None else g.numpy() for g in grads])
            for w, u in zip(ws, up):
                w.copy_(torch.as_tensor(u).to(self.dtype))
            for kname, v in ctx.new_stats.items():
                self.params[kname].copy_(v)
        return float(loss.detach())
