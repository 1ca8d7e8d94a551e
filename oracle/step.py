"""Oracle (test infrastructure / timed CPU baseline): ONE training step with the reference's
own step structure, on PyTorch-CPU float32.  PARITY UNPINNED for the backbone part (see
oracle/__init__.py); the mining and loss parts are the pinned restatements.

Per step, as /root/reference does it:
  1. for each of the P sampled classes: predict() on its K images — inference-mode forward
     (datagenerators.py:211-215);
  2. all-pairs Euclidean matrix with f64 accumulation (datagenerators.py:219);
  3. Python mining loop over the P*C(K,2) ordered positive pairs (datagenerators.py:225-250);
  4. three-branch training forward/backward on the T mined triplets, squared-L2 hinge, mean over T
     plus the kernel regularisers, optimizer update (models.py:181-185,
     losses_and_accuracies.py:26-42, train.py:160-177).
"""
import numpy as np
import torch

from . import backbones as OB
from . import mining


class ReferenceStep:
    def __init__(self, backbone_name, input_shape, encodings_len, k_classes, k_samples, margin, mode,
                 lr=1e-3, seed=0, params=None, optimizer="adam"):
        self.kw = dict(backbone_name=backbone_name, encodings_len=encodings_len)
        self.p, self.k, self.margin, self.mode = k_classes, k_samples, margin, mode
        self.shape = tuple(input_shape)
        ctx = OB.Ctx(params, training=False, seed=seed)
        with torch.no_grad():                                  # materialise the weights
            OB.base_model(ctx, torch.zeros((2,) + self.shape), **self.kw)
        self.params = ctx.params
        self.trainable = [v for k, v in self.params.items() if "moving_" not in k]
        for v in self.trainable:
            v.requires_grad_(True)
        if optimizer == "radam":
            self.opt = torch.optim.RAdam(self.trainable, lr=lr, eps=1e-7)
        elif optimizer == "adam":
            self.opt = torch.optim.Adam(self.trainable, lr=lr, eps=1e-7)
        else:
            self.opt = torch.optim.SGD(self.trainable, lr=lr)

    def step(self, images, rng=None):
        """images: float32 [P*K, H, W, 3] class-contiguous.  Returns (loss, T)."""
        p, k = self.p, self.k
        x = torch.as_tensor(images, dtype=torch.float32)
        with torch.no_grad():                                   # 1. P predict() calls
            ctx = OB.Ctx(self.params, training=False)
            emb = torch.cat([OB.base_model(ctx, x[c * k:(c + 1) * k], **self.kw) for c in range(p)])
        mined = mining.mine_from_embeddings(emb.numpy(), p, k, self.margin, self.mode, rng)   # 2. + 3.
        t = torch.as_tensor(mined["triplets"], dtype=torch.long)
        ctx = OB.Ctx(self.params, training=True)                # 4. three-branch train step
        y = OB.triplet_model(ctx, x[t[:, 0]], x[t[:, 1]], x[t[:, 2]], **self.kw)
        e = y.shape[1] // 3
        pos = ((y[:, :e] - y[:, e:2 * e]) ** 2).sum(1)
        neg = ((y[:, :e] - y[:, 2 * e:]) ** 2).sum(1)
        loss = torch.clamp(pos - neg + self.margin, min=0).mean()
        total = loss + OB.regularisation(ctx)
        self.opt.zero_grad(set_to_none=True)
        total.backward()
        self.opt.step()
        with torch.no_grad():
            for kname, v in ctx.new_stats.items():
                self.params[kname].copy_(v)
        return float(loss.detach()), int(len(t))
