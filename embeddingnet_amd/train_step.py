"""The fused metric-learning training step (BASELINE.json north_star).

One forward of the class-contiguous batch [P*K, H, W, 3] -> embeddings [N,E] ->
N x N distance matrix -> mine-and-select -> hinge on the mined triplets ->
backward -> (all-reduce) -> optimizer.  Everything between the image batch and
the optimizer is HIP kernels from libembnet_hip.so; torch provides the tape and
the optimizer.  No host synchronisation inside a step (the triplet count stays
on the device).

What this replaces in the reference, per step:
  datagenerators.py:211-215  P separate predict() calls           -> the one training forward
  datagenerators.py:219      sklearn pairwise_distances           -> ops.pairwise_distances
  datagenerators.py:225-250  Python mining loop                   -> ops.mine_triplets / ops.batch_hard
  models.py:181-185          3 forwards of the mined images       -> rows gathered from the same embeddings
  losses_and_accuracies.py:26-42 + Keras mean + regularisers      -> ops.triplet_gather_loss + layers.regularization_loss
Documented semantic difference (SURVEY §7 hard part f): the reference mines with inference-mode
embeddings and normalises the a/p/n branches as three separate BN batches; the fused step mines on
the training-mode embeddings of the one batch.
"""
import os

import torch

from . import layers as L
from . import ops


class TripletTrainer:
    def __init__(self, base_model, optimizer, k_classes, k_samples, margin=0.5,
                 negatives_selection_mode="semihard", seed=0, reducer=None):
        self.model, self.opt = base_model, optimizer
        self.p, self.k, self.margin, self.mode = int(k_classes), int(k_samples), float(margin), negatives_selection_mode
        self.seed, self.step_no, self.reducer = int(seed), 0, reducer
        # one launch for distance matrix + mining + hinge + mean when the batch fits the fused kernel (N <= 512)
        self.fused_loss = os.environ.get("EMBNET_FUSED_LOSS", "1") == "1"
        if self.mode not in tuple(ops.MINING_MODES) + ("batch_hard",):
            raise KeyError(self.mode)

    def mine(self, emb):
        with torch.no_grad():
            dist = ops.pairwise_distances(emb)
            if self.mode == "batch_hard":
                trip, count = ops.batch_hard(dist, self.p, self.k)
            else:
                trip, count, _ = ops.mine_triplets(dist, self.p, self.k, self.margin, self.mode,
                                                   seed=(self.seed << 20) + self.step_no)
            self.last_triplets = (trip, count)      # device tensors; only the first count[0] rows are live
            return trip, count

    def loss(self, images):
        """-> (total loss incl. regularisers, triplet loss, live triplet count tensor)."""
        if images.shape[0] != self.p * self.k:
            raise ValueError(f"batch of {images.shape[0]} images != k_classes*k_samples = {self.p * self.k}")
        emb = self.model(images)
        if self.fused_loss and ops.fused_loss_supported(self.p, self.k, emb.shape[1]):
            mean, _, trip, count = ops.fused_triplet_loss(emb, self.p, self.k, self.margin, self.mode,
                                                          seed=(self.seed << 20) + self.step_no)
            self.last_triplets = (trip, count)
        else:
            trip, count = self.mine(emb)
            mean, _ = ops.triplet_gather_loss(emb, trip, count, self.margin)
        reg = L.regularization_loss(self.model)
        return (mean if reg is None else mean + reg), mean, count

    def step(self, images):
        self.model.train()
        self.step_no += 1
        if self.reducer is not None:
            self.reducer.zero()                 # one memset of the flat gradient buffer
        else:
            self.opt.zero_grad(set_to_none=True)
        total, mean, count = self.loss(images)
        self.last_total = total.detach()        # triplet mean + kernel regularisers (what Keras reports as `loss`)
        total.backward()
        if self.reducer is not None:
            self.reducer.finish()
        self.opt.step()
        return mean.detach()
