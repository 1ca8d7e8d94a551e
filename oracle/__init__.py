"""oracle/ — CPU restatement of the reference's metric-learning hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke()
and bench.py's `cpu_baseline` leg may import it, and only as the checker /
the timed CPU baseline.  `embeddingnet_amd` never imports it; the product path
raises when the HIP library is missing instead of falling back to this code.

Pinning status (DESIGN.md §Oracle has the full table):
  * losses.py, pairwise.py, mining.py — PINNED against tests/golden/*.npz,
    which tests/golden/gen_golden.py produced by running the reference's own
    losses_and_accuracies.py / datagenerators.py (and the sklearn
    pairwise_distances it calls) in the build container.
  * backbones.py, step.py — PARITY UNPINNED: they restate Keras layer
    semantics (tensorflow 2.2, image-classifiers, efficientnet — none of which
    is installable here, and the reference has no tests or golden outputs for
    them).  They follow backbones.py / models.py line by line and are only
    self-consistency-checked.
  * mining.batch_hard — PARITY UNPINNED: Hermans batch-hard does not exist in
    the reference (README.md:112 only cites it); BASELINE.json names it.
"""
