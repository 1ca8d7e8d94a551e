#!/usr/bin/env python3
"""Loss path of the fused training step at the BASELINE config sizes: launches and microseconds, fused vs separate.

  python tools/losspath_bench.py [--json profiles/r02_losspath.json]          (GPU box)

For C1 (8x4 = 32 rows, E = 256, semihard), C2 (32x4 = 128, E = 256, hardest), C2 batch-hard and C5 (64x4 = 256, E = 512,
semihard) the forward + backward of the loss path is run both ways on the same clustered embeddings:
  separate: ops.pairwise_distances -> ops.mine_triplets / ops.batch_hard -> ops.triplet_gather_loss -> backward
  fused:    ops.fused_triplet_loss (one forward launch) -> backward
Reported per variant: library launches per pass and the sum of their device times (embnet_trace_*: HIP events on the
launch stream), and the wall time per pass of back-to-back passes (launch overheads overlap with the device work).
SURVEY.md §8d asks for microseconds and launch counts at these sizes: the path is latency-bound, not bandwidth-bound.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from embeddingnet_amd import _lib, ops  # noqa: E402

CASES = [("c1", 8, 4, 256, "semihard"), ("c2", 32, 4, 256, "hardest"), ("c2_batch_hard", 32, 4, 256, "batch_hard"),
         ("c5", 64, 4, 512, "semihard")]


def measure(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    _lib.trace_enable(True); _lib.trace_reset()
    fn()
    torch.cuda.synchronize()
    rec = _lib.trace_records()
    _lib.trace_enable(False)
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / iters
    return dict(launches=len(rec), device_us=round(1e3 * sum(r[1] for r in rec), 1), wall_us_per_pass=round(1e6 * wall, 1),
                kernels={r[0].replace("embnet::", ""): round(1e3 * r[1], 1) for r in rec})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    out = []
    for name, p, k, e, mode in CASES:
        n = p * k
        g = torch.Generator(device=dev).manual_seed(7)
        c = torch.rand((p, e), device=dev, generator=g)
        x = (c.repeat_interleave(k, 0) + 0.25 * torch.randn((n, e), device=dev, generator=g)).abs()
        x = (x / x.norm(dim=1, keepdim=True)).requires_grad_(True)

        def separate():
            x.grad = None
            with torch.no_grad():
                d = ops.pairwise_distances(x)
                if mode == "batch_hard":
                    trip, count = ops.batch_hard(d, p, k)
                else:
                    trip, count, _ = ops.mine_triplets(d, p, k, 0.5, mode, seed=1)
            ops.triplet_gather_loss(x, trip, count, 0.5)[0].backward()

        def fused():
            x.grad = None
            ops.fused_triplet_loss(x, p, k, 0.5, mode, seed=1)[0].backward()

        row = dict(config=name, N=n, E=e, mining=mode, separate=measure(separate, a.iters))
        if ops.fused_loss_supported(p, k, e):
            row["fused"] = measure(fused, a.iters)
        out.append(row)
        print(json.dumps(row), flush=True)
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
