#!/usr/bin/env python3
"""Per-size timing of the fused squeeze-and-excite gate launches (csrc/se_mlp.hip) at EfficientNet-B0's block sizes, batch 256
(HIP events around 50 launches each; GPU box only)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from embeddingnet_amd import _lib  # noqa: E402
from embeddingnet_amd._lib import check, stream  # noqa: E402


def timeit(fn, iters=50, warm=5):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def timeit_cold(fn, flush, iters=20):
    """Each launch behind a pass over a 1 GB buffer (what the step's big tensors do to L2 / MALL); events around the launch only."""
    tot = 0.0
    for _ in range(iters):
        flush.add_(1.0)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record()
        torch.cuda.synchronize()
        tot += s.elapsed_time(e)
    return tot / iters * 1e3


def main():
    dev = torch.device("cuda:0")
    flush = torch.zeros(256 * 1024 * 1024, device=dev)
    lib = _lib.lib()
    n = int(os.environ.get("N", 256))
    for (c, s) in [(32, 8), (96, 4), (144, 6), (240, 10), (480, 20), (672, 28), (1152, 48)]:
        t = lambda *shape: torch.randn(*shape, device=dev)
        pooled, w1, b1, w2, b2 = t(n, c), t(c, s) * 0.1, t(s), t(s, c) * 0.1, t(c)
        z1, gate, dgate = torch.empty(n, s, device=dev), torch.empty(n, c, device=dev), t(n, c)
        dz1, dp = torch.empty(n, s, device=dev), torch.empty(n, c, device=dev)
        dw1, db1, dw2, db2 = torch.empty(c, s, device=dev), torch.empty(s, device=dev), torch.empty(s, c, device=dev), torch.empty(c, device=dev)
        fwd = lambda: check(lib.embnet_se_mlp_fwd(pooled.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), n, c, s,
                                                  z1.data_ptr(), gate.data_ptr(), stream()))
        bwd = lambda: check(lib.embnet_se_mlp_bwd(dgate.data_ptr(), gate.data_ptr(), z1.data_ptr(), pooled.data_ptr(), w1.data_ptr(), w2.data_ptr(),
                                                  n, c, s, dz1.data_ptr(), dp.data_ptr(), dw1.data_ptr(), db1.data_ptr(), dw2.data_ptr(),
                                                  db2.data_ptr(), stream()))
        fwd()
        print(json.dumps({"c": c, "s": s, "fwd_us": round(timeit(fwd), 1), "bwd_a_plus_b_us": round(timeit(bwd), 1),
                          "fwd_cold_us": round(timeit_cold(fwd, flush), 1), "bwd_cold_us": round(timeit_cold(bwd, flush), 1)}))


if __name__ == "__main__":
    main()
