#!/usr/bin/env python3
"""Back-to-back timing of the stride-1 depthwise layers of EfficientNet-B0 at batch 256 from 28x28 down (forward with statistics,
data gradient with the BatchNorm-backward sums, weight gradient with its slab sum): run once with EMBNET_DW_TILE=0 (row kernels) and once with =1 (LDS-tile kernel);
the knob is read once per process.  GPU box only."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from embeddingnet_amd import _lib  # noqa: E402
from embeddingnet_amd._lib import check, stream  # noqa: E402

LAYERS = [(28, 240, 5), (14, 480, 3), (14, 480, 5), (14, 672, 5), (7, 1152, 5), (7, 1152, 3)]


def timeit(fn, iters, warm=3):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=40)
    ap.add_argument("--n", type=int, default=256)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.lib()
    n = args.n
    # several tensors cycled so that a launch does not find its input in the L2 / MALL the previous one left
    for (h, c, k) in LAYERS:
        pad = (k - 1) // 2
        xs = [torch.randn(n, h, h, c, device=dev) for _ in range(6)]
        es = [torch.randn(n, h, h, c, device=dev) for _ in range(6)]
        ys = [torch.empty(n, h, h, c, device=dev) for _ in range(6)]
        w = torch.randn(k, k, c, 1, device=dev)
        vec = [torch.rand(c, device=dev) + 0.5 for _ in range(4)]
        rows_f = lib.embnet_dwconv2d_fwd_stats_rows(n, c, k, k, 1, h, h)
        rows_b = lib.embnet_dwconv2d_dgrad_bnsums_rows(n, h, h, c, k, k, 1)
        st_f = torch.zeros(2, c, rows_f, device=dev)
        st_b = torch.zeros(2, c, rows_b, device=dev)
        i = [0]

        def fwd():
            j = i[0] = (i[0] + 1) % 6
            check(lib.embnet_dwconv2d_fwd_stats_f32(xs[j].data_ptr(), w.data_ptr(), ys[j].data_ptr(), n, h, h, c, k, k, 1, pad, pad, h, h,
                                                    st_f.data_ptr(), stream()))

        def bwd():
            j = i[0] = (i[0] + 1) % 6
            check(lib.embnet_dwconv2d_dgrad_bnsums_f32(xs[j].data_ptr(), w.data_ptr(), ys[j].data_ptr(), n, h, h, c, k, k, 1, pad, pad, h, h,
                                                       es[j].data_ptr(), vec[0].data_ptr(), vec[1].data_ptr(), vec[2].data_ptr(),
                                                       vec[3].data_ptr(), 2, st_b.data_ptr(), rows_b, stream()))

        ws = torch.empty(max(lib.embnet_dwconv2d_wgrad_workspace_bytes(n, c, k, k, h, h) // 4, 4), device=dev)
        dwt = torch.empty(k, k, c, 1, device=dev)

        def wgr():
            j = i[0] = (i[0] + 1) % 6
            check(lib.embnet_dwconv2d_wgrad_f32(xs[j].data_ptr(), es[j].data_ptr(), dwt.data_ptr(), ws.data_ptr(), ws.numel() * 4,
                                                n, h, h, c, k, k, 1, pad, pad, h, h, stream()))

        tw = min(timeit(wgr, args.iters), timeit(wgr, args.iters))
        tf = min(timeit(fwd, args.iters), timeit(fwd, args.iters))
        tb = min(timeit(bwd, args.iters), timeit(bwd, args.iters))
        el = n * h * h * c * 4.0
        print(json.dumps({"h": h, "c": c, "k": k, "fwd_us": round(tf, 1), "fwd_GBs": round(2 * el / tf / 1e3, 0), "fwd_rows": rows_f,
                          "dgrad_us": round(tb, 1), "dgrad_GBs": round(3 * el / tb / 1e3, 0), "dgrad_rows": rows_b,
                          "wgrad_us": round(tw, 1), "wgrad_GBs": round(2 * el / tw / 1e3, 0)}))


if __name__ == "__main__":
    main()
