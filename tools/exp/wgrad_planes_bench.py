#!/usr/bin/env python3
"""Back-to-back timing of the planes-based weight gradient (csrc/conv_wgrad_planes.hip) against the gather-loop one
(conv.hip) on the ResNet 3x3 stride-1 layer sizes at batch 128 (GPU box only).  HIP events around `iters` launches of each,
alternating A/B/A/B; operands are live-like (ReLU'd activations, small random gradients)."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from embeddingnet_amd import _lib  # noqa: E402
from embeddingnet_amd._lib import check, stream  # noqa: E402


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
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--n", type=int, default=128)
    ap.add_argument("--json")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.lib()
    rows = []
    for (h, c, k) in [(56, 64, 64), (28, 128, 128), (14, 256, 256), (7, 512, 512), (56, 128, 128), (28, 256, 256), (14, 512, 512)]:
        n = args.n
        torch.manual_seed(h)
        x = torch.relu(torch.randn(n, h, h, c, device=dev))
        dy = torch.randn(n, h, h, k, device=dev) * 1e-3
        xp = torch.empty(3 * x.numel(), device=dev, dtype=torch.int16)
        dp = torch.empty(3 * dy.numel(), device=dev, dtype=torch.int16)
        check(lib.embnet_planes_from_f32(x.data_ptr(), x.numel() // c, c, xp.data_ptr(), stream()))
        check(lib.embnet_planes_from_f32(dy.data_ptr(), dy.numel() // k, k, dp.data_ptr(), stream()))
        ws_g = torch.empty(max(lib.embnet_conv2d_wgrad_workspace_bytes(n, c, 3, 3, k, h, h) // 4, 4), device=dev)
        ws_p = torch.empty(max(lib.embnet_conv2d_wgrad_planes_workspace_bytes(n, h, h, c, k) // 4, 4), device=dev)
        dw_g, dw_p = torch.empty(3, 3, c, k, device=dev), torch.empty(3, 3, c, k, device=dev)
        gather = lambda: check(lib.embnet_conv2d_wgrad_f32(x.data_ptr(), dy.data_ptr(), dw_g.data_ptr(), ws_g.data_ptr(), ws_g.numel() * 4,
                                                           n, h, h, c, 3, 3, k, 1, 1, 1, h, h, None, None, 0, stream()))
        planes = lambda: check(lib.embnet_conv2d_wgrad_planes_f32(xp.data_ptr(), dp.data_ptr(), dw_p.data_ptr(), ws_p.data_ptr(),
                                                                  ws_p.numel() * 4, n, h, h, c, k, 1, stream()))
        tg = [timeit(gather, args.iters), 0]
        tp = [timeit(planes, args.iters), 0]
        tg[1] = timeit(gather, args.iters)
        tp[1] = timeit(planes, args.iters)
        flop = 2.0 * n * h * h * 9 * c * k
        err = ((dw_g - dw_p).abs().max() / dw_g.abs().max()).item()
        row = dict(h=h, c=c, k=k, gather_us=[round(t, 1) for t in tg], planes_us=[round(t, 1) for t in tp],
                   gather_tflops=round(flop / min(tg) / 1e6, 1), planes_tflops=round(flop / min(tp) / 1e6, 1),
                   splits=lib.embnet_conv2d_wgrad_planes_splits(n, h, h, c, k), rel_diff=err)
        rows.append(row)
        print(row, flush=True)
    if args.json:
        json.dump(rows, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
