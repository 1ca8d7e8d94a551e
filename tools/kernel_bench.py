#!/usr/bin/env python3
"""Micro-benchmarks of single HIP kernels through the C ABI (GPU box only).

  python tools/kernel_bench.py gemm          distance-GEMM sweep: N x E grid, TFLOP/s vs 157.3
  python tools/kernel_bench.py conv          ResNet18 layer shapes at batch 128: fwd / dgrad / wgrad TFLOP/s
  python tools/kernel_bench.py conv --only fwd --shape 128,56,56,64,3,64,1,1

Timing: HIP events on the launch stream around `iters` back-to-back launches after a warm-up.
Algorithmic FLOP: 2*N*N*E (full matrix, no symmetry credit) / 2*N*OH*OW*K*R*S*C.
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from embeddingnet_amd import _lib, ops  # noqa: E402
from embeddingnet_amd._lib import check, ptr, stream  # noqa: E402

PEAK = 157.3


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def gemm_sweep(args):
    dev = torch.device("cuda:0")
    rows = []
    for n in args.n:
        for e in args.e:
            g = torch.Generator(device=dev).manual_seed(7)
            x = torch.rand((n, e), device=dev, generator=g)
            x = x / x.norm(dim=1, keepdim=True)
            lib = _lib.lib()
            ws = torch.empty(max(lib.embnet_pairwise_workspace_bytes(n, e) // 4 + 4, 256), device=dev)     # (room for the K-split slabs)
            d = torch.empty((n, n), device=dev)
            fn = lambda: check(lib.embnet_pairwise_dist_f32(ptr(x), n, e, ptr(d), 0, ptr(ws), ws.numel() * 4, stream()))
            t = timeit(fn, iters=args.iters)
            tf = 2.0 * n * n * e / t / 1e12
            rows.append(dict(N=n, E=e, us=round(t * 1e6, 1), tflops=round(tf, 2), frac=round(tf / PEAK, 4)))
            print(f"pairwise N={n:6d} E={e:5d}  {t * 1e6:10.1f} us  {tf:7.2f} TFLOP/s  {tf / PEAK:6.1%} of fp32 MFMA peak",
                  flush=True)
    # the kNN evaluation's cross-distance (reference models.py:128-142 through sklearn's brute-force kneighbors): queries x gallery
    for (nq, n) in ((1024, 16384), (1024, 4096)):
        for e in args.e:
            if e > 1024:
                continue
            g = torch.Generator(device=dev).manual_seed(8)
            q, x = torch.rand((nq, e), device=dev, generator=g), torch.rand((n, e), device=dev, generator=g)
            lib = _lib.lib()
            ws = torch.empty(lib.embnet_cross_dist_workspace_bytes(nq, n) // 4 + 4, device=dev)
            d = torch.empty((nq, n), device=dev)
            fn = lambda: check(lib.embnet_cross_dist_f32(ptr(q), nq, ptr(x), n, e, ptr(d), 0, ptr(ws), ws.numel() * 4, stream()))
            t = timeit(fn, iters=args.iters)
            tf = 2.0 * nq * n * e / t / 1e12
            rows.append(dict(kind="cross", Q=nq, N=n, E=e, us=round(t * 1e6, 1), tflops=round(tf, 2), frac=round(tf / PEAK, 4)))
            print(f"cross    Q={nq:5d} N={n:6d} E={e:5d}  {t * 1e6:10.1f} us  {tf:7.2f} TFLOP/s  {tf / PEAK:6.1%} of fp32 MFMA peak", flush=True)
    if args.json:
        json.dump(rows, open(args.json, "w"), indent=1)


def losspath_bench(args):
    """Distance matrix + mining + hinge fwd/bwd at the SURVEY §8d sizes (K=4 samples per class)."""
    dev = torch.device("cuda:0")
    rows = []
    for n in args.n:
        for e in args.e:
            p, k = n // 4, 4
            g = torch.Generator(device=dev).manual_seed(7)
            c = torch.rand((p, e), device=dev, generator=g)
            x = (c.repeat_interleave(k, 0) + 0.25 * torch.randn((n, e), device=dev, generator=g)).abs()
            x = (x / x.norm(dim=1, keepdim=True)).requires_grad_(True)
            d = ops.pairwise_distances(x)
            t_pair = timeit(lambda: ops.pairwise_distances(x), iters=args.iters)
            res = {}
            for mode in ("hardest", "semihard"):
                res[mode] = timeit(lambda: ops.mine_triplets(d, p, k, 0.5, mode, seed=1), iters=args.iters)
            t_bh = timeit(lambda: ops.batch_hard(d, p, k), iters=args.iters)
            trip, count, _ = ops.mine_triplets(d, p, k, 0.5, "hardest")
            t_fwd = timeit(lambda: ops.triplet_gather_loss(x, trip, count, 0.5), iters=args.iters)

            def fb():
                x.grad = None
                ops.triplet_gather_loss(x, trip, count, 0.5)[0].backward()
            t_fb = timeit(fb, iters=args.iters)
            row = dict(N=n, E=e, T=int(count.item()), pairwise_us=round(t_pair * 1e6, 1),
                       mine_hardest_us=round(res["hardest"] * 1e6, 1), mine_semihard_us=round(res["semihard"] * 1e6, 1),
                       batch_hard_us=round(t_bh * 1e6, 1), hinge_fwd_us=round(t_fwd * 1e6, 1),
                       hinge_fwd_bwd_us=round(t_fb * 1e6, 1),
                       mine_GBps=round(4.0 * n * n / res["hardest"] / 1e9, 1),
                       hinge_GBps=round(8.0 * n * e / max(t_fb - t_fwd, 1e-9) / 1e9, 1))
            rows.append(row)
            print(row, flush=True)
    if args.json:
        json.dump(rows, open(args.json, "w"), indent=1)


RN18 = [  # n, h, w, c, k(size), cout, stride, pad
    (128, 224, 224, 3, 7, 64, 2, 3),
    (128, 56, 56, 64, 3, 64, 1, 1),
    (128, 56, 56, 64, 1, 64, 1, 0),
    (128, 56, 56, 64, 3, 128, 2, 1),
    (128, 56, 56, 64, 1, 128, 2, 0),
    (128, 28, 28, 128, 3, 128, 1, 1),
    (128, 28, 28, 128, 3, 256, 2, 1),
    (128, 14, 14, 256, 3, 256, 1, 1),
    (128, 14, 14, 256, 3, 512, 2, 1),
    (128, 7, 7, 512, 3, 512, 1, 1),
]


def conv_bench(args):
    dev = torch.device("cuda:0")
    lib = _lib.lib()
    shapes = [tuple(int(v) for v in args.shape.split(","))] if args.shape else RN18
    for (n, h, w, c, ks, k, st, pad) in shapes:
        oh, ow = (h + 2 * pad - ks) // st + 1, (w + 2 * pad - ks) // st + 1
        x = torch.randn((n, h, w, c), device=dev)
        wt = torch.randn((ks, ks, c, k), device=dev) * 0.05
        y = torch.empty((n, oh, ow, k), device=dev)
        dy = torch.randn((n, oh, ow, k), device=dev)
        dx = torch.empty_like(x)
        dw = torch.empty_like(wt)
        ws = torch.empty(max(lib.embnet_conv2d_wgrad_workspace_bytes(n, c, ks, ks, k, oh, ow) // 4, 256), device=dev)
        tws = torch.empty(max(lib.embnet_conv2d_fwd_workspace_bytes(n, c, ks, ks, k, oh, ow),
                              lib.embnet_conv2d_dgrad_workspace_bytes(n, h, w, c, ks, ks, k, st), 1024) // 4, device=dev)
        flop = 2.0 * n * oh * ow * k * ks * ks * c
        calls = {
            "fwd": lambda: check(lib.embnet_conv2d_fwd_f32(ptr(x), ptr(wt), None, ptr(y), n, h, w, c, ks, ks, k, st, pad, pad,
                                                           oh, ow, 0, None, None, None, 0, None, ptr(tws), tws.numel() * 4, stream())),
            "dgrad": lambda: check(lib.embnet_conv2d_dgrad_f32(ptr(dy), ptr(wt), ptr(dx), n, h, w, c, ks, ks, k, st, pad, pad,
                                                               oh, ow, 0, None, ptr(tws), tws.numel() * 4, stream())),
            "wgrad": lambda: check(lib.embnet_conv2d_wgrad_f32(ptr(x), ptr(dy), ptr(dw), ptr(ws), ws.numel() * 4, n, h, w, c,
                                                               ks, ks, k, st, pad, pad, oh, ow, None, None, 0, stream())),
        }
        line = f"n{n} {h}x{w}x{c} k{ks} s{st} -> {k}: "
        for kind, fn in calls.items():
            if args.only and kind != args.only:
                continue
            t = timeit(fn, iters=args.iters)
            line += f" {kind} {t * 1e6:8.1f} us {flop / t / 1e12:6.1f} TF/s |"
        print(line, flush=True)


def libcmp_bench(args):
    """Vendor-library reference points for the same shapes (measurement only; nothing in the product path calls
    them): rocBLAS/hipBLASLt fp32 GEMM through torch.matmul and MIOpen fp32 convolutions through torch (NHWC)."""
    dev = torch.device("cuda:0")
    torch.backends.cuda.matmul.allow_tf32 = False
    torch.backends.cudnn.allow_tf32 = False
    for n, e in [(16384, 256), (16384, 512), (16384, 4096), (4096, 4096)]:
        x = torch.rand((n, e), device=dev)
        t = timeit(lambda: torch.matmul(x, x.t()), iters=args.iters)
        print(f"torch.matmul fp32  N={n} E={e}: {t * 1e6:9.1f} us {2.0 * n * n * e / t / 1e12:6.1f} TF/s", flush=True)
    for (n, h, w, c, ks, k, st, pad) in RN18[1:]:
        x = torch.randn((n, c, h, w), device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        conv = torch.nn.Conv2d(c, k, ks, stride=st, padding=pad, bias=False).to(dev).to(memory_format=torch.channels_last)
        y = conv(x)
        dy = torch.randn_like(y)
        flop = 2.0 * n * y.shape[2] * y.shape[3] * k * ks * ks * c
        tf = timeit(lambda: conv(x), iters=args.iters)

        def fb():
            x.grad = None; conv.weight.grad = None
            conv(x).backward(dy)
        tfb = timeit(fb, iters=args.iters)
        print(f"MIOpen fp32 n{n} {h}x{w}x{c} k{ks} s{st} -> {k}:  fwd {tf * 1e6:8.1f} us {flop / tf / 1e12:6.1f} TF/s | "
              f"fwd+bwd {tfb * 1e6:8.1f} us {3 * flop / tfb / 1e12:6.1f} TF/s", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["gemm", "conv", "losspath", "libcmp"])
    ap.add_argument("--n", type=int, nargs="+", default=[128, 256, 1024, 4096, 16384])
    ap.add_argument("--e", type=int, nargs="+", default=[256, 512, 4096])
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default=None)
    ap.add_argument("--shape", default=None)
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    {"gemm": gemm_sweep, "conv": conv_bench, "losspath": losspath_bench, "libcmp": libcmp_bench}[a.what](a)
