#!/usr/bin/env python3
"""A/B of conv kernel builds in ONE process, interleaved rounds (GPU box only).

  python tools/exp/ab_conv.py build_variants/a.so build_variants/b.so [--shapes rn18] [--only fwd] [--rounds 5]

Every library is a full libembnet_hip.so build (tools/build_variant.sh); each round times `iters` back-to-back launches
of each kernel for each library in turn, and the table shows the median over rounds (us and TFLOP/s) per library.
A long warm-up first, so the chip is at its sustained clock.  Experiments only — nothing in the product uses this.
"""
import argparse
import ctypes
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from embeddingnet_amd import _lib  # noqa: E402

RN18 = [(128, 56, 56, 64, 3, 64, 1, 1), (128, 56, 56, 64, 3, 128, 2, 1), (128, 28, 28, 128, 3, 128, 1, 1),
        (128, 28, 28, 128, 3, 256, 2, 1), (128, 14, 14, 256, 3, 256, 1, 1), (128, 14, 14, 256, 3, 512, 2, 1),
        (128, 7, 7, 512, 3, 512, 1, 1), (128, 56, 56, 64, 1, 128, 2, 0), (128, 224, 224, 4, 7, 64, 2, 3)]
RN50 = [(128, 56, 56, 64, 1, 256, 1, 0), (128, 56, 56, 256, 1, 64, 1, 0), (128, 28, 28, 512, 1, 128, 1, 0),
        (128, 14, 14, 1024, 1, 256, 1, 0), (128, 7, 7, 2048, 1, 512, 1, 0), (128, 7, 7, 512, 1, 2048, 1, 0)]


# every 1x1 geometry of a Keras ResNet50 at C3's 512 images per step (stride on the block's first 1x1 conv), with its count
RN50_1X1 = [(512, 56, 56, 64, 1, 64, 1, 0), (512, 56, 56, 64, 1, 256, 1, 0), (512, 56, 56, 256, 1, 64, 1, 0),
            (512, 56, 56, 256, 1, 128, 2, 0), (512, 56, 56, 256, 1, 512, 2, 0), (512, 28, 28, 128, 1, 512, 1, 0),
            (512, 28, 28, 512, 1, 128, 1, 0), (512, 28, 28, 512, 1, 256, 2, 0), (512, 28, 28, 512, 1, 1024, 2, 0),
            (512, 14, 14, 256, 1, 1024, 1, 0), (512, 14, 14, 1024, 1, 256, 1, 0), (512, 14, 14, 1024, 1, 512, 2, 0),
            (512, 14, 14, 1024, 1, 2048, 2, 0), (512, 7, 7, 512, 1, 2048, 1, 0), (512, 7, 7, 2048, 1, 512, 1, 0)]


def load(path):
    l = ctypes.CDLL(os.path.abspath(path))
    for name, (res, argtypes) in _lib.parse_header().items():
        if hasattr(l, name):
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, argtypes
    return l


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--shapes", default="rn18")
    ap.add_argument("--only", default=None)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    libs = [(os.path.basename(p).replace(".so", ""), load(p)) for p in a.libs]
    shapes = {"rn18": RN18, "rn50": RN50, "rn50_1x1": RN50_1X1}.get(a.shapes) or [tuple(int(v) for v in a.shapes.split(","))]
    st = torch.cuda.current_stream().cuda_stream
    P = lambda t: t.data_ptr()
    tot = {nm: {} for nm, _ in libs}
    print(f"{'shape':34s} {'pass':6s} " + " ".join(f"{nm:>22s}" for nm, _ in libs))
    for (n, h, w, c, ks, k, s_, pad) in shapes:
        oh, ow = (h + 2 * pad - ks) // s_ + 1, (w + 2 * pad - ks) // s_ + 1
        x = torch.randn((n, h, w, c), device=dev)
        wt = torch.randn((ks, ks, c, k), device=dev) * 0.05
        y = torch.empty((n, oh, ow, k), device=dev)
        dy = torch.randn((n, oh, ow, k), device=dev)
        dx, dw = torch.empty_like(x), torch.empty_like(wt)
        flop = 2.0 * n * oh * ow * k * ks * ks * c
        res = {}
        for kind in ("fwd", "dgrad", "wgrad"):
            if a.only and kind != a.only:
                continue
            calls = []
            for nm, l in libs:
                ws = torch.empty(max(l.embnet_conv2d_wgrad_workspace_bytes(n, c, ks, ks, k, oh, ow) // 4, 256), device=dev)
                tws = torch.empty(max(l.embnet_conv2d_fwd_workspace_bytes(n, c, ks, ks, k, oh, ow),
                                      l.embnet_conv2d_dgrad_workspace_bytes(n, h, w, c, ks, ks, k, s_), 1024) // 4, device=dev)
                if kind == "fwd":
                    f = lambda l=l, tws=tws: l.embnet_conv2d_fwd_f32(P(x), P(wt), None, P(y), n, h, w, c, ks, ks, k, s_, pad, pad, oh, ow,
                                                                     0, None, None, None, 0, None, P(tws), tws.numel() * 4, st)
                elif kind == "dgrad":
                    f = lambda l=l, tws=tws: l.embnet_conv2d_dgrad_f32(P(dy), P(wt), P(dx), n, h, w, c, ks, ks, k, s_, pad, pad, oh, ow,
                                                                       0, None, P(tws), tws.numel() * 4, st)
                else:
                    f = lambda l=l, ws=ws: l.embnet_conv2d_wgrad_f32(P(x), P(dy), P(dw), P(ws), ws.numel() * 4, n, h, w, c, ks, ks, k,
                                                                     s_, pad, pad, oh, ow, None, None, 0, st)
                assert f() == 0, l.embnet_last_error()
                calls.append((nm, f))
            for _ in range(30):
                for _, f in calls:
                    f()
            times = {nm: [] for nm, _ in calls}
            for _ in range(a.rounds):
                for nm, f in calls:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(a.iters):
                        f()
                    e1.record()
                    torch.cuda.synchronize()
                    times[nm].append(e0.elapsed_time(e1) / a.iters * 1e3)
            line = f"n{n} {h}x{w}x{c} k{ks} s{s_} -> {k:<5d}".ljust(34) + f" {kind:6s} "
            for nm, _ in calls:
                us = statistics.median(times[nm])
                tot[nm][kind] = tot[nm].get(kind, 0.0) + us
                line += f" {us:9.1f} us {flop / us / 1e6:6.1f} TF"
            # floors: the pass's compulsory HBM bytes at 5 TB/s (what the streaming kernels reach), its FLOP at 200 TFLOP/s fp32-eq
            byts = 4.0 * {"fwd": x.numel() + y.numel(), "dgrad": dy.numel() + dx.numel(), "wgrad": x.numel() + dy.numel()}[kind]
            line += f"   floors: hbm {byts / 5e6:7.1f} us  mfma {flop / 200e6:7.1f} us"
            print(line, flush=True)
    for kind in ("fwd", "dgrad", "wgrad"):
        if any(kind in tot[nm] for nm in tot):
            print(f"{'sum':34s} {kind:6s} " + " ".join(f"{tot[nm].get(kind, 0):12.1f} us      " for nm, _ in libs))


if __name__ == "__main__":
    main()
