#!/usr/bin/env python3
"""Tile / split-count sweep of the conv kernels on the ResNet18 and ResNet50 layer shapes (GPU box only).

  python tools/exp/tile_sweep.py [--shapes rn18] [--only fwd]

Per shape and pass: time with the library's own choice, then with every tile forced (EMBNET_CONV_TILE for forward /
data gradient, EMBNET_WGRAD_TILE x EMBNET_WGRAD_BLOCKS for the weight gradient; the library reads them per call).
Tiles: 0 = 128x128, 1 = 128x64, 2 = 128x32, 3 = 64x64, 4 = 192x64 (weight gradient only).  Experiments only.
"""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from embeddingnet_amd import _lib  # noqa: E402
from tools.exp.ab_conv import RN18, RN50  # noqa: E402

TILE = {0: "128x128", 1: "128x64", 2: "128x32", 3: "64x64", 4: "192x64"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="rn18")
    ap.add_argument("--only", default=None)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    l = _lib.lib()
    st = torch.cuda.current_stream().cuda_stream
    P = lambda t: t.data_ptr()
    shapes = {"rn18": RN18, "rn50": RN50}.get(a.shapes) or [tuple(int(v) for v in a.shapes.split(","))]

    def timed(f):
        for _ in range(10):
            f()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                f()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / a.iters * 1e3)
        return statistics.median(ts)

    for (n, h, w, c, ks, k, s_, pad) in shapes:
        oh, ow = (h + 2 * pad - ks) // s_ + 1, (w + 2 * pad - ks) // s_ + 1
        x = torch.randn((n, h, w, c), device=dev)
        wt = torch.randn((ks, ks, c, k), device=dev) * 0.05
        y = torch.empty((n, oh, ow, k), device=dev)
        dy = torch.randn((n, oh, ow, k), device=dev)
        dx, dw = torch.empty_like(x), torch.empty_like(wt)
        flop = 2.0 * n * oh * ow * k * ks * ks * c
        big = torch.empty(1 << 28, device=dev)                    # 1 GiB workspace: enough for any forced plan here
        for kind in ("fwd", "dgrad", "wgrad"):
            if a.only and kind != a.only:
                continue
            if kind == "fwd":
                f = lambda: l.embnet_conv2d_fwd_f32(P(x), P(wt), None, P(y), n, h, w, c, ks, ks, k, s_, pad, pad, oh, ow,
                                                    0, None, None, None, 0, None, P(big), big.numel() * 4, st)
            elif kind == "dgrad":
                f = lambda: l.embnet_conv2d_dgrad_f32(P(dy), P(wt), P(dx), n, h, w, c, ks, ks, k, s_, pad, pad, oh, ow,
                                                      0, None, P(big), big.numel() * 4, st)
            else:
                f = lambda: l.embnet_conv2d_wgrad_f32(P(x), P(dy), P(dw), P(big), big.numel() * 4, n, h, w, c, ks, ks, k,
                                                      s_, pad, pad, oh, ow, None, None, 0, st)
            for v in ("EMBNET_CONV_TILE", "EMBNET_WGRAD_TILE", "EMBNET_WGRAD_BLOCKS"):
                os.environ.pop(v, None)
            assert f() == 0, l.embnet_last_error()
            base = timed(f)
            res = {}
            if kind != "wgrad":
                for t in (0, 1, 2, 3):
                    os.environ["EMBNET_CONV_TILE"] = str(t)
                    if f() != 0:
                        continue
                    res[TILE[t]] = timed(f)
                os.environ.pop("EMBNET_CONV_TILE")
            else:
                for t in (0, 1, 3, 4):
                    for blocks in (512, 768, 1024, 1536, 2048, 3072):
                        os.environ["EMBNET_WGRAD_TILE"] = str(t); os.environ["EMBNET_WGRAD_BLOCKS"] = str(blocks)
                        if f() != 0:
                            continue
                        res[f"{TILE[t]}/{blocks}"] = timed(f)
                os.environ.pop("EMBNET_WGRAD_TILE"); os.environ.pop("EMBNET_WGRAD_BLOCKS")
            best = min(res, key=res.get)
            line = f"n{n} {h}x{w}x{c} k{ks} s{s_} -> {k:<5d}".ljust(34) + f" {kind:6s} default {base:7.1f} us {flop / base / 1e6:6.1f} TF | " \
                f"best {best:14s} {res[best]:7.1f} us {flop / res[best] / 1e6:6.1f} TF | "
            if kind != "wgrad":
                line += "  ".join(f"{t} {v:7.1f}" for t, v in res.items())
            else:
                top = sorted(res, key=res.get)[:5]
                line += "  ".join(f"{t} {res[t]:7.1f}" for t in top)
            print(line, flush=True)


if __name__ == "__main__":
    main()
