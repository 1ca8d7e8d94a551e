#!/usr/bin/env python3
"""Experiment (GPU box): conv forward on PRE-SPLIT bf16 planes with LDS-DMA staging (csrc/conv_planes.hip) against the
library's in-loop-split kernel (csrc/conv.hip), same process, interleaved rounds.  Checks bit-identity first.

  python tools/exp/ab_planes.py [--shapes rn18] [--rounds 5] [--tiles 0,1,2,3]
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
        (128, 7, 7, 512, 3, 512, 1, 1), (128, 56, 56, 64, 1, 128, 2, 0)]
TILE = {0: "128x64/4w", 1: "256x64/8w", 2: "256x128/8w", 3: "128x128/4w"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--tiles", default="0,1,2,3")
    ap.add_argument("--first", type=int, default=99)
    a = ap.parse_args()
    tiles = [int(t) for t in a.tiles.split(",")]
    dev = torch.device("cuda:0")
    l = _lib.lib()
    vp = ctypes.c_void_p
    l.embnet_split_planes_f32.argtypes = [vp, ctypes.c_long, vp, vp]
    l.embnet_conv2d_fwd_planes.argtypes = [vp, vp, vp] + [ctypes.c_int] * 12 + [vp, vp, ctypes.c_int, vp]
    l.embnet_prep_weight_planes.argtypes = [vp] + [ctypes.c_int] * 5 + [vp, vp]
    l.embnet_conv2d_patch_planes.argtypes = [vp, vp, vp, vp] + [ctypes.c_int] * 12 + [vp, vp, vp, ctypes.c_size_t, vp]
    l.embnet_conv2d_patch_workspace_bytes.restype = ctypes.c_size_t
    l.embnet_split_planes_cm_f32.argtypes = [vp, ctypes.c_long, ctypes.c_int, vp, vp]
    l.embnet_prep_weight_planes2.argtypes = [vp] + [ctypes.c_int] * 5 + [vp, vp]
    l.embnet_conv2d_patch2_planes.argtypes = [vp, vp, vp, vp] + [ctypes.c_int] * 12 + [vp, vp, vp, ctypes.c_size_t, vp]
    l.embnet_conv2d_patch2_workspace_bytes.restype = ctypes.c_size_t
    st = torch.cuda.current_stream().cuda_stream
    P = lambda t: t.data_ptr()
    print(f"{'shape':34s} {'library (in-loop split)':>26s} " + " ".join(f"{TILE[t]:>22s}" for t in tiles) + f" {'patch 256xN/8w':>22s} {'patch2 8+2 waves':>22s}")
    for (n, h, w, c, ks, k, s_, pad) in RN18[:a.first]:
        oh, ow = (h + 2 * pad - ks) // s_ + 1, (w + 2 * pad - ks) // s_ + 1
        x = torch.randn((n, h, w, c), device=dev).abs_()
        wt = torch.randn((ks, ks, c, k), device=dev) * 0.05
        y0 = torch.empty((n, oh, ow, k), device=dev)
        y1 = torch.empty_like(y0)
        xp = torch.empty((3, x.numel()), device=dev, dtype=torch.int16)
        assert l.embnet_split_planes_f32(P(x), x.numel(), P(xp), st) == 0
        wtt = wt.reshape(ks * ks * c, k).t().contiguous()                      # [K][R*S*C]
        wp = torch.empty((3, wtt.numel()), device=dev, dtype=torch.int16)
        assert l.embnet_split_planes_f32(P(wtt), wtt.numel(), P(wp), st) == 0
        flop = 2.0 * n * oh * ow * k * ks * ks * c
        ws = torch.empty(max(l.embnet_conv2d_fwd_workspace_bytes(n, c, ks, ks, k, oh, ow), 1024) // 4, device=dev)
        calls = [("lib", lambda: l.embnet_conv2d_fwd_f32(P(x), P(wt), None, P(y0), n, h, w, c, ks, ks, k, s_, pad, pad, oh, ow,
                                                         0, None, None, None, 0, None, P(ws), ws.numel() * 4, st))]
        for t in tiles:
            calls.append((TILE[t], lambda t=t: l.embnet_conv2d_fwd_planes(P(xp), P(wp), P(y1), n, h, w, c, ks, ks, k, s_, pad, pad,
                                                                       oh, ow, None, None, t, st)))
        patch = s_ == 1 and l.embnet_conv2d_patch_supported(n, c, ks, ks, k, s_, oh, ow)
        if patch:
            wp2 = torch.empty((3, wt.numel()), device=dev, dtype=torch.int16)
            assert l.embnet_prep_weight_planes(P(wt), ks, ks, c, k, 0, P(wp2), st) == 0
            assert torch.equal(wp2, wp), "prep_weight_planes differs from transpose + split"
            pws = torch.empty(max(l.embnet_conv2d_patch_workspace_bytes(n, c, ks, ks, k, s_, oh, ow), 1024) // 4, device=dev)
            y2 = torch.empty_like(y0)
            calls.append(("patch", lambda: l.embnet_conv2d_patch_planes(P(xp), P(wp2), None, P(y2), n, h, w, c, ks, ks, k, pad, pad, oh, ow,
                                                                        0, None, None, P(pws), pws.numel() * 4, st)))
        patch2 = s_ == 1 and l.embnet_conv2d_patch2_supported(n, c, ks, ks, k, oh, ow)
        if patch2:
            xc = torch.empty((3, x.numel()), device=dev, dtype=torch.int16)
            assert l.embnet_split_planes_cm_f32(P(x), n * h * w, c, P(xc), st) == 0
            wp3 = torch.empty((3, wt.numel()), device=dev, dtype=torch.int16)
            assert l.embnet_prep_weight_planes2(P(wt), ks, ks, c, k, 0, P(wp3), st) == 0
            pws2 = torch.empty(max(l.embnet_conv2d_patch2_workspace_bytes(n, c, ks, ks, k, oh, ow), 1024) // 4, device=dev)
            y3 = torch.empty_like(y0)
            calls.append(("patch2", lambda: l.embnet_conv2d_patch2_planes(P(xc), P(wp3), None, P(y3), n, h, w, c, ks, ks, k, pad, pad, oh, ow,
                                                                          0, None, None, P(pws2), pws2.numel() * 4, st)))
        # correctness: bit-identical to the unsplit-tail library launch
        assert l.embnet_conv2d_fwd_f32(P(x), P(wt), None, P(y0), n, h, w, c, ks, ks, k, s_, pad, pad, oh, ow, 0, None, None,
                                       None, 0, None, None, 0, st) == 0
        for t in tiles:
            y1.fill_(float("nan"))
            assert calls[1 + tiles.index(t)][1]() == 0, l.embnet_last_error()
            torch.cuda.synchronize()
            same = torch.equal(y0, y1)
            if not same:
                d = (y0 - y1).abs()
                print(f"  tile {TILE[t]}: NOT bit-identical: max |d| {d.max().item():.3e} (nan: {torch.isnan(y1).sum().item()})")
        if patch2:
            y3.fill_(float("nan"))
            assert calls[-1][1]() == 0, l.embnet_last_error()
            torch.cuda.synchronize()
            err = ((y3 - y0).abs().max() / y0.abs().max()).item()
            if not err < 2e-6:
                print(f"  patch2 kernel: max |d| / max |y| = {err:.3e} (nan: {torch.isnan(y3).sum().item()})")
        if patch:
            y2.fill_(float("nan"))
            assert calls[-1 - int(bool(patch2))][1]() == 0, l.embnet_last_error()
            torch.cuda.synchronize()
            err = ((y2 - y0).abs().max() / y0.abs().max()).item()
            if not err < 2e-6:
                print(f"  patch kernel: max |d| / max |y| = {err:.3e} (nan: {torch.isnan(y2).sum().item()})")
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
        line = f"n{n} {h}x{w}x{c} k{ks} s{s_} -> {k:<5d}".ljust(34)
        for nm, _ in calls:
            us = statistics.median(times[nm])
            line += f" {us:9.1f} us {flop / us / 1e6:6.1f} TF" + ("    " if nm == "lib" else "")
        print(line, flush=True)


if __name__ == "__main__":
    main()
