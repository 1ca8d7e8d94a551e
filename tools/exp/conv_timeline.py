#!/usr/bin/env python3
"""Where a conv kernel's time goes, from in-kernel stamps (diagnostic build, GPU box only).

  bash tools/build_variant.sh stamps -DEMBNET_STAMPS=1          (here; the .so travels with gpurun)
  python tools/exp/conv_timeline.py [--shape n,h,w,c,ks,k,stride,pad] [--only fwd|dgrad|wgrad] [--json out.json]

Every workgroup of the stamped build writes 8 x uint64 (gemm_engine.h: stamp()): entry, loaders ready, first K tile
in LDS, main loop done, exit (shader cycles), entry/exit on the chip-wide 100 MHz clock, and its HW_ID/XCC_ID.
Per launch this prints
  * span of the launch and the clock the workgroups saw,
  * per workgroup (cycles): prologue (entry -> first MFMA can issue), main loop, epilogue; main-loop cycles per K
    tile against the matrix-pipe time of that tile (TM*TN*16 MFMAs x 64 cycles): how many waves share a SIMD's pipe,
  * per CU: resident workgroups, matrix-pipe busy fraction = sum of the workgroups' pipe time / the CU's busy span,
    and how much of the span had fewer workgroups in their main loop than were resident (prologue/epilogue bubbles).
Nothing here is part of the product; the product build compiles stamp() to nothing.
"""
import argparse
import ctypes
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("EMBNET_LIB", os.path.join(ROOT, "build_variants", "stamps.so"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from embeddingnet_amd import _lib  # noqa: E402
from embeddingnet_amd._lib import check, ptr, stream  # noqa: E402

RN18 = [(128, 56, 56, 64, 3, 64, 1, 1), (128, 28, 28, 128, 3, 128, 1, 1), (128, 14, 14, 256, 3, 256, 1, 1),
        (128, 7, 7, 512, 3, 512, 1, 1), (128, 56, 56, 64, 3, 128, 2, 1),
        (128, 56, 56, 576, 1, 64, 1, 0)]       # last: the GEMM of the first shape without a 3x3 gather (1x1, C=576)


def analyse(st, bm, bn, tm, tn, kt_full, label, flop, out):
    st = st[st[:, 0] != 0]
    t8, t9 = st[:, 8].astype(np.int64), st[:, 9].astype(np.int64)                      # workgroups that returned before stamping (wgrad padding ids)
    n = len(st)
    t0, rt0, t2, t3, t4, t5, rt5, hw = (st[:, i].astype(np.int64) for i in range(8))
    span_us = (rt5.max() - rt0.min()) / 100.0
    clk = np.median((t5 - t0) / np.maximum(rt5 - rt0, 1)) * 100.0       # MHz
    pro, loop, epi = t3 - t0, t4 - t3, t5 - t4
    ideal_tile = tm * tn * 16 * 64
    # K tiles of a workgroup: full tiles run kt_full; K-split parts fewer — estimate from its loop time vs the median
    med_loop = np.median(loop)
    kt = np.where(loop > 0.75 * med_loop, kt_full, np.maximum(np.round(kt_full * loop / med_loop), 1))
    per_tile = loop / kt
    cu = ((hw >> 32) & 0xF) * 256 + ((hw >> 8) & 0xFF)
    busy, frac_full, res = [], [], []
    for c in np.unique(cu):
        m = cu == c
        a0, a5 = rt0[m], rt5[m]
        lo, hi = a0.min(), a5.max()
        # residency over time on the 10 ns clock
        ev = sorted([(x, 1) for x in a0] + [(x, -1) for x in a5])
        cur = peak = 0
        for _, d in ev:
            cur += d
            peak = max(peak, cur)
        res.append(peak)
        span_cyc = (hi - lo) * clk / 100.0
        busy.append((kt[m] * ideal_tile).sum() / max(span_cyc, 1))
        # fraction of the CU's span during which `peak` workgroups were inside their main loops (entry..exit mapped
        # linearly from cycles to the 10 ns clock per workgroup)
        f = (a5 - a0) / np.maximum(t5[m] - t0[m], 1)
        l0, l1 = a0 + (t3[m] - t0[m]) * f, a0 + (t4[m] - t0[m]) * f
        ev = sorted([(x, 1) for x in l0] + [(x, -1) for x in l1])
        cur, last, full = 0, lo, 0.0
        for x, d in ev:
            if cur >= peak:
                full += x - last
            cur += d
            last = x
        frac_full.append(full / max(hi - lo, 1))
    row = dict(kernel=label, workgroups=int(n), span_us=round(float(span_us), 1), tflops=round(flop / span_us / 1e6, 1),
               clock_mhz=round(float(clk)), tile=f"{bm}x{bn}", kt=int(kt_full),
               prologue_cyc=int(np.median(pro)), loop_cyc=int(med_loop), epilogue_cyc=int(np.median(epi)),
               loop_cyc_per_ktile=round(float(np.median(per_tile))), pipe_cyc_per_ktile=ideal_tile,
               waves_sharing_pipe=round(float(np.median(per_tile)) / ideal_tile, 2),
               resident_wg_per_cu=float(np.median(res)), cu_pipe_busy=round(float(np.mean(busy)), 3),
               cu_span_all_in_loop=round(float(np.mean(frac_full)), 3),
               wg_share_prologue=round(float(pro.sum() / (t5 - t0).sum()), 3),
               wg_share_epilogue=round(float(epi.sum() / (t5 - t0).sum()), 3))
    late = rt0 > rt0.min() + 300                # workgroups dispatched into a busy CU (later than 3 us after the first)
    for nm, sel in (("first_round", ~late), ("later_rounds", late)):
        if sel.sum():
            row[nm] = dict(n=int(sel.sum()), init=int(np.median((t2 - t0)[sel])), issue=int(np.median((t8 - t2)[sel])),
                           landed=int(np.median((t9 - t8)[sel])), barrier=int(np.median((t3 - t9)[sel])),
                           loop_per_kt=int(np.median(per_tile[sel])), epilogue=int(np.median(epi[sel])))
    out.append(row)
    print(json.dumps(row), flush=True)
    # one CU's timeline (us from the launch start)
    c = np.unique(cu)[len(np.unique(cu)) // 2]
    m = np.where(cu == c)[0]
    m = m[np.argsort(rt0[m])][:16]
    base = rt0.min()
    print("   CU %d: " % c + " ".join(f"[{(rt0[i] - base) / 100:.1f}-{(rt5[i] - base) / 100:.1f}]" for i in m), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default=None)
    ap.add_argument("--only", default=None)
    ap.add_argument("--json", default=None)
    ap.add_argument("--gemm", action="store_true")
    ap.add_argument("--warm", type=int, default=5, help="launches before the stamped one (sustained-load clock: ~200)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.lib()
    raw = ctypes.CDLL(os.environ["EMBNET_LIB"])
    raw.embnet_debug_set_stamps.argtypes = [ctypes.c_void_p]
    shapes = [tuple(int(v) for v in a.shape.split(","))] if a.shape else RN18
    out = []
    cap = 1 << 16
    stamps = torch.zeros((cap, 16), dtype=torch.int64, device=dev)
    for (n, h, w, c, ks, k, st_, pad) in shapes:
        oh, ow = (h + 2 * pad - ks) // st_ + 1, (w + 2 * pad - ks) // st_ + 1
        x = torch.randn((n, h, w, c), device=dev)
        wt = torch.randn((ks, ks, c, k), device=dev) * 0.05
        y = torch.empty((n, oh, ow, k), device=dev)
        dy = torch.randn((n, oh, ow, k), device=dev)
        dx, dw = torch.empty_like(x), torch.empty_like(wt)
        wsb = lib.embnet_conv2d_wgrad_workspace_bytes(n, c, ks, ks, k, oh, ow)
        ws = torch.empty(max(wsb // 4, 256), device=dev)
        tws = torch.empty(max(lib.embnet_conv2d_fwd_workspace_bytes(n, c, ks, ks, k, oh, ow),
                              lib.embnet_conv2d_dgrad_workspace_bytes(n, h, w, c, ks, ks, k, st_), 1024) // 4, device=dev)
        flop = 2.0 * n * oh * ow * k * ks * ks * c
        calls = {
            "fwd": lambda: check(lib.embnet_conv2d_fwd_f32(ptr(x), ptr(wt), None, ptr(y), n, h, w, c, ks, ks, k, st_, pad, pad,
                                                           oh, ow, 0, None, None, None, 0, None, ptr(tws), tws.numel() * 4, stream())),
            "dgrad": lambda: check(lib.embnet_conv2d_dgrad_f32(ptr(dy), ptr(wt), ptr(dx), n, h, w, c, ks, ks, k, st_, pad, pad,
                                                               oh, ow, 0, None, ptr(tws), tws.numel() * 4, stream())),
            "wgrad": lambda: check(lib.embnet_conv2d_wgrad_slabs_f32(ptr(x), ptr(dy), ptr(dw), ptr(ws), ws.numel() * 4, n, h, w, c,
                                                                     ks, ks, k, st_, pad, pad, oh, ow, None, None, 0, stream())),
        }
        for kind, fn in calls.items():
            if a.only and kind != a.only:
                continue
            if kind == "dgrad" and st_ != 1:
                continue                                # several classes per launch: one stamp row per (x,y) id still works
            raw.embnet_debug_set_stamps(None)
            for _ in range(a.warm):
                fn()                                    # warm: clocks, caches
            stamps.zero_()
            raw.embnet_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
            fn()
            torch.cuda.synchronize()
            raw.embnet_debug_set_stamps(None)
            name = lib.embnet_conv2d_kernel_name({"fwd": 0, "dgrad": 1, "wgrad": 2}[kind], n, h, w, c, ks, ks, k, oh, ow).decode()
            bm, bn, wm_, wn_ = (int(v) for v in re.search(r"Geom<(\d+), (\d+), (\d+), (\d+)>", name).groups())
            tm, tn = bm // wm_ // 32, bn // wn_ // 32
            if kind == "wgrad":
                rows = ks * ks * c
                splits = max(wsb // (rows * k * 4), 1)
                kt_full = -(-(-(-(n * oh * ow) // 32)) // splits)
            else:
                kt_full = -(-(ks * ks * (c if kind == "fwd" else k)) // 32)
            s = stamps.cpu().numpy().astype(np.uint64)
            analyse(s, bm, bn, tm, tn, kt_full, f"{kind} n{n} {h}x{w}x{c} k{ks} s{st_} -> {k}", flop, out)
    if a.gemm:                                          # the same main loop as a plain GEMM (distance matrix)
        raw.embnet_debug_set_stamps_pairwise.argtypes = [ctypes.c_void_p]
        for n, e in [(8192, 576), (8192, 4096)]:
            xg = torch.rand((n, e), device=dev)
            d = torch.empty((n, n), device=dev)
            wsg = torch.empty(max(n, 256), device=dev)
            fn = lambda: check(lib.embnet_pairwise_dist_f32(ptr(xg), n, e, ptr(d), 0, ptr(wsg), wsg.numel() * 4, stream()))
            for _ in range(max(a.warm // 8, 3)):
                fn()
            stamps.zero_()
            raw.embnet_debug_set_stamps_pairwise(ctypes.c_void_p(stamps.data_ptr()))
            fn()
            torch.cuda.synchronize()
            raw.embnet_debug_set_stamps_pairwise(None)
            analyse(stamps.cpu().numpy().astype(np.uint64), 128, 128, 2, 2, -(-e // 32), f"pairwise N={n} E={e}",
                    2.0 * n * n * e, out)
    if a.json:
        json.dump(out, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
