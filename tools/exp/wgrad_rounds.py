#!/usr/bin/env python3
"""Weight-gradient split count: whole rounds of workgroups (EMBNET_WGRAD_BLOCKS = 512 / 1024 / 1536 / 2048) per layer."""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from embeddingnet_amd import _lib
from tools.exp.ab_conv import RN18, RN50
dev = torch.device("cuda:0"); l = _lib.lib(); st = torch.cuda.current_stream().cuda_stream; P = lambda t: t.data_ptr()
big = torch.empty(1 << 28, device=dev)
for (n, h, w, c, ks, k, s_, pad) in RN18 + RN50:
    oh, ow = (h + 2 * pad - ks) // s_ + 1, (w + 2 * pad - ks) // s_ + 1
    x = torch.randn((n, h, w, c), device=dev); dy = torch.randn((n, oh, ow, k), device=dev); dw = torch.empty((ks, ks, c, k), device=dev)
    f = lambda: l.embnet_conv2d_wgrad_f32(P(x), P(dy), P(dw), P(big), big.numel() * 4, n, h, w, c, ks, ks, k, s_, pad, pad, oh, ow, None, None, 0, st)
    res = {}
    for b in (0, 512, 1024, 1536, 2048):
        if b: os.environ["EMBNET_WGRAD_BLOCKS"] = str(b)
        else: os.environ.pop("EMBNET_WGRAD_BLOCKS", None)
        for _ in range(15): f()
        ts = []
        for _ in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 20 * 1e3)
        res[b] = statistics.median(ts)
    rows, tiles = ks * ks * c, -(-ks * ks * c // 128) * -(-k // 128)
    print(f"n{n} {h}x{w}x{c} k{ks} s{s_} -> {k:<5d} rows {rows:5d} tiles128 {tiles:4d}  default {res[0]:7.1f} | " + "  ".join(f"{b}: {res[b]:7.1f}" for b in (512, 1024, 1536, 2048)), flush=True)
