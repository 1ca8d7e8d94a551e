#!/usr/bin/env python3
"""Experiment (GPU box, -DEMBNET_PLANES_STAMPS=1 build given as EMBNET_LIB): where a conv_fwd_planes workgroup spends its
time — entry -> first K tile landed -> main loop done -> exit, per workgroup, plus the launch's span."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from embeddingnet_amd import _lib

SHAPES = [(128, 56, 56, 64, 3, 64, 1, 1), (128, 28, 28, 128, 3, 128, 1, 1), (128, 7, 7, 512, 3, 512, 1, 1)]
TILE = {0: (128, 64), 1: (256, 64), 2: (256, 128), 3: (128, 128)}
dev = torch.device("cuda:0")
l = _lib.lib()
vp = ctypes.c_void_p
l.embnet_split_planes_f32.argtypes = [vp, ctypes.c_long, vp, vp]
l.embnet_conv2d_fwd_planes.argtypes = [vp, vp, vp] + [ctypes.c_int] * 12 + [vp, vp, ctypes.c_int, vp]
l.embnet_debug_set_planes_stamps.argtypes = [vp]
st = torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr()
for (n, h, w, c, ks, k, s_, pad) in SHAPES:
    oh, ow = (h + 2 * pad - ks) // s_ + 1, (w + 2 * pad - ks) // s_ + 1
    x = torch.randn((n, h, w, c), device=dev).abs_()
    wt = torch.randn((ks, ks, c, k), device=dev) * 0.05
    y = torch.empty((n, oh, ow, k), device=dev)
    xp = torch.empty((3, x.numel()), device=dev, dtype=torch.int16)
    l.embnet_split_planes_f32(P(x), x.numel(), P(xp), st)
    wtt = wt.reshape(ks * ks * c, k).t().contiguous()
    wp = torch.empty((3, wtt.numel()), device=dev, dtype=torch.int16)
    l.embnet_split_planes_f32(P(wtt), wtt.numel(), P(wp), st)
    for tile in (int(t) for t in os.environ.get("TILES", "0,1,2").split(",")):
        bm, bn = TILE[tile]
        ntiles = -(-n * oh * ow // bm) * -(-k // bn)
        buf = torch.zeros((ntiles, 8), device=dev, dtype=torch.int64)
        run = lambda: l.embnet_conv2d_fwd_planes(P(xp), P(wp), P(y), n, h, w, c, ks, ks, k, s_, pad, pad, oh, ow, None, None, tile, st)
        for _ in range(20):
            run()
        l.embnet_debug_set_planes_stamps(P(buf))
        run()
        torch.cuda.synchronize()
        l.embnet_debug_set_planes_stamps(None)
        t = buf.cpu().numpy().astype(np.float64)[:, :4] * 0.01          # us
        span = t[:, 3].max() - t[:, 0].min()
        pro, loop, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
        kt = -(-ks * ks * c // 32)
        print(f"n{n} {h}x{w}x{c} k{ks} -> {k} tile {bm}x{bn}: {ntiles} workgroups, span {span:.1f} us | per workgroup (median / p90 us): "
              f"first tile {np.median(pro):.2f} / {np.percentile(pro, 90):.2f}, main loop {np.median(loop):.2f} / {np.percentile(loop, 90):.2f} "
              f"({kt} K tiles: {np.median(loop) / kt * 1e3:.0f} ns each), epilogue {np.median(epi):.2f} / {np.percentile(epi, 90):.2f}, "
              f"sum of workgroup lifetimes / span = {(t[:, 3] - t[:, 0]).sum() / span:.1f} resident on average", flush=True)
