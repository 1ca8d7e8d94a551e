#!/usr/bin/env python3
"""Experiment (GPU box, -DEMBNET_PLANES_STAMPS=1 build as EMBNET_LIB): cycles a wave of conv_patch_kernel spends per step in
(a) the vmcnt(0) + barrier at the step's top, (b) issuing the next step's LDS-DMA, (c) fragment reads + MFMAs, and per tile
in the epilogue.  Shares, not lengths (the stamps fence the schedule)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from embeddingnet_amd import _lib
SHAPES = [(128, 56, 56, 64, 3, 64, 1, 1), (128, 28, 28, 128, 3, 128, 1, 1), (128, 14, 14, 256, 3, 256, 1, 1)]
dev = torch.device("cuda:0")
l = _lib.lib()
vp = ctypes.c_void_p
l.embnet_split_planes_f32.argtypes = [vp, ctypes.c_long, vp, vp]
l.embnet_prep_weight_planes.argtypes = [vp] + [ctypes.c_int] * 5 + [vp, vp]
l.embnet_conv2d_patch_planes.argtypes = [vp, vp, vp, vp] + [ctypes.c_int] * 12 + [vp, vp, vp, ctypes.c_size_t, vp]
l.embnet_conv2d_patch_workspace_bytes.restype = ctypes.c_size_t
l.embnet_debug_set_planes_stamps.argtypes = [vp]
l.embnet_split_planes_cm_f32.argtypes = [vp, ctypes.c_long, ctypes.c_int, vp, vp]
l.embnet_prep_weight_planes2.argtypes = [vp] + [ctypes.c_int] * 5 + [vp, vp]
l.embnet_conv2d_patch2_planes.argtypes = [vp, vp, vp, vp] + [ctypes.c_int] * 12 + [vp, vp, vp, ctypes.c_size_t, vp]
l.embnet_conv2d_patch2_workspace_bytes.restype = ctypes.c_size_t
V2 = os.environ.get("PATCH_V", "1") == "2"
st = torch.cuda.current_stream().cuda_stream
P = lambda t: t.data_ptr()
for (n, h, w, c, ks, k, s_, pad) in SHAPES:
    oh, ow = h, w
    x = torch.randn((n, h, w, c), device=dev).abs_()
    wt = torch.randn((ks, ks, c, k), device=dev) * 0.05
    y = torch.empty((n, oh, ow, k), device=dev)
    xp = torch.empty((3, x.numel()), device=dev, dtype=torch.int16)
    l.embnet_split_planes_f32(P(x), x.numel(), P(xp), st)
    wp = torch.empty((3, wt.numel()), device=dev, dtype=torch.int16)
    l.embnet_prep_weight_planes(P(wt), ks, ks, c, k, 0, P(wp), st)
    pws = torch.empty(max(l.embnet_conv2d_patch_workspace_bytes(n, c, ks, ks, k, s_, oh, ow), 1024) // 4, device=dev)
    if V2:
        xp = torch.empty((3, x.numel()), device=dev, dtype=torch.int16)
        l.embnet_split_planes_cm_f32(P(x), n * h * w, c, P(xp), st)
        l.embnet_prep_weight_planes2(P(wt), ks, ks, c, k, 0, P(wp), st)
        pws = torch.empty(max(l.embnet_conv2d_patch2_workspace_bytes(n, c, ks, ks, k, oh, ow), 1024) // 4, device=dev)
    run = lambda: (l.embnet_conv2d_patch2_planes if V2 else l.embnet_conv2d_patch_planes)(P(xp), P(wp), None, P(y), n, h, w, c, ks, ks, k, pad, pad, oh, ow, 0, None, None, P(pws), pws.numel() * 4, st)
    for _ in range(20):
        run()
    buf = torch.zeros((256 * 8, 8), device=dev, dtype=torch.int64)
    l.embnet_debug_set_planes_stamps(P(buf))
    run()
    torch.cuda.synchronize()
    l.embnet_debug_set_planes_stamps(None)
    t = buf.cpu().numpy().astype(np.float64)
    t = t[t[:, 4] > 0]
    steps = t[:, 4]
    print(f"n{n} {h}x{w}x{c} k{ks} -> {k}: per wave, cycles per step (mean over {len(t)} waves): wait+barrier {np.mean(t[:,0]/steps):.0f}, "
          f"DMA issue {np.mean(t[:,1]/steps):.0f}, reads+MFMA {np.mean(t[:,2]/steps):.0f}; epilogue+setup per step {np.mean(t[:,3]/steps):.0f}; "
          f"steps per wave {steps.mean():.1f}; kernel cycles {t[:,5].mean():.0f}", flush=True)
