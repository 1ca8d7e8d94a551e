"""Time the depthwise convolution kernels (forward, data gradient, weight gradient) on the EfficientNet-B0 layers at batch
256 through the library's kernel trace.  EMBNET_DW_ROWS2=0/1 (read once per process): one / two output rows per thread."""
import sys, torch
sys.path.insert(0, '.')
from embeddingnet_amd import _lib
from embeddingnet_amd.layers import workspace, same_pad
dev = torch.device('cuda', 0); lib = _lib.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
LAYERS = [(112, 32, 3, 1), (112, 96, 3, 2), (56, 144, 3, 1), (56, 144, 5, 2), (28, 240, 5, 1), (28, 240, 3, 2), (14, 480, 3, 1),
          (14, 480, 5, 1), (14, 672, 5, 1), (14, 672, 5, 2), (7, 1152, 5, 1), (7, 1152, 3, 1)]
tot = {}
for h, c, k, st in LAYERS:
    oh, pt = same_pad(h, k, st)
    x = torch.randn(N, h, h, c, device=dev); w = torch.randn(k, k, c, device=dev); y = torch.empty(N, oh, oh, c, device=dev)
    dy = torch.randn_like(y); dx = torch.empty_like(x); dw = torch.empty_like(w)
    ws = workspace(lib.embnet_dwconv2d_wgrad_workspace_bytes(N, c, k, k, oh, oh), dev)
    ops = {
        "fwd": lambda: _lib.check(lib.embnet_dwconv2d_fwd_f32(x.data_ptr(), w.data_ptr(), y.data_ptr(), N, h, h, c, k, k, st, pt, pt, oh, oh, _lib.stream())),
        "dgrad": lambda: _lib.check(lib.embnet_dwconv2d_dgrad_f32(dy.data_ptr(), w.data_ptr(), dx.data_ptr(), N, h, h, c, k, k, st, pt, pt, oh, oh, _lib.stream())),
        "wgrad": lambda: _lib.check(lib.embnet_dwconv2d_wgrad_f32(x.data_ptr(), dy.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel() * 4, N, h, h, c, k, k, st, pt, pt, oh, oh, _lib.stream())),
    }
    line = f"{h:3d}x{h:<3d} C={c:<5d} k{k} s{st}: "
    for what, fn in ops.items():
        for _ in range(3): fn()
        torch.cuda.synchronize()
        _lib.trace_reset(); _lib.trace_enable(True)
        for _ in range(10): fn()
        torch.cuda.synchronize()
        _lib.trace_enable(False)
        recs = _lib.trace_records()
        ms = sum(r[1] for r in recs) / 10
        nb = sum(r[4] for r in recs) / 10
        tot[what] = tot.get(what, 0.0) + ms
        line += f"{what} {recs[0][0].split('::')[-1][:26]:26s} {1e3 * ms:7.1f} us {nb / ms / 1e6:6.0f} GB/s | "
    print(line, flush=True)
print("totals (ms): " + "  ".join(f"{k} {v:.3f}" for k, v in tot.items()))
