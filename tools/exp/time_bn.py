"""Time the BatchNorm backward kernels (embnet_bn_bwd: reduce + finalize + apply) per tensor size through the library's kernel
trace.  Shapes: the ResNet18 / EfficientNet-B0 activations at batch 128 / 256."""
import sys, torch
sys.path.insert(0, '.')
from embeddingnet_amd import _lib
from embeddingnet_amd.layers import workspace
dev = torch.device('cuda', 0); lib = _lib.lib()
SHAPES = [(128 * 56 * 56, 64), (128 * 28 * 28, 128), (128 * 14 * 14, 256), (128 * 7 * 7, 512),
          (256 * 112 * 112, 32), (256 * 56 * 56, 144), (256 * 28 * 28, 240), (256 * 14 * 14, 672), (256 * 7 * 7, 1152)]
for m, c in SHAPES:
    x = torch.randn(m, c, device=dev); dy = torch.randn(m, c, device=dev); dx = torch.empty_like(x)
    stats = torch.rand(4, c, device=dev) + 0.5
    dg = torch.empty(c, device=dev); db = torch.empty(c, device=dev)
    ws = workspace(lib.embnet_bn_workspace_bytes(m, c), dev)
    def run():
        _lib.check(lib.embnet_bn_bwd(dy.data_ptr(), x.data_ptr(), m, c, stats.data_ptr(), stats.data_ptr() + 4 * c, stats.data_ptr() + 8 * c,
                                     stats.data_ptr() + 12 * c, 1, 1, None, dx.data_ptr(), dg.data_ptr(), db.data_ptr(), None, ws.data_ptr(),
                                     ws.numel() * 4, _lib.stream()))
    for _ in range(5): run()
    torch.cuda.synchronize()
    _lib.trace_reset(); _lib.trace_enable(True)
    for _ in range(20): run()
    torch.cuda.synchronize()
    _lib.trace_enable(False)
    by = {}
    for name, ms, work, unit, nbytes in _lib.trace_records():
        e = by.setdefault(name, [0.0, 0.0, 0]); e[0] += ms; e[1] += nbytes; e[2] += 1
    print(f"m={m} c={c} ({m * c / 1e6:.1f} M elements): " + "  ".join(f"{k.split('::')[-1]} {1e3 * v[0] / v[2]:.1f} us {v[1] / v[0] / 1e6:.0f} GB/s" for k, v in by.items()), flush=True)
