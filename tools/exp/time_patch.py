"""Time embnet_conv2d_patch_f32 on the ResNet 3x3 stride-1 layers of a batch (HIP events around 20 launches)."""
import sys, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from embeddingnet_amd import _lib, layers as L
from test_conv_patch_gpu import planes_of
dev = torch.device('cuda', 0); lib = _lib.lib()
import os
ZEROS = os.environ.get('ZEROS', '0') == '1'          # all-zero operands: what the same instruction stream does when the data costs no power
N = int(sys.argv[1]) if len(sys.argv) > 1 else 128
SHAPES = [(56, 64, 64), (28, 128, 128), (14, 256, 256), (7, 512, 512)]
if len(sys.argv) > 2:
    SHAPES = [SHAPES[int(sys.argv[2])]]
for (h, c, k) in SHAPES:
    x = torch.randn(N, h, h, c, device=dev); w = torch.randn(3, 3, c, k, device=dev) * 0.05
    if ZEROS: x.zero_(); w.zero_()
    xp = planes_of(x); wp = L.weight_planes(w, 0)
    y = torch.empty(N, h, h, k, device=dev)
    wsb = lib.embnet_conv2d_patch_workspace_bytes(N, c, 3, 3, k, h, h)
    ws = torch.empty(max(wsb, 4) // 4, device=dev)
    def run():
        _lib.check(lib.embnet_conv2d_patch_f32(xp.data_ptr(), wp.data_ptr(), None, y.data_ptr(), N, h, h, c, 3, 3, k, 1, 1, h, h, 0, None, None,
                                               ws.data_ptr(), ws.numel() * 4, _lib.stream()))
    fl = 2.0 * N * h * h * k * 9 * c
    # WAVES="8 4": one-process A/B of the MFMA wave count (EMBNET_PATCH_WAVES is read at every launch), interleaved rounds
    variants = os.environ.get('WAVES', '8').split()
    best = {v: [] for v in variants}
    for rnd in range(3 if len(variants) > 1 else 1):
        for v in variants:
            os.environ['EMBNET_PATCH_WAVES'] = v
            for _ in range(10): run()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50): run()
            e1.record(); torch.cuda.synchronize()
            best[v].append(e0.elapsed_time(e1) * 1e3 / 50)
    for v in variants:
        us = sorted(best[v])[len(best[v]) // 2]
        print(f"n{N} {h}x{h}x{c}->{k} waves {v}: {us:7.1f} us  {fl / us / 1e6:6.1f} TFLOP/s  (rounds: {' '.join(f'{u:.1f}' for u in best[v])})  workspace {wsb >> 20} MiB", flush=True)
