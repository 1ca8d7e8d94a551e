"""Back-to-back timing of the ResNet stem forward at batch 128 x 224 x 224: csrc/conv_stem.hip against the three-product gather kernel.
EMBNET_LIB=/path/to/variant.so EMBNET_LIB_LAX=1 to time a build variant."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from embeddingnet_amd import _lib

dev = torch.device("cuda:0")
lib = _lib.lib()
n, h = int(sys.argv[1]) if len(sys.argv) > 1 else 128, 224
g = torch.Generator(device=dev).manual_seed(1)
xs, ys = [], []
for i in range(3):
    x = torch.zeros((n, h, h, 4), device=dev)
    x[..., :3] = torch.randn((n, h, h, 3), device=dev, generator=g)
    xs.append(x); ys.append(torch.empty((n, 112, 112, 64), device=dev))
w = torch.zeros((7, 7, 4, 64), device=dev)
w[:, :, :3, :] = torch.randn((7, 7, 3, 64), device=dev, generator=g) / 12
slot = lambda t: t.abs().max().reshape(1).float().view(torch.int32).clone()
rx, rw = [slot(x) for x in xs], slot(w)
rows = lib.embnet_conv2d_stem_stats_rows(n, 112, 112)
st = torch.empty((2, 64, rows), device=dev)
ws = torch.empty(max(lib.embnet_conv2d_fwd_workspace_bytes(n, 4, 7, 7, 64, 112, 112) // 4, 4), device=dev)
st2 = torch.empty((2, 64, max(lib.embnet_conv2d_fwd_stats_rows(n, 4, 7, 7, 64, 112, 112), 1)), device=dev)


def stem(i):
    j = i % 3
    _lib.check(lib.embnet_conv2d_stem_f32(xs[j].data_ptr(), w.data_ptr(), ys[j].data_ptr(), n, h, h, 3, 3, 112, 112, st.data_ptr(), rx[j].data_ptr(),
                                          rw.data_ptr(), _lib.stream()))


def gather(i):
    j = i % 3
    _lib.check(lib.embnet_conv2d_fwd_f32_ex(xs[j].data_ptr(), w.data_ptr(), None, ys[j].data_ptr(), n, h, h, 4, 7, 7, 64, 2, 3, 3, 112, 112, 0, None,
                                            None, None, 0, st2.data_ptr(), ws.data_ptr(), ws.numel() * 4, rx[j].data_ptr(), rw.data_ptr(), _lib.stream()))


def timed(fn, reps=30):
    for i in range(3):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps):
        fn(i)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / reps


ts, tg = timed(stem), timed(gather)
flop = 2.0 * n * 112 * 112 * 147 * 64
print(f"stem kernel {ts:7.1f} us ({flop / ts / 1e6:6.1f} TFLOP/s on 3 channels; output {4e-6 * n * 112 * 112 * 64 / ts:5.2f} TB/s)   gather {tg:7.1f} us")

# ---- the weight gradient (needs tools/exp/conv_stem_wgrad.diff applied: the kernel is not in the library)
if not hasattr(lib, "embnet_conv2d_stem_wgrad_f32"):
    sys.exit(0)
dys = [torch.randn((n, 112, 112, 64), device=dev, generator=g) * 1e-4 for _ in range(3)]
rd = [slot(d) for d in dys]
dw = torch.empty((7, 7, 4, 64), device=dev)
wsw = torch.empty(lib.embnet_conv2d_stem_wgrad_workspace_bytes() // 4, device=dev)
wsg = torch.empty(max(lib.embnet_conv2d_wgrad_workspace_bytes(n, 4, 7, 7, 64, 112, 112) // 4, 4), device=dev)


def stem_w(i):
    j = i % 3
    _lib.check(lib.embnet_conv2d_stem_wgrad_f32(xs[j].data_ptr(), dys[j].data_ptr(), dw.data_ptr(), n, h, h, 3, 3, 112, 112, wsw.data_ptr(),
                                                wsw.numel() * 4, rx[j].data_ptr(), rd[j].data_ptr(), _lib.stream()))


def gather_w(i):
    j = i % 3
    _lib.check(lib.embnet_conv2d_wgrad_f32_ex(xs[j].data_ptr(), dys[j].data_ptr(), dw.data_ptr(), wsg.data_ptr(), wsg.numel() * 4, n, h, h, 4, 7, 7, 64,
                                              2, 3, 3, 112, 112, None, None, 0, rx[j].data_ptr(), rd[j].data_ptr(), _lib.stream()))


ts, tg = timed(stem_w), timed(gather_w)
print(f"stem weight gradient {ts:7.1f} us ({flop / ts / 1e6:6.1f} TFLOP/s; dy read at {4e-6 * n * 112 * 112 * 64 / ts:5.2f} TB/s)   gather {tg:7.1f} us")
