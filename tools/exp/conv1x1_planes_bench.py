"""Back-to-back timing of the ResNet50 1x1 conv classes at batch 256 (one Siamese branch of C3): the 1x1 planes GEMM
(embnet_conv2d_planes1x1_f32) against the three-product gather kernel (embnet_conv2d_fwd_f32_ex with both ranges), forward, operands
cycled over several tensors (> 256 MB in all where they fit) so that neither runs out of the Infinity Cache.  Prints one row per
layer: us, fp32-equivalent TFLOP/s, and the layer's HBM floor at 5.2 TB/s (input + output (+ residual) once).
usage: python tools/exp/conv1x1_planes_bench.py [> gpurun_out/r06_exp_conv1x1_planes.txt]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from embeddingnet_amd import _lib
from embeddingnet_amd import layers as L

dev = torch.device("cuda", 0)
lib = _lib.lib()
LAYERS = [  # name, h, c, k, stride, residual
    ("s0 conv1 64->64", 56, 64, 64, 1, 0), ("s0 conv3 64->256 +res", 56, 64, 256, 1, 1), ("s0 sc 64->256", 56, 64, 256, 1, 0),
    ("s0 conv1 256->64", 56, 256, 64, 1, 0),
    ("s1 conv1 256->128", 56, 256, 128, 1, 0), ("s1 sc 256->512 /2", 56, 256, 512, 2, 0), ("s1 conv3 128->512 +res", 28, 128, 512, 1, 1),
    ("s1 conv1 512->128", 28, 512, 128, 1, 0),
    ("s2 conv1 512->256", 28, 512, 256, 1, 0), ("s2 sc 512->1024 /2", 28, 512, 1024, 2, 0), ("s2 conv3 256->1024 +res", 14, 256, 1024, 1, 1),
    ("s2 conv1 1024->256", 14, 1024, 256, 1, 0),
    ("s3 conv1 1024->512", 14, 1024, 512, 1, 0), ("s3 sc 1024->2048 /2", 14, 1024, 2048, 2, 0), ("s3 conv3 512->2048 +res", 7, 512, 2048, 1, 1),
    ("s3 conv1 2048->512", 7, 2048, 512, 1, 0),
]
N = 256


def range_of(t):
    slot = torch.zeros(1, dtype=torch.int32, device=dev)
    table = torch.tensor([[t.data_ptr(), t.numel(), slot.data_ptr()]], dtype=torch.int64, device=dev)
    ce = lib.embnet_range_chunk_elems()
    chunks = torch.tensor([(0, j) for j in range(-(-t.numel() // ce))], dtype=torch.int32, device=dev)
    _lib.check(lib.embnet_range_multi(table.data_ptr(), 1, chunks.data_ptr(), chunks.shape[0], _lib.stream()))
    return slot


def timed(fn, reps):
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


print(f"{'layer':28s} {'planes us':>10s} {'TF/s':>7s} {'fp32-DMA us':>12s} {'TF/s':>7s} {'gather us':>10s} {'TF/s':>7s} {'HBM floor us':>13s} {'dma vs gather':>14s}")
for name, h, c, k, stride, res in LAYERS:
    oh = (h - 1) // stride + 1
    m = N * oh * oh
    bytes_io = 4.0 * (N * h * h * c / (stride * stride) + m * k * (2 if res else 1))
    copies = max(2, min(6, int(400e6 / (4.0 * N * h * h * c + 4.0 * m * k)) + 1))
    g = torch.Generator(device=dev).manual_seed(h + c + k)
    xs = [torch.relu(torch.randn((N, h, h, c), device=dev, generator=g)) for _ in range(copies)]
    w = torch.randn((1, 1, c, k), device=dev, generator=g) * (2.0 / c) ** 0.5
    ys = [torch.empty((N, oh, oh, k), device=dev) for _ in range(copies)]
    rs = [torch.randn((N, oh, oh, k), device=dev, generator=g) for _ in range(copies)] if res else [None] * copies
    xps = []
    for x in xs:
        p = torch.empty(3 * x.numel(), device=dev, dtype=torch.int16)
        _lib.check(lib.embnet_planes_from_f32(x.data_ptr(), x.numel() // c, c, p.data_ptr(), _lib.stream()))
        xps.append(p)
    wp = L.weight_planes(w, 0)
    rw, rx = range_of(w), [range_of(x) for x in xs]
    ws = torch.empty(max(lib.embnet_conv2d_patch_workspace_bytes(N, c, 1, 1, k, oh, oh), lib.embnet_conv2d_fwd_workspace_bytes(N, c, 1, 1, k, oh, oh), 16) // 4,
                     device=dev)

    def planes(i):
        j = i % copies
        _lib.check(lib.embnet_conv2d_planes1x1_f32(xps[j].data_ptr(), wp.data_ptr(), None, ys[j].data_ptr(), N, h, h, c, k, stride, oh, oh, 0,
                                                   _lib.ptr(rs[j]), None, ws.data_ptr(), ws.numel() * 4, _lib.stream()))

    def gather(i):
        j = i % copies
        _lib.check(lib.embnet_conv2d_fwd_f32_ex(xs[j].data_ptr(), w.data_ptr(), None, ys[j].data_ptr(), N, h, h, c, 1, 1, k, stride, 0, 0, oh, oh, 0,
                                                _lib.ptr(rs[j]), None, None, 0, None, ws.data_ptr(), ws.numel() * 4, rx[j].data_ptr(), rw.data_ptr(),
                                                _lib.stream()))

    def dma(i):
        j = i % copies
        _lib.check(lib.embnet_conv2d_dma1x1_f32(xs[j].data_ptr(), wp.data_ptr(), None, ys[j].data_ptr(), N, h, h, c, k, stride, oh, oh, 0,
                                                _lib.ptr(rs[j]), None, rx[j].data_ptr(), ws.data_ptr(), ws.numel() * 4, _lib.stream()))

    gather(0); ref = ys[0].clone(); dma(0)
    err = float((ys[0] - ref).abs().max()) / float(ref.abs().max())
    tp, td, tg = timed(planes, 20), timed(dma, 20), timed(gather, 20)
    flop = 2.0 * m * c * k
    print(f"{name:28s} {tp:10.1f} {flop / tp / 1e6:7.1f} {td:12.1f} {flop / td / 1e6:7.1f} {tg:10.1f} {flop / tg / 1e6:7.1f} {bytes_io / 5.2e6:13.1f} {err:14.2e}")
    del xs, ys, rs, xps
    torch.cuda.empty_cache()
