"""Same GEMM shape through the engine with the B operand k-major (dense_fwd: W[in,out]) and k-contiguous
(dense_dgrad: rows of W), and both operands k-major (dense_wgrad): isolates the cost of the k-major LDS tile."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from embeddingnet_amd import _lib
from embeddingnet_amd._lib import check, ptr, stream
lib = _lib.lib()
dev = torch.device("cuda:0")
def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
for n in (4096, 8192):
    x = torch.randn((n, n), device=dev); w = torch.randn((n, n), device=dev) * 0.02
    y = torch.empty((n, n), device=dev)
    fl = 2.0 * n ** 3
    t1 = timeit(lambda: check(lib.embnet_dense_fwd_f32(ptr(x), ptr(w), None, ptr(y), n, n, n, 0, stream())))
    t2 = timeit(lambda: check(lib.embnet_dense_dgrad_f32(ptr(x), ptr(w), ptr(y), n, n, n, stream())))
    t3 = timeit(lambda: check(lib.embnet_dense_wgrad_f32(ptr(x), ptr(w), ptr(y), n, n, n, stream())))
    t4 = timeit(lambda: torch.matmul(x, w))
    print(f"n={n}: A kc x B km (dense_fwd) {fl/t1/1e12:6.1f} | A kc x B kc (dense_dgrad) {fl/t2/1e12:6.1f} | A km x B km (dense_wgrad) {fl/t3/1e12:6.1f} | rocBLAS {fl/t4/1e12:6.1f} TFLOP/s", flush=True)
