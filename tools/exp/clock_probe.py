"""rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -- python3 tools/exp/clock_probe.py
Runs the vendor fp32 GEMM and this repo's distance GEMM back to back on the same sizes; the counters give the
shader clock (GRBM_GUI_ACTIVE / 8 XCDs / duration) and the MFMA-busy fraction of each."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from embeddingnet_amd import _lib
from embeddingnet_amd._lib import check, ptr, stream
dev = torch.device("cuda:0")
n, e = 16384, 4096
x = torch.rand((n, e), device=dev)
x = x / x.norm(dim=1, keepdim=True)
d = torch.empty((n, n), device=dev)
ws = torch.empty(n, device=dev)
lib = _lib.lib()
for _ in range(6):
    torch.matmul(x, x.t())
torch.cuda.synchronize()
for _ in range(6):
    check(lib.embnet_pairwise_dist_f32(ptr(x), n, e, ptr(d), 0, ptr(ws), ws.numel() * 4, stream()))
torch.cuda.synchronize()
