"""Per-workgroup wall-clock stamps (100 MHz) of the patch kernel: entry, item setup done, first data landed, main loop end,
after the epilogue.  Needs a build of conv_patch.hip with the stamp hooks (experiment)."""
import sys, ctypes, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from embeddingnet_amd import _lib, layers as L
from test_conv_patch_gpu import planes_of
dev = torch.device('cuda', 0); lib = _lib.lib()
N = 128
for (h, c, k) in [(56, 64, 64), (28, 128, 128), (14, 256, 256), (7, 512, 512)]:
    x = torch.randn(N, h, h, c, device=dev); w = torch.randn(3, 3, c, k, device=dev) * 0.05
    xp = planes_of(x); wp = L.weight_planes(w, 0)
    y = torch.empty(N, h, h, k, device=dev)
    wsb = lib.embnet_conv2d_patch_workspace_bytes(N, c, 3, 3, k, h, h)
    ws = torch.empty(max(wsb, 4) // 4, device=dev)
    st = torch.zeros(256 * 32, dtype=torch.int64, device=dev)
    def run():
        _lib.check(lib.embnet_conv2d_patch_f32(xp.data_ptr(), wp.data_ptr(), None, y.data_ptr(), N, h, h, c, 3, 3, k, 1, 1, h, h, 0, None, None,
                                               ws.data_ptr(), ws.numel() * 4, _lib.stream()))
    for _ in range(20): run()
    torch.cuda.synchronize()
    lib.embnet_debug_patch_stamps(ctypes.c_void_p(st.data_ptr()))
    run(); torch.cuda.synchronize()
    lib.embnet_debug_patch_stamps(None)
    s = st.cpu().numpy().reshape(256, 32).astype(np.float64) / 100.0        # us
    t0 = s[:, 0].min()
    end = s[:, 31]
    span = end.max() - t0
    rows = []
    for b in range(256):
        items = [i for i in range(9) if s[b, 1 + 3 * i] > 0]
        if not items: continue
        setup = s[b, 1] - s[b, 0]
        wait = sum(s[b, 2 + 3 * i] - s[b, 1 + 3 * i] for i in items)
        loop = sum(s[b, 3 + 3 * i] - s[b, 2 + 3 * i] for i in items)
        epi = sum((s[b, 1 + 3 * (i + 1)] if (i + 1) in items else s[b, 31]) - s[b, 3 + 3 * i] for i in items)
        rows.append((s[b, 0] - t0, setup, wait, loop, epi, s[b, 31] - t0, len(items)))
    r = np.array(rows)
    print(f"n{N} {h}x{h}x{c}->{k}: kernel span {span:.1f} us; per workgroup (mean / max): start +{r[:,0].mean():.1f}/{r[:,0].max():.1f}, setup {r[:,1].mean():.1f}/{r[:,1].max():.1f}, "
          f"waiting for first data {r[:,2].mean():.1f}/{r[:,2].max():.1f}, main loops {r[:,3].mean():.1f}/{r[:,3].max():.1f}, epilogues(+item setup) {r[:,4].mean():.1f}/{r[:,4].max():.1f}, "
          f"end at {r[:,5].mean():.1f}/{r[:,5].max():.1f} (min {r[:,5].min():.1f}); items {r[:,6].mean():.2f}", flush=True)
