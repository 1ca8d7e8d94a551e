import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from embeddingnet_amd import _lib, layers as L
from test_conv_patch_gpu import planes_of, conv64, dgrad64, GEOMS
dev = torch.device('cuda', 0)
lib = _lib.lib()
for geom in GEOMS:
    n, h, w, c, k = geom
    rng = np.random.default_rng(sum(geom))
    x = rng.standard_normal((n, h, w, c)).astype(np.float32)
    wt = (rng.standard_normal((3, 3, c, k)) / np.sqrt(9 * c)).astype(np.float32)
    xd, wd = torch.from_numpy(x).to(dev), torch.from_numpy(wt).to(dev)
    y = torch.full((n, h, w, k), float('nan'), device=dev)
    wsb = lib.embnet_conv2d_patch_workspace_bytes(n, c, 3, 3, k, h, w)
    ws = torch.empty(max(wsb, 4) // 4, device=dev)
    rows = lib.embnet_conv2d_patch_stats_rows(n, h, w)
    stats = torch.full((2, k, rows), float('nan'), device=dev)
    _lib.check(lib.embnet_conv2d_patch_f32(planes_of(xd).data_ptr(), L.weight_planes(wd, 0).data_ptr(), None, y.data_ptr(), n, h, w, c, 3, 3, k,
                                           1, 1, h, w, 0, None, stats.data_ptr(), ws.data_ptr(), ws.numel() * 4, _lib.stream()))
    want = conv64(x.astype(np.float64), wt.astype(np.float64), 1)
    got = y.cpu().numpy()
    d = np.abs(got - want) / np.abs(want).max()
    bad = np.argwhere(~(d < 3e-6))
    st = stats.cpu().numpy().astype(np.float64).sum(axis=2)
    flat = got.reshape(-1, k).astype(np.float64)
    e0 = np.abs(st[0] - flat.sum(0)).max(); e1 = np.abs(st[1] / (flat ** 2).sum(0) - 1).max()
    print(geom, 'ws', wsb, 'fwd err', np.nanmax(d), 'nan', int(np.isnan(got).sum()), 'bad', len(bad), bad[:3].tolist(), bad[-2:].tolist(), 'stats', e0, e1,
          'statnan', int(np.isnan(stats.cpu().numpy()).sum()), flush=True)
    if lib.embnet_conv2d_patch_supported(n, k, 3, 3, c, 1, h, w):
        dy = rng.standard_normal((n, h, w, k)).astype(np.float32)
        dyd = torch.from_numpy(dy).to(dev)
        dx = torch.full((n, h, w, c), float('nan'), device=dev)
        L._patch_dgrad(planes_of(dyd), wd, dx, n, h, w, c, 3, 3, k, 1, 1, h, w, None)
        want = dgrad64(dy.astype(np.float64), wt.astype(np.float64), 1)
        got = dx.cpu().numpy()
        d = np.abs(got - want) / np.abs(want).max()
        bad = np.argwhere(~(d < 3e-6))
        print('   dgrad err', np.nanmax(d), 'nan', int(np.isnan(got).sum()), 'bad', len(bad), bad[:3].tolist(), bad[-2:].tolist(), flush=True)
