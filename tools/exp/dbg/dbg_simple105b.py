import sys, os, numpy as np, torch
sys.path.insert(0, '/root/repo')
from embeddingnet_amd import backbones as B, layers as L, _lib
from embeddingnet_amd.backbones import keras_weights
from oracle import backbones as OB
dev = torch.device('cuda:0')
g = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
shape, enc, batch = (105, 105, 3), 64, int(os.environ.get("BATCH", "8"))
base, _ = B.get_backbone(shape, encodings_len=enc, backbone_name="simple", backbone_weights=None, seed=1, device=dev)
rs = np.random.RandomState(3)
x = rs.rand(batch, *shape).astype(np.float32); wgt = rs.randn(batch, enc).astype(np.float32)
ctx = OB.Ctx({k: v.detach().cpu().double().clone().requires_grad_(True) for k, v in keras_weights(base).items()}, training=True)
ctx.taps = {}
e = OB.base_model(ctx, torch.tensor(x, dtype=torch.float64), backbone_name="simple", encodings_len=enc)
for t in ctx.taps.values(): t.retain_grad()
(e * torch.tensor(wgt, dtype=torch.float64)).sum().backward()
got = {}
hs = []
for nm in ("pool1", "pool2", "pool3", "conv4"):
    m = getattr(base.net.backbone, nm)
    def fh(mod, i, o, nm=nm):
        o.register_hook(lambda gr, nm=nm: got.__setitem__(nm, gr.detach().clone()))
    hs.append(m.register_forward_hook(fh))
base.train(); emb = base(g(x)); (emb * g(wgt)).sum().backward()
for nm in ("conv4", "pool3", "pool2", "pool1"):
    ref = ctx.taps[nm].grad; d = got[nm].cpu().double() - ref
    print(f"d({nm} out): L2 {d.norm()/ref.norm():.2e} max {d.abs().max()/ref.abs().max():.2e}")
kw = keras_weights(base)
for k in ("conv3/kernel", "conv2/kernel", "conv2/bias", "conv1/kernel"):
    ref = ctx.params[k].grad; d = kw[k].grad.cpu().double() - ref
    print(f"{k}: L2 {d.norm()/ref.norm():.2e}")
# recompute conv2's dz by hand from d(pool2 out) with torch ops on the device tensors
acts = {}
h1 = base.net.backbone.conv2.register_forward_hook(lambda m, i, o: acts.__setitem__("c2", o.detach().clone()))
for p in base.parameters(): p.grad = None
emb = base(g(x)); (emb * g(wgt)).sum().backward()
y2 = acts["c2"]                                   # relu(conv2) on the device, [8,42,42,128]
dp2 = got["pool2"]                                # d(pool2 out) on the device
yt = y2.permute(0, 3, 1, 2).double().requires_grad_(True)
pt = torch.nn.functional.max_pool2d(yt, 2)
pt.backward(dp2.permute(0, 3, 1, 2).double())
dz = (yt.grad * (yt > 0)).permute(0, 2, 3, 1)
db = dz.sum(dim=(0, 1, 2))
ours = kw["conv2/bias"].grad.double()
print("conv2 bias grad: ours vs torch ops on our own tensors: L2", ((ours - db).norm() / db.norm()).item(), " oracle vs torch-on-ours:", ((ctx.params["conv2/bias"].grad.to(dev) - db).norm() / db.norm()).item())
# and the mask statistics
print("fraction of y2 == 0:", (y2 == 0).double().mean().item(), " windows whose max is 0:", (pt == 0).double().mean().item())
