import sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from embeddingnet_amd import backbones as B, layers as L
from embeddingnet_amd.backbones import keras_weights
from oracle import backbones as OB
dev = torch.device('cuda:0')
g = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
shape, enc, batch = (105, 105, 3), 64, 8
base, _ = B.get_backbone(shape, encodings_len=enc, backbone_name="simple", backbone_weights=None, seed=1, device=dev)
rs = np.random.RandomState(3)
x = rs.rand(batch, *shape).astype(np.float32); wgt = rs.randn(batch, enc).astype(np.float32)
base.train(); emb = base(g(x)); (emb * g(wgt)).sum().backward()
def oracle(dtype):
    ctx = OB.Ctx({k: v.detach().cpu().to(dtype).clone().requires_grad_(True) for k, v in keras_weights(base).items()}, training=True)
    e = OB.base_model(ctx, torch.tensor(x, dtype=dtype), backbone_name="simple", encodings_len=enc)
    (e * torch.tensor(wgt, dtype=dtype)).sum().backward()
    return {k: p.grad.double() for k, p in ctx.params.items() if p.grad is not None}
g64, g32 = oracle(torch.float64), oracle(torch.float32)
got = keras_weights(base)
for k, ref in g64.items():
    d = got[k].grad.cpu().double() - ref
    print(f"{k:16s} L2 {d.norm()/ref.norm():.2e} (f32 oracle {(g32[k]-ref).norm()/ref.norm():.2e})  max {d.abs().max()/ref.abs().max():.2e} (f32 {(g32[k]-ref).abs().max()/ref.abs().max():.2e})")
# single layer: wgrad of a 10x10x3->64 conv at 105x105, dy with ~half zeros
conv = L.Conv2D(3, 64, 10, use_bias=True).to(dev)
xt = g(x).requires_grad_(True)
y = conv(xt)
dy = (rs.randn(*y.shape) * (rs.rand(*y.shape) < 0.3)).astype(np.float32)
y.backward(g(dy))
xr = torch.tensor(x, dtype=torch.float64); kr = conv.kernel.detach().cpu().double().requires_grad_(True)
yr = torch.nn.functional.conv2d(xr.permute(0, 3, 1, 2), kr.permute(3, 2, 0, 1)).permute(0, 2, 3, 1)
yr.backward(torch.tensor(dy, dtype=torch.float64))
mag = torch.nn.functional.conv2d(xr.permute(0,3,1,2).abs().transpose(0,1), torch.tensor(np.abs(dy)).double().permute(0,3,1,2).transpose(0,1)).abs()
d = conv.kernel.grad.cpu().double() - kr.grad
print("single layer wgrad: L2", (d.norm()/kr.grad.norm()).item(), "max/max", (d.abs().max()/kr.grad.abs().max()).item(), " |grad| max", kr.grad.abs().max().item(), " sum|a||b| ~", mag.max().item())
x32 = torch.tensor(x); k32 = conv.kernel.detach().cpu().clone().requires_grad_(True)
y32 = torch.nn.functional.conv2d(x32.permute(0, 3, 1, 2), k32.permute(3, 2, 0, 1)).permute(0, 2, 3, 1); y32.backward(torch.tensor(dy))
d32 = k32.grad.double() - kr.grad
print("float32 cpu wgrad: L2", (d32.norm()/kr.grad.norm()).item(), "max/max", (d32.abs().max()/kr.grad.abs().max()).item())
print("---- pool2 arg-max decisions")
acts = {}
hs = [base.net.backbone.conv2.register_forward_hook(lambda m, i, o: acts.__setitem__("c2", o.detach())),
      base.net.backbone.pool2.register_forward_hook(lambda m, i, o: acts.__setitem__("p2", o.detach()))]
base(g(x))
ctx = OB.Ctx({k: v.detach().cpu().double().clone() for k, v in keras_weights(base).items()}, training=True)
import oracle.backbones as OBm
xr = torch.tensor(x, dtype=torch.float64)
c1 = OBm.maxpool(OBm.conv2d(ctx, "conv1", xr, 64, 10, relu=True, l2=2e-4))
c2 = OBm.conv2d(ctx, "conv2", c1, 128, 7, relu=True, l2=2e-4)
ours = acts["c2"].cpu().double()
print("conv2 out rel err", ((ours - c2).abs().max() / c2.abs().max()).item())
def amax(t):
    n, h, w, c = t.shape
    v = t[:, :h // 2 * 2, :w // 2 * 2].reshape(n, h // 2, 2, w // 2, 2, c).permute(0, 1, 3, 5, 2, 4).reshape(n, h // 2, w // 2, c, 4)
    return v.argmax(-1), v
a_o, v_o = amax(c2); a_g, v_g = amax(ours)
diff = (a_o != a_g)
pos = v_o.max(-1).values > 0
print("windows", diff.numel(), "differing arg-max", int(diff.sum()), "of which with a positive max", int((diff & pos).sum()))
srt = v_o.sort(-1).values
gap = (srt[..., 3] - srt[..., 2])[diff & pos]
print("gap between the two largest (oracle) where decisions differ: max", gap.max().item() if gap.numel() else None, "relative to max act", (gap.max() / c2.abs().max()).item() if gap.numel() else None)
# our kernel's own argmax vs argmax of OUR activations (is the pooling kernel consistent?)
p2 = acts["p2"].cpu().double()
print("pool2 out == max of our conv2 windows:", bool(torch.equal(p2, v_g.max(-1).values)))
