"""Stem kernel on / off inside ResNet18: the first layers' outputs side by side (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from embeddingnet_amd import layers as L
from embeddingnet_amd.backbones import get_backbone

dev = torch.device("cuda:0")
size, batch = int(sys.argv[1]), int(sys.argv[2])
outs = {}
for on in (True, False):
    L.STEM_CONV[0] = on
    torch.manual_seed(0)
    base, _ = get_backbone((size, size, 3), encodings_len=64, backbone_name="resnet18", backbone_weights=None, seed=1, device=dev)
    base.train()
    cap = {}
    orig = L._InputBNConvFn.apply
    def spy(*a, **k):
        y = orig(*a, **k)
        cap["stem"] = y.detach().clone()
        cap["stats"] = getattr(y, "_bn_partials", None)
        return y
    g = torch.Generator().manual_seed(0)
    x = torch.rand((batch, size, size, 3), generator=g).to(dev)
    import embeddingnet_amd.layers as LL
    old = LL.input_bn_conv
    def wrapped(*a, **k):
        y = old(*a, **k)
        cap["stem"] = y.detach().clone()
        p = getattr(y, "_bn_partials", None)
        cap["sums"] = None if p is None else p.detach().double().sum(2).clone()
        cap["rows"] = None if p is None else p.shape
        return y
    LL.input_bn_conv = wrapped
    import embeddingnet_amd.backbones as BB
    if hasattr(BB.L, "input_bn_conv"):
        BB.L.input_bn_conv = wrapped
    y = base(x)
    LL.input_bn_conv = old
    outs[on] = (cap, y.detach().clone())
a, b = outs[True][0], outs[False][0]
print("stem out   max|b| %.3e  max diff %.3e" % (float(b["stem"].abs().max()), float((a["stem"] - b["stem"]).abs().max())))
print("rows", a["rows"], b["rows"])
if a["sums"] is not None:
    yv = b["stem"].double().reshape(-1, 64)
    print("sum   on  vs true: %.3e   off vs true: %.3e" % (float((a["sums"][0] - yv.sum(0)).abs().max()), float((b["sums"][0] - yv.sum(0)).abs().max())))
    print("sumsq on  vs true: %.3e   off vs true: %.3e  (max %.3e)" % (float((a["sums"][1] - (yv ** 2).sum(0)).abs().max()), float((b["sums"][1] - (yv ** 2).sum(0)).abs().max()), float((yv ** 2).sum(0).max())))
print("final out  max diff %.3e" % float((outs[True][1] - outs[False][1]).abs().max()))
