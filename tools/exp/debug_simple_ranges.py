"""Per-layer comparison of the `simple` backbone on three products vs six terms (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from embeddingnet_amd import layers as L
from embeddingnet_amd.backbones import get_backbone

dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "simple"
shape = (105, 105, 3) if name == "simple" else (64, 64, 3)


def run(f16):
    L.CONV_F16[0] = f16
    torch.manual_seed(0)
    base, backbone = get_backbone(shape, encodings_len=64, backbone_name=name, backbone_weights=None, seed=3, device=dev)
    for m in base.modules():
        if hasattr(m, "enabled"):
            m.enabled = False
    base.train()
    outs = {}
    hooks = []
    for n, m in base.named_modules():
        if isinstance(m, (L.Conv2D, L.MaxPool2D, L.BatchNormalization, L.Dense)):
            def hook(mod, inp, out, n=n):
                o = out[0] if isinstance(out, tuple) else out
                outs[n] = o.detach().clone()
                if o.requires_grad:
                    o.register_hook(lambda g, n=n: outs.__setitem__("d " + n, g.detach().clone()))
            hooks.append(m.register_forward_hook(hook))
    g = torch.Generator().manual_seed(1)
    x = torch.rand((8,) + shape, generator=g).to(dev)
    y = base(x)
    t = torch.randn(y.shape, generator=g).to(dev)
    (y * t).sum().backward()
    grads = {n: p.grad.detach().clone() for n, p in base.named_parameters() if p.grad is not None}
    return outs, grads


o3, g3 = run(True)
o6, g6 = run(False)
for k in o6:
    a, b = o3[k], o6[k]
    print("%-28s max|six| %.3e   max diff %.3e   rel %.2e" % (k, float(b.abs().max()), float((a - b).abs().max()),
                                                           float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)))
for k in g6:
    a, b = g3[k], g6[k]
    print("grad %-23s max|six| %.3e   max diff %.3e   rel %.2e" % (k, float(b.abs().max()), float((a - b).abs().max()),
                                                                float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30)))
