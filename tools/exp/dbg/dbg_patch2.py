import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from embeddingnet_amd import _lib, layers as L, backbones as B
dev = torch.device('cuda', 0)
base, _ = B.get_backbone(input_shape=(64, 64, 3), encodings_len=64, backbone_name="resnet18", embeddings_normalization=True, backbone_weights=None, seed=5)
base.to(dev).train()
imgs = torch.rand(16, 64, 64, 3, generator=torch.Generator().manual_seed(1)).to(dev)
outs = {}
for on in (True, False):
    L.PATCH_CONV[0] = on
    rec = {}
    hs = []
    for name, m in base.named_modules():
        if isinstance(m, (L.Conv2D, L.BatchNormalization, B.ResidualUnit)):
            def hook(mod, inp, out, name=name):
                o = out[0] if isinstance(out, tuple) else out
                if isinstance(o, torch.Tensor):
                    rec[name] = o.detach().clone()
                    if o.requires_grad:
                        def gh(g, name=name):
                            rec['dy:' + name] = g.detach().clone()
                        o.register_hook(gh)
            hs.append(m.register_forward_hook(hook))
    # keep BN moving stats from drifting between runs: irrelevant in train mode
    y = base(imgs)
    g = torch.ones_like(y)
    for p in base.parameters(): p.grad = None
    y.backward(g)
    rec['__out'] = y.detach().clone()
    for n_, p in base.named_parameters():
        rec['grad:' + n_] = p.grad.clone()
    for h in hs: h.remove()
    outs[on] = rec
for k in outs[True]:
    a, b = outs[True][k], outs[False][k]
    rel = float((a - b).norm() / b.norm().clamp_min(1e-30))
    if k.startswith('dy:') or 'stage1_unit1' in k or 'bn0' in k:
        print(f"{k:60s} {tuple(a.shape)} rel {rel:.3e}  maxabs diff {float((a-b).abs().max()):.3e} of {float(b.abs().max()):.3e}")
