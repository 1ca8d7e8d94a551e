import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from embeddingnet_amd import _lib, layers as L, backbones as B
dev = torch.device('cuda', 0)
base, _ = B.get_backbone(input_shape=(64, 64, 3), encodings_len=64, backbone_name="resnet18", embeddings_normalization=True, backbone_weights=None, seed=5)
base.to(dev).train()
imgs = torch.rand(16, 64, 64, 3, generator=torch.Generator().manual_seed(1)).to(dev)
bn2 = base.net.backbone.stage1_unit1.bn2
for on in (True, False):
    L.PATCH_CONV[0] = on
    keep = {}
    def pre(mod, inp):
        keep['x'] = inp[0]; keep['x0'] = inp[0].detach().clone()
    def post(mod, inp, out):
        def gh(g):
            torch.cuda.synchronize()
            d = float((keep['x'] - keep['x0']).abs().max())
            print('on', on, 'bn2 input changed between forward and backward by', d, 'dy norm', float(g.norm()))
            keep['dy'] = g.detach().clone()
        out.register_hook(gh)
    h1 = bn2.register_forward_pre_hook(pre); h2 = bn2.register_forward_hook(post)
    y = base(imgs)
    for p in base.parameters(): p.grad = None
    y.backward(torch.ones_like(y))
    torch.cuda.synchronize()
    h1.remove(); h2.remove()
    # recompute bn2's parameter gradients with torch from the tensors of this run
    x, dy = keep['x0'].double(), keep['dy'].double()
    mean = x.mean((0, 1, 2)); var = x.var((0, 1, 2), unbiased=False)
    xh = (x - mean) / torch.sqrt(var + bn2.eps)
    yv = xh * bn2.gamma.double() + bn2.beta.double()
    dz = dy * (yv > 0)
    print('   dbeta rel', float((bn2.beta.grad.double() - dz.sum((0, 1, 2))).norm() / dz.sum((0, 1, 2)).norm()),
          'dgamma rel', float((bn2.gamma.grad.double() - (dz * xh).sum((0, 1, 2))).norm() / (dz * xh).sum((0, 1, 2)).norm()))
