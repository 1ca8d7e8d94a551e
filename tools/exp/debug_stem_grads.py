"""Stem kernel on / off inside ResNet18: per-parameter gradient differences (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from embeddingnet_amd import layers as L
from embeddingnet_amd import backbones as B

dev = torch.device("cuda:0")
size, batch = int(sys.argv[1]), int(sys.argv[2])
res = {}
for on in (True, False, "six"):
    L.STEM_CONV[0] = on is True
    L.CONV_F16[0] = on != "six"
    base, _ = B.get_backbone((size, size, 3), encodings_len=64, backbone_name="resnet18", backbone_weights=None, seed=1, device=dev)
    base.train()
    rs = np.random.RandomState(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
    x = torch.tensor(rs.rand(batch, size, size, 3).astype(np.float32)).to(dev)
    wgt = torch.tensor(rs.randn(batch, 64).astype(np.float32)).to(dev)
    emb = base(x)
    (emb * wgt).sum().backward()
    res[on] = {k: v.grad.detach().clone() for k, v in B.keras_weights(base).items() if v.grad is not None}
L.CONV_F16[0] = True
for k in list(res[False]):
    a, b, c = res[True][k], res[False][k], res["six"][k]
    sc = float(c.abs().max())
    print("%-34s max %.3e   stem-on vs six %.2e   stem-off vs six %.2e   on vs off %.2e" % (k, sc, float((a - c).abs().max()) / sc, float((b - c).abs().max()) / sc, float((a - b).abs().max()) / sc))
