"""The squeeze-and-excite gate sigmoid(Dense(swish(Dense(pooled)))) of an MBConv block (reference embedding_net/backbones.py:84-98
via efficientnet's MBConv) as one forward and two backward launches (csrc/se_mlp.hip) against float64: gate, the gradient of the
pooled input and of both layers' kernels and biases; sizes of EfficientNet-B0 / B5 blocks, ragged sample counts, one unit; the
kernel trace; and the composed form (EMBNET_SE_MLP=0) for sizes the fused launches do not take."""
import pytest
import torch

from embeddingnet_amd import _lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _ref(pooled, w1, b1, w2, b2):
    z1 = pooled @ w1 + b1
    return torch.sigmoid((z1 * torch.sigmoid(z1)) @ w2 + b2)


def _names(fn):
    _lib.trace_reset(); _lib.trace_enable(True)
    out = fn()
    torch.cuda.synchronize()
    names = [r[0] for r in _lib.trace_records()]
    _lib.trace_enable(False)
    return out, names


@pytest.mark.parametrize("n,c,s", [(256, 1152, 48), (256, 96, 4), (37, 144, 6), (3, 32, 8), (64, 480, 20), (130, 672, 28), (9, 2048, 128),
                                   (5, 16, 1), (70, 3072, 160)])
def test_fused_gate_vs_float64(dev, n, c, s):
    from embeddingnet_amd import layers as L
    gen = torch.Generator().manual_seed(n + c)
    red, exp = L.Dense(c, s, gen=gen).to(dev), L.Dense(s, c, gen=gen).to(dev)
    with torch.no_grad():
        red.bias.copy_(torch.linspace(-0.2, 0.3, s)); exp.bias.copy_(torch.linspace(-0.5, 0.5, c))
    pooled = (torch.rand(n, c, generator=gen) * 2 - 0.5).to(dev).requires_grad_(True)
    wgt = torch.cos(torch.arange(n * c, dtype=torch.float32).reshape(n, c) * 0.37).to(dev)
    assert _lib.lib().embnet_se_mlp_supported(n, c, s)
    gate, names = _names(lambda: L.se_mlp(pooled, red, exp))
    assert names == ["embnet::semlp::se_mlp_fwd_kernel"], names
    _, names = _names(lambda: (gate * wgt).sum().backward())
    assert [nm for nm in names if "semlp" in nm] == ["embnet::semlp::se_mlp_bwd_a_kernel", "embnet::semlp::se_mlp_bwd_b_kernel"], names
    assert not any("dense" in nm or "colsum" in nm or "act_" in nm for nm in names), names
    p64 = pooled.detach().cpu().double().requires_grad_(True)
    ws = [t.detach().cpu().double().requires_grad_(True) for t in (red.kernel, red.bias, exp.kernel, exp.bias)]
    want = _ref(p64, *ws)
    (want * wgt.cpu().double()).sum().backward()
    err = lambda got, ref: (got.detach().cpu().double() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)
    assert err(gate, want.detach()) < 2e-6
    assert err(pooled.grad, p64.grad) < 1e-5
    for got, ref, what in zip((red.kernel, red.bias, exp.kernel, exp.bias), ws, ("dW1", "db1", "dW2", "db2")):
        assert err(got.grad, ref.grad) < 1e-5, what


def test_composed_form_where_the_fused_launches_do_not_apply(dev):
    from embeddingnet_amd import layers as L
    gen = torch.Generator().manual_seed(2)
    n, c, s = 6, 64, 200                         # 200 units: beyond the fused launches
    assert not _lib.lib().embnet_se_mlp_supported(n, c, s)
    red, exp = L.Dense(c, s, gen=gen).to(dev), L.Dense(s, c, gen=gen).to(dev)
    pooled = torch.rand(n, c, generator=gen).to(dev)
    gate, names = _names(lambda: L.se_mlp(pooled, red, exp))
    assert any("dense_fwd" in nm for nm in names) and not any("semlp" in nm for nm in names)
    want = _ref(pooled.cpu().double(), red.kernel.detach().cpu().double(), red.bias.detach().cpu().double(),
                exp.kernel.detach().cpu().double(), exp.bias.detach().cpu().double())
    assert (gate.detach().cpu().double() - want).abs().max().item() < 2e-6


def test_fused_gate_equals_composed_gate_in_an_mbconv_block(dev):
    """One MBConv block forward + backward with the fused gate and with the composed one (EMBNET_SE_MLP switch): outputs and every
    gradient within fp32 summation order."""
    from embeddingnet_amd import efficientnet as E
    from embeddingnet_amd import layers as L
    x0 = torch.randn(6, 14, 14, 40, device=dev)
    res = {}
    for on in (True, False):
        L.SE_MLP[0] = on
        try:
            blk = E.MBConv(5, 40, 40, 6, 1, 0.0, 7, torch.Generator().manual_seed(3)).to(dev).train()
            x = x0.clone().requires_grad_(True)
            y = blk(x)
            y.backward(torch.sin(y.detach() * 1.7))
            torch.cuda.synchronize()
            res[on] = [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in blk.parameters()]
        finally:
            L.SE_MLP[0] = True
    for a, b in zip(res[True], res[False]):
        assert float((a - b).norm() / b.norm().clamp_min(1e-30)) < 2e-5
