"""Oracle (test infrastructure): the reference's backbones + heads restated on
PyTorch-CPU float32 with Keras semantics.  PARITY UNPINNED (see __init__.py):
TensorFlow/Keras and the two model zoos are not installable here, and the
reference holds no golden outputs for them.

Follows /root/reference/embedding_net/backbones.py:
  'simple'  :19-41, 'simple2' :42-81, zoo branch + head :82-121,
and models.py:44 (classification head), :181-185 (3 shared-weight branches),
:217-228 (siamese distance heads).  ResNet18/50 and EfficientNet-B0 layer
stacks are the published `image-classifiers` / `efficientnet` (qubvel)
architectures that backbones.py:84-104 instantiates.

Keras semantics restated:
  * tensors NHWC; Conv2D kernel [R,S,Cin,Cout]; Dense kernel [in,out];
  * Conv2D default stride 1, 'valid', bias, glorot_uniform; 'same' with stride
    s: out=ceil(in/s), extra pad goes bottom/right;
  * MaxPool2D() = 2x2/2 floor; Flatten in (h,w,c) order;
  * BatchNormalization: momentum .99, eps 1e-3 (zoo ResNet: 2e-5), training
    uses biased batch variance, moving <- .99 moving + .01 batch;
  * K.l2_normalize(x,1) = x * rsqrt(max(sum x^2, 1e-12)).
Parameter names are the contract shared with embeddingnet_amd
so tests can load identical weights into both sides.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


class Ctx:
    """Holds parameters (created on first use with the Keras default
    initialisers when absent), BN moving statistics, and the mode."""

    def __init__(self, params=None, training=False, seed=0, dropout=False):
        self.params = {} if params is None else params
        self.training = training
        self.dropout = dropout
        self.rs = np.random.RandomState(seed)
        self.new_stats = {}
        self.reg = []          # (lambda, tensor) kernel regularisers hit in this pass
        self.taps = None       # set to a dict to record named intermediate activations (tests)

    def tap(self, name, x):
        if self.taps is not None:
            self.taps[name] = x
        return x

    def get(self, name, shape, init):
        if name not in self.params:
            self.params[name] = torch.from_numpy(init(self.rs, shape).astype(np.float32))
        t = self.params[name]
        assert tuple(t.shape) == tuple(shape), (name, tuple(t.shape), shape)
        return t


def _fans(shape):
    if len(shape) == 2:
        return shape[0], shape[1]
    rf = int(np.prod(shape[:-2]))
    return shape[-2] * rf, shape[-1] * rf


def glorot_uniform(rs, shape):
    fi, fo = _fans(shape)
    lim = math.sqrt(6.0 / (fi + fo))
    return rs.uniform(-lim, lim, size=shape)


def he_uniform(rs, shape):
    fi, _ = _fans(shape)
    lim = math.sqrt(6.0 / fi)
    return rs.uniform(-lim, lim, size=shape)


def zeros(rs, shape):
    return np.zeros(shape)


def ones(rs, shape):
    return np.ones(shape)


def same_pad(n, k, s):
    out = -(-n // s)
    total = max((out - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def conv2d(ctx, name, x, cout, k, stride=1, padding="valid", bias=True, init=glorot_uniform,
           l2=0.0, relu=False):
    """x NHWC.  padding: 'valid' | 'same' | int (symmetric ZeroPadding2D before a valid conv)."""
    n, h, w, cin = x.shape
    kern = ctx.get(name + "/kernel", (k, k, cin, cout), init)
    if l2:
        ctx.reg.append((l2, kern))
    if padding == "same":
        pt, pb = same_pad(h, k, stride)
        pl, pr = same_pad(w, k, stride)
    elif padding == "valid":
        pt = pb = pl = pr = 0
    else:
        pt = pb = pl = pr = int(padding)
    xc = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    b = ctx.get(name + "/bias", (cout,), zeros) if bias else None
    y = F.conv2d(xc, kern.permute(3, 2, 0, 1), b, stride=stride).permute(0, 2, 3, 1)
    return torch.relu(y) if relu else y


def depthwise_conv2d(ctx, name, x, k, stride, init):
    n, h, w, c = x.shape
    kern = ctx.get(name + "/depthwise_kernel", (k, k, c, 1), init)
    pt, pb = same_pad(h, k, stride)
    pl, pr = same_pad(w, k, stride)
    xc = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    y = F.conv2d(xc, kern.permute(2, 3, 0, 1), None, stride=stride, groups=c)
    return y.permute(0, 2, 3, 1)


def batchnorm(ctx, name, x, eps=1e-3, momentum=0.99, scale=True):
    c = x.shape[-1]
    gamma = ctx.get(name + "/gamma", (c,), ones) if scale else None
    beta = ctx.get(name + "/beta", (c,), zeros)
    mm = ctx.get(name + "/moving_mean", (c,), zeros)
    mv = ctx.get(name + "/moving_variance", (c,), ones)
    if ctx.training:
        red = tuple(range(x.dim() - 1))
        mean = x.mean(dim=red)
        var = x.var(dim=red, unbiased=False)
        # a layer called several times in one step (the three branches of models.py:181-183) updates its moving
        # statistics once per call, each update starting from the previous one's result
        mm = ctx.new_stats.get(name + "/moving_mean", mm)
        mv = ctx.new_stats.get(name + "/moving_variance", mv)
        ctx.new_stats[name + "/moving_mean"] = (momentum * mm + (1 - momentum) * mean).detach()
        ctx.new_stats[name + "/moving_variance"] = (momentum * mv + (1 - momentum) * var).detach()
    else:
        mean, var = mm, mv
    y = (x - mean) * torch.rsqrt(var + eps)
    if scale:
        y = y * gamma
    return y + beta


def maxpool(x, k=2, s=2, zero_pad=0):
    xc = x.permute(0, 3, 1, 2)
    if zero_pad:
        xc = F.pad(xc, (zero_pad,) * 4)      # ZeroPadding2D: pads with 0, not -inf
    return F.max_pool2d(xc, k, s).permute(0, 2, 3, 1)


def dense(ctx, name, x, units, relu=False, l2=0.0, init=glorot_uniform):
    kern = ctx.get(name + "/kernel", (x.shape[-1], units), init)
    if l2:
        ctx.reg.append((l2, kern))
    y = x @ kern + ctx.get(name + "/bias", (units,), zeros)
    return torch.relu(y) if relu else y


def dropout(ctx, x, rate):
    if ctx.training and ctx.dropout:
        return F.dropout(x, rate, training=True)
    return x


def l2_normalize(x):
    return x * torch.rsqrt(torch.clamp((x * x).sum(dim=1, keepdim=True), min=1e-12))


# ---------------------------------------------------------------------------
# backbones.py:19-41
def simple(ctx, x):
    x = ctx.tap("pool1", maxpool(conv2d(ctx, "conv1", x, 64, 10, relu=True, l2=2e-4)))
    x = ctx.tap("pool2", maxpool(conv2d(ctx, "conv2", x, 128, 7, relu=True, l2=2e-4)))
    x = ctx.tap("pool3", maxpool(conv2d(ctx, "conv3", x, 128, 4, relu=True, l2=2e-4)))
    x = ctx.tap("conv4", conv2d(ctx, "conv4", x, 256, 4, relu=True, l2=2e-4))
    return x.reshape(x.shape[0], -1)


def simple_head(ctx, feat, enc, norm):
    e = dense(ctx, "dense", feat, enc, relu=True, l2=1e-3)
    return l2_normalize(e) if norm else e


# backbones.py:42-81 — note conv -> ReLU -> BN order
def simple2(ctx, x):
    def cbr(i, x, c, k, stride=1, padding="valid"):
        x = conv2d(ctx, f"conv{i}", x, c, k, stride=stride, padding=padding, relu=True, l2=2e-4)
        return ctx.tap(f"bn{i}", batchnorm(ctx, f"bn{i}", x))
    x = cbr(1, x, 32, 3)
    x = cbr(2, x, 32, 3)
    x = cbr(3, x, 32, 5, 2, "same")
    x = dropout(ctx, x, 0.4)
    x = cbr(4, x, 64, 3)
    x = cbr(5, x, 64, 3)
    x = cbr(6, x, 64, 5, 2, "same")
    x = dropout(ctx, x, 0.4)
    x = cbr(7, x, 128, 4)
    return x                                    # backbone_model output (:69-70)


def simple2_head(ctx, feat, enc, norm):
    x = feat.reshape(feat.shape[0], -1)
    x = dense(ctx, "dense1", x, 512, relu=True)
    x = dropout(ctx, x, 0.5)
    e = dense(ctx, "dense2", x, enc, relu=True, l2=1e-3)
    return l2_normalize(e) if norm else e


# image-classifiers ResNet (pre-activation) — backbones.py:99-104
RESNET = {"resnet18": ("basic", (2, 2, 2, 2)), "resnet34": ("basic", (3, 4, 6, 3)),
          "resnet50": ("bottleneck", (3, 4, 6, 3))}
RN_EPS = 2e-5


def _rn_conv(ctx, name, x, c, k, stride=1, pad=0):
    return conv2d(ctx, name, x, c, k, stride=stride, padding=pad, bias=False, init=he_uniform)


def resnet(ctx, x, name):
    kind, reps = RESNET[name]
    x = batchnorm(ctx, "bn_data", x, eps=RN_EPS, scale=False)
    x = _rn_conv(ctx, "conv0", x, 64, 7, 2, 3)
    x = ctx.tap("conv0", x)
    x = torch.relu(batchnorm(ctx, "bn0", x, eps=RN_EPS))
    x = ctx.tap("pooling0", maxpool(x, 3, 2, zero_pad=1))
    for stage, rep in enumerate(reps):
        f = 64 * 2 ** stage
        for blk in range(rep):
            pre = f"stage{stage + 1}_unit{blk + 1}_"
            stride = 2 if (blk == 0 and stage > 0) else 1
            post = blk == 0                      # projection shortcut from the activated tensor
            a = torch.relu(batchnorm(ctx, pre + "bn1", x, eps=RN_EPS))
            if kind == "basic":
                sc = _rn_conv(ctx, pre + "sc", a, f, 1, stride) if post else x
                y = _rn_conv(ctx, pre + "conv1", a, f, 3, stride, 1)
                y = torch.relu(batchnorm(ctx, pre + "bn2", y, eps=RN_EPS))
                y = _rn_conv(ctx, pre + "conv2", y, f, 3, 1, 1)
            else:
                sc = _rn_conv(ctx, pre + "sc", a, f * 4, 1, stride) if post else x
                y = _rn_conv(ctx, pre + "conv1", a, f, 1)
                y = torch.relu(batchnorm(ctx, pre + "bn2", y, eps=RN_EPS))
                y = _rn_conv(ctx, pre + "conv2", y, f, 3, stride, 1)
                y = torch.relu(batchnorm(ctx, pre + "bn3", y, eps=RN_EPS))
                y = _rn_conv(ctx, pre + "conv3", y, f * 4, 1)
            x = ctx.tap(pre[:-1], y + sc)
    return ctx.tap("bn1", torch.relu(batchnorm(ctx, "bn1", x, eps=RN_EPS)))


# efficientnet (qubvel) — backbones.py:84-98
EFN_BLOCKS = [(3, 1, 32, 16, 1, 1), (3, 2, 16, 24, 6, 2), (5, 2, 24, 40, 6, 2), (3, 3, 40, 80, 6, 2),
              (5, 3, 80, 112, 6, 1), (5, 4, 112, 192, 6, 2), (3, 1, 192, 320, 6, 1)]
EFN_SCALING = {"efficientnet-b0": (1.0, 1.0), "efficientnet-b1": (1.0, 1.1), "efficientnet-b2": (1.1, 1.2),
               "efficientnet-b3": (1.2, 1.4), "efficientnet-b4": (1.4, 1.8), "efficientnet-b5": (1.6, 2.2),
               "efficientnet-b6": (1.8, 2.6), "efficientnet-b7": (2.0, 3.1)}


def conv_normal(rs, shape):
    """VarianceScaling(scale=2, mode='fan_out', distribution='normal')"""
    _, fo = _fans(shape)
    return rs.randn(*shape) * math.sqrt(2.0 / fo)


def _efn_round_filters(f, width, divisor=8):
    f *= width
    new = max(divisor, int(f + divisor / 2) // divisor * divisor)
    if new < 0.9 * f:
        new += divisor
    return int(new)


def efficientnet(ctx, x, name):
    width, depth = EFN_SCALING[name]
    sw = lambda t: t * torch.sigmoid(t)
    cv = lambda nm, t, c, k, s: conv2d(ctx, nm, t, c, k, stride=s, padding="same", bias=False, init=conv_normal)
    x = ctx.tap("stem", sw(batchnorm(ctx, "stem_bn", cv("stem_conv", x, _efn_round_filters(32, width), 3, 2))))
    idx = 0
    for k, rep, cin, cout, e, s in EFN_BLOCKS:
        cin, cout = _efn_round_filters(cin, width), _efn_round_filters(cout, width)
        for i in range(int(math.ceil(depth * rep))):
            idx += 1
            pre = f"block{idx}_"
            bin_, stride = (cin, s) if i == 0 else (cout, 1)
            inp = x
            mid = bin_ * e
            if e != 1:
                x = sw(batchnorm(ctx, pre + "expand_bn", cv(pre + "expand_conv", x, mid, 1, 1)))
            x = sw(batchnorm(ctx, pre + "bn", depthwise_conv2d(ctx, pre + "dwconv", x, k, stride, conv_normal)))
            se = max(1, int(bin_ * 0.25))
            sq = x.mean(dim=(1, 2))
            sq = sw(dense(ctx, pre + "se_reduce", sq, se, init=conv_normal))
            sq = torch.sigmoid(dense(ctx, pre + "se_expand", sq, mid, init=conv_normal))
            x = x * sq[:, None, None, :]
            x = batchnorm(ctx, pre + "project_bn", cv(pre + "project_conv", x, cout, 1, 1))
            if stride == 1 and bin_ == cout:
                x = x + inp                      # drop-connect is off in parity runs (ctx.dropout False)
            ctx.tap(pre[:-1], x)
    return sw(batchnorm(ctx, "top_bn", cv("top_conv", x, _efn_round_filters(1280, width), 1, 1)))


# backbones.py:110-121
def zoo_head(ctx, feat, enc, norm):
    x = ctx.tap("gap", feat.mean(dim=(1, 2)))
    x = ctx.tap("dense1", dense(ctx, "dense1", x, enc // 2, relu=True))
    e = ctx.tap("dense2", dense(ctx, "dense2", x, enc, relu=True))
    return l2_normalize(e) if norm else e


def base_model(ctx, x, backbone_name="simple", encodings_len=4096, embeddings_normalization=True):
    """images NHWC float32 [0,1] -> embeddings [n, E]  (get_backbone()[0])."""
    if backbone_name == "simple":
        return simple_head(ctx, simple(ctx, x), encodings_len, embeddings_normalization)
    if backbone_name == "simple2":
        return simple2_head(ctx, simple2(ctx, x), encodings_len, embeddings_normalization)
    if backbone_name in RESNET:
        return zoo_head(ctx, resnet(ctx, x, backbone_name), encodings_len, embeddings_normalization)
    if backbone_name in EFN_SCALING:
        return zoo_head(ctx, efficientnet(ctx, x, backbone_name), encodings_len, embeddings_normalization)
    raise KeyError(backbone_name)


def regularisation(ctx):
    """sum lambda * sum(w^2) over kernels touched by the last forward (Keras l2())."""
    seen, total = set(), 0.0
    for lam, w in ctx.reg:
        if id(w) not in seen:
            seen.add(id(w))
            total = total + lam * (w * w).sum()
    return total


def triplet_model(ctx, a, p, n, **kw):
    """models.py:176-186 — three calls of the shared base model, concat on last axis."""
    return torch.cat([base_model(ctx, a, **kw), base_model(ctx, p, **kw), base_model(ctx, n, **kw)], dim=-1)


def classification_head(ctx, emb):
    """models.py:44 — Dense(units=1, activation='sigmoid', name='output_img') on the embedding."""
    return torch.sigmoid(dense(ctx, "output_img", emb, 1))


def siamese_l1_output(ctx, e1, e2):
    """models.py:217-221 — sigmoid(Dense(1, name='output_siamese')(|e1 - e2|))."""
    return torch.sigmoid(dense(ctx, "output_siamese", (e1 - e2).abs(), 1))


def siamese_l2_distance(e1, e2):
    """models.py:225 — sqrt(max(sum (e1-e2)^2, K.epsilon())), keepdims."""
    return torch.sqrt(torch.clamp(((e1 - e2) ** 2).sum(dim=1, keepdim=True), min=1e-7))
