"""EfficientNet-B0..B7 backbones (include_top=False) on the HIP layers — the `efficientnet.tfkeras`
models the reference instantiates at embedding_net/backbones.py:84-98.  The architecture is the
published qubvel/efficientnet one (third-party, not under /root/reference): stem
Conv3x3/2 -> BN -> swish; MBConv stages (expand 1x1 -> depthwise kxk -> squeeze-excite -> project
1x1, identity skip with drop-connect); top Conv1x1 -> BN -> swish.  BN eps 1e-3, momentum .99;
SE ratio 0.25 on the block's INPUT filters; drop-connect 0.2 scaled linearly over the blocks.
"""
import math

from torch import nn

from . import layers as L

# (kernel, repeats, in, out, expand, stride)
BASE_BLOCKS = [(3, 1, 32, 16, 1, 1), (3, 2, 16, 24, 6, 2), (5, 2, 24, 40, 6, 2), (3, 3, 40, 80, 6, 2),
               (5, 3, 80, 112, 6, 1), (5, 4, 112, 192, 6, 2), (3, 1, 192, 320, 6, 1)]
# width, depth coefficients
SCALING = {"efficientnet-b0": (1.0, 1.0), "efficientnet-b1": (1.0, 1.1), "efficientnet-b2": (1.1, 1.2),
           "efficientnet-b3": (1.2, 1.4), "efficientnet-b4": (1.4, 1.8), "efficientnet-b5": (1.6, 2.2),
           "efficientnet-b6": (1.8, 2.6), "efficientnet-b7": (2.0, 3.1)}
BN_EPS = 1e-3


def round_filters(f, width, divisor=8):
    f *= width
    new = max(divisor, int(f + divisor / 2) // divisor * divisor)
    if new < 0.9 * f:
        new += divisor
    return int(new)


def round_repeats(r, depth):
    return int(math.ceil(depth * r))


def block_list(name):
    width, depth = SCALING[name]
    out = []
    for k, rep, cin, cout, e, s in BASE_BLOCKS:
        cin, cout = round_filters(cin, width), round_filters(cout, width)
        for i in range(round_repeats(rep, depth)):
            out.append((k, cin if i == 0 else cout, cout, e, s if i == 0 else 1))
    return out, round_filters(32, width), round_filters(1280, width)


EFFNET_F16 = __import__("os").environ.get("EMBNET_EFFNET_F16", "1") != "0"


def _conv(cin, cout, k, stride, gen):
    conv = L.Conv2D(cin, cout, k, strides=stride, padding="same", use_bias=False, kernel_initializer="conv_normal",
                    gen=gen)
    # the 1x1 convs that run the implicit-GEMM kernels multiply on three products where the operands' ranges are known
    # (layers.CONV_F16): forward always (kernel range; activations scale 1), backward where the gradient comes out of a plain
    # BatchNormalization backward (expand convs, project convs of blocks without a skip, the top conv); the thin streams ignore it
    conv.f16 = EFFNET_F16
    return conv


class MBConv(nn.Module):
    def __init__(self, k, cin, cout, expand, stride, drop_rate, seed, gen):
        super().__init__()
        mid = cin * expand
        self.has_expand = expand != 1
        if self.has_expand:
            self.expand_conv = _conv(cin, mid, 1, 1, gen)
            self.expand_bn = L.BatchNormalization(mid, epsilon=BN_EPS, activation="swish")
        self.dwconv = L.DepthwiseConv2D(mid, k, stride, gen=gen)
        self.bn = L.BatchNormalization(mid, epsilon=BN_EPS, activation="swish")
        se = max(1, int(cin * 0.25))
        self.gap = L.GlobalAveragePooling2D()
        self.se_reduce = L.Dense(mid, se, gen=gen)           # 1x1 convs with bias on a [n,1,1,c] tensor
        self.se_expand = L.Dense(se, mid, gen=gen)
        L.conv_normal_(self.se_reduce.kernel.data.view(1, 1, mid, se), gen)
        L.conv_normal_(self.se_expand.kernel.data.view(1, 1, se, mid), gen)
        self.project_conv = _conv(mid, cout, 1, 1, gen)
        self.project_bn = L.BatchNormalization(cout, epsilon=BN_EPS)
        self.skip = stride == 1 and cin == cout
        self.drop = L.DropConnect(drop_rate, seed=seed) if self.skip and drop_rate > 0 else None
        self.out_channels = cout

    def forward(self, inp):
        x = inp
        if self.has_expand:
            # skip blocks: the block input feeds the expand conv and the final Add; with_skip folds the Add's gradient
            # into the expand conv's data-gradient epilogue
            if self.skip:
                x, inp = self.expand_conv(x, emit_stats=self.training, with_skip=True)
            else:
                x = self.expand_conv(x, emit_stats=self.training)           # BN sums from the conv epilogue
            x = self.expand_bn(x)
        # BN + swish and the squeeze-and-excite pooling of its output in one pass; in backward the pooled gradient is
        # broadcast into the scaling's gradient by one kernel (no accumulation pass)
        # BN + swish, the squeeze-and-excite pooling of its output and the gating in two passes over the depthwise output (the
        # activated tensor is never written); in backward two more (layers.BatchNormalization.se_gate)
        x = self.bn.se_gate(self.dwconv(x, emit_stats=self.training),          # (the BN's statistics from the depthwise kernel)
                            lambda g: L.se_mlp(g, self.se_reduce, self.se_expand))           # (one launch; three in backward)
        x = self.project_conv(x, emit_stats=self.training)
        if self.skip:                # BN apply + drop-connect + Add in one pass (layers.BatchNormalization.drop_add)
            return self.project_bn.drop_add(x, inp, self.drop)
        return self.project_bn(x)


class EfficientNet(nn.Module):
    def __init__(self, name, gen, drop_connect_rate=0.2):
        super().__init__()
        blocks, stem, top = block_list(name)
        self.stem_conv = _conv(3, stem, 3, 2, gen)
        self.stem_bn = L.BatchNormalization(stem, epsilon=BN_EPS, activation="swish")
        self._blocks = []
        for i, (k, cin, cout, e, s) in enumerate(blocks):
            blk = MBConv(k, cin, cout, e, s, drop_connect_rate * i / len(blocks), 100 + i, gen)
            nm = f"block{i + 1}"
            setattr(self, nm, blk)
            self._blocks.append(nm)
        self.top_conv = _conv(blocks[-1][2], top, 1, 1, gen)
        self.top_bn = L.BatchNormalization(top, epsilon=BN_EPS, activation="swish")
        self.out_channels = top

    def forward(self, x):
        x = self.stem_bn(self.stem_conv(x))
        for nm in self._blocks:
            x = getattr(self, nm)(x)
        return self.top_bn(self.top_conv(x, emit_stats=self.training))
