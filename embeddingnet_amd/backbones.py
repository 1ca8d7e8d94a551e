"""Drop-in for embedding_net/backbones.py: get_backbone() with the reference's
signature, names and return value, built from embeddingnet_amd.layers (HIP).

  'simple'            reference backbones.py:19-41
  'simple2'           reference backbones.py:42-81   (conv -> ReLU -> BN order)
  'resnet18/34/50'    reference backbones.py:99-104  (image-classifiers pre-activation ResNet)
  head                reference backbones.py:110-121 (GAP -> Dense(E//2) -> Dense(E) -> l2norm)

Parameter names follow oracle/backbones.py (`<layer>/<weight>`), see
keras_weights().
"""
import warnings

import numpy as np
import torch
from torch import nn

from . import layers as L
from . import ops


def default_device():
    import os
    # one process per GPU: LOCAL_RANK picks it (more ranks than GPUs only in the gloo debug mode, parallel.init_distributed)
    return torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))


class Model(nn.Module):
    """Small Keras-Model facade: callable on NHWC tensors, .predict() on NumPy batches."""

    def __init__(self, net, name="model"):
        super().__init__()
        self.net = net
        self.name = name

    def forward(self, x):
        return self.net(x)

    @property
    def layers(self):
        return [m for m in self.net.modules() if not list(m.children())]

    @torch.no_grad()
    def predict(self, images, batch_size=32):
        """NumPy NHWC in [0,1] -> NumPy; inference mode (moving BN statistics, no dropout),
        as Keras Model.predict (used by the mining generator, datagenerators.py:214)."""
        was = self.training
        self.eval()
        dev = next(self.parameters()).device
        out = []
        x = np.asarray(images, dtype=np.float32)
        for i in range(0, len(x), batch_size):
            out.append(self(torch.from_numpy(x[i:i + batch_size]).to(dev)).cpu().numpy())
        self.train(was)
        return np.concatenate(out, axis=0)

    def summary(self):
        n = sum(p.numel() for p in self.parameters())
        print(f"Model: {self.name}\n{self.net}\nTotal params: {n:,}")


class L2Norm(nn.Module):
    def forward(self, x):
        return ops.l2_normalize(x)


class Seq(nn.Module):
    """Ordered container keeping attribute names (they are the weight names)."""

    def __init__(self, **mods):
        super().__init__()
        self._order = list(mods)
        for k, m in mods.items():
            setattr(self, k, m)

    def forward(self, x):
        # a Conv2D directly followed by a training-mode BatchNormalization (simple2's conv -> ReLU -> BN blocks) hands it the
        # per-channel sums from its epilogue (bias and ReLU applied): the BN does not read the tensor for its statistics.
        # (The other direction — the BN's affine applied in the NEXT conv's loader, bn(x, defer=True) — was measured on
        # simple2: 0.878 -> 0.943 ms/step; the transform kernels cost more than the four apply launches they remove.)
        # A Dropout directly behind a BatchNormalization (bn3 -> drop1, bn6 -> drop2) is applied by the BatchNormalization's
        # own kernels (same mask, same arithmetic): two forward and two backward launches off simple2's dependency chain.
        mods = [getattr(self, k) for k in self._order]
        skip = False
        for i, m in enumerate(mods):
            if skip:
                skip = False
                continue
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            if (isinstance(m, L.Conv2D) and isinstance(nxt, L.BatchNormalization) and nxt.training and EPILOGUE_STATS
                    and torch.is_grad_enabled()):
                x = m(x, emit_stats=True)
            elif (isinstance(m, L.BatchNormalization) and isinstance(nxt, L.Dropout) and nxt.active()
                    and L.FUSE_DROPOUT_BN[0]):
                x = m(x, dropout=nxt)
                skip = True
            else:
                x = m(x)
        return x


def _three_products(seq):
    """The small backbones' convs on three products per fp32 product (layers.CONV_F16 note): every operand with a range —
      kernels: their exact maximum; the image: exact, from the pass that widens it to four channels (where that pass runs);
      activations: `simple` has no BatchNormalization — each MaxPool2D leaves the exact max of what it writes; `simple2`: the bound
        a training-mode BatchNormalization derives from its statistics (x 1 / (1 - rate) behind a fused Dropout);
      gradients: the exact max |dz| from the pass that applies the fused ReLU's mask (MaxPool2D / BatchNormalization backward,
        embnet_relu_bwd_colsum_ex for the last conv).
    A pass without both ranges (c or k % 4 != 0, an inference-mode BatchNormalization) runs the six-term kernels."""
    for m in seq.modules():
        if isinstance(m, L.Conv2D):
            m.f16 = True
        elif isinstance(m, L.MaxPool2D):
            m.emit_range = True
    return seq


def _simple(gen):
    return _three_products(Seq(
        conv1=L.Conv2D(3, 64, 10, activation="relu", l2=2e-4, gen=gen), pool1=L.MaxPool2D(),
        conv2=L.Conv2D(64, 128, 7, activation="relu", l2=2e-4, gen=gen), pool2=L.MaxPool2D(),
        conv3=L.Conv2D(128, 128, 4, activation="relu", l2=2e-4, gen=gen), pool3=L.MaxPool2D(),
        conv4=L.Conv2D(128, 256, 4, activation="relu", l2=2e-4, gen=gen), flatten=L.Flatten()))


def _simple2(gen):
    def cbr(i, cin, c, k, s=1, pad="valid"):
        return {f"conv{i}": L.Conv2D(cin, c, k, strides=s, padding=pad, activation="relu", l2=2e-4, gen=gen),
                f"bn{i}": L.BatchNormalization(c)}
    mods = {}
    mods.update(cbr(1, 3, 32, 3)); mods.update(cbr(2, 32, 32, 3)); mods.update(cbr(3, 32, 32, 5, 2, "same"))
    mods["drop1"] = L.Dropout(0.4, seed=1)
    mods.update(cbr(4, 32, 64, 3)); mods.update(cbr(5, 64, 64, 3)); mods.update(cbr(6, 64, 64, 5, 2, "same"))
    mods["drop2"] = L.Dropout(0.4, seed=2)
    mods.update(cbr(7, 64, 128, 4))
    return _three_products(Seq(**mods))


RN_EPS = 2e-5
DEFER_BN = __import__("os").environ.get("EMBNET_DEFER_BN", "0") == "1"
EPILOGUE_STATS = __import__("os").environ.get("EMBNET_EPILOGUE_STATS", "1") == "1"
RESNET = {"resnet18": ("basic", (2, 2, 2, 2)), "resnet34": ("basic", (3, 4, 6, 3)),
          "resnet50": ("bottleneck", (3, 4, 6, 3))}


def _rn_conv(cin, c, k, stride, pad, gen):
    conv = L.Conv2D(cin, c, k, strides=stride, padding=pad if pad else "valid", use_bias=False,
                    kernel_initializer="he_uniform", gen=gen)
    # every conv of the zoo ResNets sits between BatchNormalizations: where it does not run the patch kernel (stem, stride-2 3x3,
    # 1x1) its gather kernels multiply on three products per fp32 product, the ranges coming from its neighbours (layers.CONV_F16)
    conv.f16 = True
    return conv


class ResidualUnit(nn.Module):
    """image-classifiers residual_conv_block / residual_bottleneck_block (pre-activation)."""

    def __init__(self, cin, filters, stride, post, kind, gen):
        super().__init__()
        self.kind, self.post = kind, post
        cout = filters if kind == "basic" else filters * 4
        self.bn1 = L.BatchNormalization(cin, epsilon=RN_EPS, relu=True)
        if post:
            self.sc = _rn_conv(cin, cout, 1, stride, 0, gen)
        if kind == "basic":
            self.conv1 = _rn_conv(cin, filters, 3, stride, 1, gen)
            self.bn2 = L.BatchNormalization(filters, epsilon=RN_EPS, relu=True)
            self.conv2 = _rn_conv(filters, filters, 3, 1, 1, gen)
        else:
            self.conv1 = _rn_conv(cin, filters, 1, 1, 0, gen)
            self.bn2 = L.BatchNormalization(filters, epsilon=RN_EPS, relu=True)
            self.conv2 = _rn_conv(filters, filters, 3, stride, 1, gen)
            self.bn3 = L.BatchNormalization(filters, epsilon=RN_EPS, relu=True)
            self.conv3 = _rn_conv(filters, cout, 1, 1, 0, gen)
            # layers.CONV1X1_PLANES (off by default): the 1x1 convs whose FORWARD on the planes GEMM gains more, back to back
            # (profiles/r06_exp_conv1x1_planes.txt), than writing their input as planes beside the fp32 copy costs — conv3 from 128
            # filters (it reads the unit's thin tensor; the 56x56 layers sit on their HBM floor either way), conv1 of the identity units
            # from 1 024 input channels (14x14 / 7x7: 150 -> 81 us against 40 us of planes written)
            self.conv3.planes1x1 = filters >= 128
            self.conv1.planes1x1 = (not post) and cin >= 1024
        self.out_channels = cout

    def forward(self, x):
        # DEFER_BN (off): each conv applies the affine + ReLU of the BN in front of it while gathering, so the
        # normalised tensors are never written.  Bit-identical, saves the BN-apply pass (0.31 ms of a 15.8 ms
        # ResNet18 step) but the extra VALU work in the conv / wgrad loaders costs 0.65 ms — measured, kept as a knob.
        # emit_stats: every conv here feeds a BatchNormalization (bn2/bn3, the next unit's bn1, the net's last bn1):
        # its epilogue also produces that BN's per-channel sums, so training reads each activation once less.
        # planes_for: a BatchNormalization in front of a 3x3 stride-1 conv writes its output also as bf16 planes, which
        # that conv (forward and data gradient) reads through the patch kernel (csrc/conv_patch.hip).
        d, st = DEFER_BN, self.training and EPILOGUE_STATS
        if self.post:                                    # projection shortcut: conv1 and sc read the same tensor
            y1, sc = L.conv_pair(self.bn1(x, defer=d, planes_for=self.conv1), self.conv1, self.sc, emit_stats=st)
        else:                                            # identity shortcut: its gradient joins dx inside bn1's backward
            # (sole: conv1 is a's only reader -> a may exist as planes only)
            a, sc = self.bn1(x, defer=d, with_skip=True, planes_for=self.conv1, sole=self.kind == "basic")
            y1 = self.conv1(a, emit_stats=st)
        # bn2 is y1's only reader and conv2 its output's: both tensors may live as planes only (layers.PLANES_ONLY)
        y = self.bn2(y1, defer=d, planes_for=self.conv2, sole=True, owns_input=True)
        if self.kind == "basic":
            return self.conv2(y, residual=sc, emit_stats=st)   # the unit's Add runs in the last conv's epilogue
        # bottleneck: bn3 is conv2's only reader (owns_input: its dx may be planes only when conv2 is a stride-1 patch conv)
        return self.conv3(self.bn3(self.conv2(y, emit_stats=st), defer=d, owns_input=True, planes_for=self.conv3), residual=sc, emit_stats=st)


class ResNet(nn.Module):
    def __init__(self, name, gen):
        super().__init__()
        kind, reps = RESNET[name]
        self.bn_data = L.BatchNormalization(3, epsilon=RN_EPS, scale=False)
        self.conv0 = _rn_conv(3, 64, 7, 2, 3, gen)
        self.bn0 = L.BatchNormalization(64, epsilon=RN_EPS, relu=True)
        self.pooling0 = L.MaxPool2D(3, 2, zero_pad=1)
        self._units = []
        cin = 64
        for stage, rep in enumerate(reps):
            f = 64 * 2 ** stage
            for blk in range(rep):
                u = ResidualUnit(cin, f, 2 if (blk == 0 and stage > 0) else 1, blk == 0, kind, gen)
                nm = f"stage{stage + 1}_unit{blk + 1}"
                setattr(self, nm, u)
                self._units.append(nm)
                cin = u.out_channels
        self.bn1 = L.BatchNormalization(cin, epsilon=RN_EPS, relu=True)
        self.out_channels = cin

    def forward(self, x):
        # conv0's output feeds bn0 only: in training mode (batch statistics) bn0's data gradient sums to zero per channel
        x = L.bn_act_maxpool(L.input_bn_conv(x, self.bn_data, self.conv0, emit_stats=self.training and EPILOGUE_STATS,
                                             zero_sum_dy=self.bn0.training), self.bn0, self.pooling0)
        for nm in self._units:
            x = getattr(self, nm)(x)
        return self.bn1(x)


def _spatial_out(net, input_shape, device):
    """Feature-map shape by a dry geometry walk (no kernel launch): run on the meta device is not
    possible for HIP ops, so compute from layer geometry."""
    h, w, c = input_shape
    for m in net.modules():
        if isinstance(m, L.Conv2D):
            _, _, _, h, w = m.geometry(h, w)
            c = m.kernel.shape[-1]
        elif isinstance(m, L.MaxPool2D):
            h, w = (h + 2 * m.p - m.k) // m.s + 1, (w + 2 * m.p - m.k) // m.s + 1
    return h, w, c


class BaseModel(nn.Module):
    """images -> embeddings: backbone + head (+ l2_norm)."""

    def __init__(self, backbone, head):
        super().__init__()
        self.backbone, self.head = backbone, head

    def forward(self, x):
        return self.head(self.backbone(x))


def get_backbone(input_shape,
                 encodings_len=4096,
                 backbone_name='simple',
                 embeddings_normalization=True,
                 backbone_weights='imagenet',
                 freeze_backbone=False,
                 **kwargs):
    """Same contract as the reference: returns (base_model, backbone_model); extra MODEL keys
    (mode, distance_type, ...) are swallowed.  Build-side extras: seed=<int>, device=<torch.device>."""
    seed = int(kwargs.get("seed", 0))
    device = kwargs.get("device") or default_device()
    gen = torch.Generator().manual_seed(seed)
    input_shape = tuple(input_shape)
    if backbone_name == 'simple':
        backbone = _simple(gen)
        h, w, c = _spatial_out(backbone, input_shape, device)
        head = Seq(dense=L.Dense(h * w * c, encodings_len, activation="relu", l2=1e-3, gen=gen))
    elif backbone_name == 'simple2':
        backbone = _simple2(gen)
        h, w, c = _spatial_out(backbone, input_shape, device)
        head = Seq(flatten=L.Flatten(), dense1=L.Dense(h * w * c, 512, activation="relu", gen=gen),
                   drop=L.Dropout(0.5, seed=3),
                   dense2=L.Dense(512, encodings_len, activation="relu", l2=1e-3, gen=gen))
    else:
        if backbone_name.startswith('efficientnet'):
            from .efficientnet import EfficientNet
            backbone = EfficientNet(backbone_name, gen)
        elif backbone_name in RESNET:
            backbone = ResNet(backbone_name, gen)
        else:
            raise KeyError(f"backbone '{backbone_name}' is not implemented in embeddingnet_amd "
                           f"(available: simple, simple2, {', '.join(RESNET)}, efficientnet-b0)")
        if backbone_weights is not None:
            if isinstance(backbone_weights, str) and backbone_weights.endswith(".npz"):
                load_keras_weights(backbone, np.load(backbone_weights), strict=False)
            else:
                warnings.warn(f"backbone_weights='{backbone_weights}': pretrained weight sets are not bundled "
                              "(no network); the backbone is randomly initialised")
        if freeze_backbone:
            # reference :106-108 sets trainable=False on backbone_model.layers[:-2]; the last two layers (final BN +
            # its activation) stay trainable.  A frozen Keras BatchNormalization also runs in inference mode.
            last_bn = getattr(backbone, "bn1", None) or getattr(backbone, "top_bn", None)
            tail = {id(p) for p in last_bn.parameters()} if last_bn is not None else set()
            for m in backbone.modules():
                if isinstance(m, L.BatchNormalization) and m is not last_bn:
                    m.freeze()
            for p in backbone.parameters():
                if id(p) not in tail:
                    p.requires_grad_(False)
        head = Seq(gap=L.GlobalAveragePooling2D(),
                   dense1=L.Dense(backbone.out_channels, encodings_len // 2, activation="relu", gen=gen),
                   dense2=L.Dense(encodings_len // 2, encodings_len, activation="relu", gen=gen))
    if embeddings_normalization:
        head.l2_norm = L2Norm()
        head._order.append("l2_norm")
    backbone_model = Model(backbone, name="backbone_model").to(device)
    base_model = Model(BaseModel(backbone, head), name="base_model").to(device)
    base_model.frozen_backbone = bool(freeze_backbone)
    return base_model, backbone_model


# ---------------------------------------------------------------------------- weight naming
def keras_weights(module):
    """{'<layer>/<weight>': tensor} with the layer path joined by '_' and container prefixes
    (net/backbone/head) dropped — the naming oracle/backbones.py uses."""
    out = {}
    for k, v in list(module.named_parameters()) + list(module.named_buffers()):
        parts = [p for p in k.split(".") if p not in ("net", "backbone", "head", "base_model", "classification_model")]
        out["_".join(parts[:-1]) + "/" + parts[-1]] = v
    return out


@torch.no_grad()
def load_keras_weights(module, weights, strict=True):
    mine = keras_weights(module)
    for k, t in mine.items():
        if k in weights:
            t.copy_(torch.as_tensor(np.asarray(weights[k]), dtype=t.dtype).reshape(t.shape))
        elif strict:
            raise KeyError(f"missing weight {k}")
    L.WEIGHT_EPOCH[0] += 1               # cached bf16 planes of conv kernels are stale


def pretrain_backbone_softmax(backbone_model, data_loader, params_softmax, params_save_paths, max_epochs=None,
                              distributed=False):
    """Optional softmax pre-training of the backbone (reference backbones.py:128-204, run by
    tools/train.py:164-170 when the config has SOFTMAX_PRETRAINING): GAP -> Dense(n_classes) trained with
    categorical cross-entropy on SimpleDataGenerator batches; LR lr0*decay^floor(epoch/step),
    ReduceLROnPlateau(0.1, patience 20), EarlyStopping(patience 10, restore best), best-only checkpoints
    under <work_dir>/<project>/pretraining_model/weights/.  Returns the history dict.
    distributed=True (a torch.distributed world is up): every rank trains on its own batches with the gradients averaged
    over the ranks (parallel.GradReducer), the monitored values are all-reduced so every rank takes the same schedule /
    stopping decisions, and rank 0 writes the checkpoints."""
    import os
    from .datagenerators import SimpleDataGenerator
    p = params_softmax
    dev = next(backbone_model.parameters()).device
    gen = torch.Generator().manual_seed(0)
    head = Seq(gap=L.GlobalAveragePooling2D(), dense=L.Dense(backbone_model.net.out_channels
                                                             if hasattr(backbone_model.net, "out_channels")
                                                             else _spatial_out(backbone_model.net, tuple(p['input_shape']), dev)[2],
                                                             data_loader.n_classes, gen=gen)).to(dev)
    kw = dict(input_shape=p['input_shape'], batch_size=p['batch_size'], n_batches=p['steps_per_epoch'],
              augmentations=p.get('augmentations'))
    train_gen = SimpleDataGenerator(data_loader.train_data, data_loader.class_names, **kw)
    val_gen = SimpleDataGenerator(data_loader.val_data, data_loader.class_names, **kw) if data_loader.validate else None
    params = [q for q in list(backbone_model.parameters()) + list(head.parameters()) if q.requires_grad]
    opt = p['optimizer'].build(params)
    reducer, rank0, mean_over_ranks = None, True, float
    if distributed:
        import torch.distributed as dist
        from .parallel import GradReducer, all_reduce_mean, broadcast_model
        broadcast_model(head)                          # the backbone was broadcast by the caller; the head is built here
        reducer, rank0, mean_over_ranks = GradReducer(params), dist.get_rank() == 0, all_reduce_mean
    wdir = os.path.join(params_save_paths['work_dir'], params_save_paths['project_name'], 'pretraining_model/weights/')
    os.makedirs(wdir, exist_ok=True)
    best, best_state, since_best, since_reduce, scale = float('inf'), None, 0, 0, 1.0
    history = {'loss': [], 'accuracy': [], 'val_loss': [], 'val_accuracy': []}
    n_epochs = min(p['n_epochs'], max_epochs or p['n_epochs'])
    for epoch in range(n_epochs):
        for g in opt.param_groups:
            g['lr'] = p['learning_rate'] * p['decay_factor'] ** (epoch // p['step_size']) * scale
        backbone_model.train(); head.train()
        ls, ac = [], []
        for _ in range(len(train_gen)):
            (x,), t = train_gen[0]
            opt.zero_grad(set_to_none=True) if reducer is None else reducer.zero()
            loss, acc, _ = ops.softmax_cross_entropy(head(backbone_model(torch.from_numpy(x).to(dev))),
                                                     torch.from_numpy(t).to(dev))
            loss.backward()
            if reducer is not None:
                reducer.finish()
            opt.step()
            ls.append(loss.detach()); ac.append(acc)
        history['loss'].append(mean_over_ranks(float(torch.stack(ls).mean())))
        history['accuracy'].append(mean_over_ranks(float(torch.stack(ac).mean())))
        monitor = history['loss'][-1]
        if val_gen is not None:
            backbone_model.eval(); head.eval()
            ls, ac = [], []
            with torch.no_grad():
                for _ in range(min(p.get('val_steps', len(val_gen)), len(val_gen))):
                    (x,), t = val_gen[0]
                    loss, acc, _ = ops.softmax_cross_entropy(head(backbone_model(torch.from_numpy(x).to(dev))),
                                                             torch.from_numpy(t).to(dev))
                    ls.append(loss); ac.append(acc)
            history['val_loss'].append(mean_over_ranks(float(torch.stack(ls).mean())))
            history['val_accuracy'].append(mean_over_ranks(float(torch.stack(ac).mean())))
            monitor = history['val_loss'][-1]
        if rank0:
            print(f"softmax pre-training epoch {epoch + 1}/{n_epochs}: " +
                  " - ".join(f"{k} {v[-1]:.4f}" for k, v in history.items() if v), flush=True)
        if monitor < best:
            best, since_best, since_reduce = monitor, 0, 0
            best_state = {k: v.detach().clone() for k, v in backbone_model.state_dict().items()}
            if rank0:
                np.savez(os.path.join(wdir, f"{params_save_paths['project_name']}_{epoch + 1:03d}.npz"),
                         **{k: v.detach().cpu().numpy() for k, v in keras_weights(backbone_model).items()})
        else:
            since_best += 1; since_reduce += 1
            if since_reduce >= 20:
                scale *= 0.1; since_reduce = 0
            if since_best >= 10:
                if rank0:
                    print('EarlyStopping (restoring best weights)')
                break
    if best_state is not None:
        backbone_model.load_state_dict(best_state)
    if reducer is not None:
        reducer.close()                                # the main training builds its own reducer / gradient buffers
    return history
