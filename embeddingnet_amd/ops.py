"""Tensor-level entry points over the C ABI (include/embnet.h).

PyTorch is plumbing here: it owns device memory, the stream and the autograd
tape; every forward/backward below is one or two launches of hand-written HIP
kernels from libembnet_hip.so.  Nothing in this module computes on the CPU or
through torch's own operators.
"""
import ctypes

import torch

from . import _lib
from ._lib import check, f32, ptr, stream

MINING_MODES = {"semihard": 0, "hardest": 1, "random_hard": 2}


def _prep(t):
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _new(shape, like, dtype=torch.float32):
    return torch.empty(shape, device=like.device, dtype=dtype)


# --------------------------------------------------------------------------- distances / mining
def pairwise_distances(x, squared=False):
    """[n,e] -> [n,n] Euclidean matrix with sklearn semantics (datagenerators.py:219)."""
    x = _prep(x.detach())
    n, e = x.shape
    lib = _lib.lib()
    ws = _new((max(lib.embnet_pairwise_workspace_bytes(n, e) // 4, 1),), x)
    d = _new((n, n), x)
    check(lib.embnet_pairwise_dist_f32(ptr(x), n, e, ptr(d), int(bool(squared)), ptr(ws),
                                       ws.numel() * 4, stream()))
    return d


def mine_triplets(dist, k_classes, k_samples, margin, mode, seed=0, with_candidates=False):
    """Online mining on a class-contiguous distance matrix (datagenerators.py:225-250).

    Returns (triplets [max_t,3] int32, count [1] int32, selected [pairs] int32[, cand_mask]).
    Only the first count[0] rows of `triplets` are live; nothing is copied to the host.
    """
    dist = _prep(dist)
    lib = _lib.lib()
    p, k = int(k_classes), int(k_samples)
    n = p * k
    if dist.shape != (n, n):
        raise _lib.EmbnetError(f"distance matrix {tuple(dist.shape)} != ({n},{n}) for {p}x{k}")
    max_t = lib.embnet_mine_max_triplets(p, k)
    trip = _new((max_t, 3), dist, torch.int32)
    count = _new((1,), dist, torch.int32)
    npairs = p * (k * (k - 1) // 2)
    sel = _new((max(npairs, 1),), dist, torch.int32)
    mask = _new((max(npairs, 1), (n - k + 31) // 32), dist, torch.int32) if with_candidates else None
    check(lib.embnet_mine_triplets(ptr(dist), p, k, f32(margin), MINING_MODES[mode],
                                   int(seed) & (2 ** 64 - 1), ptr(trip), ptr(count),
                                   ptr(sel), ptr(mask), stream()))
    return (trip, count, sel, mask) if with_candidates else (trip, count, sel)


def batch_hard(dist, k_classes, k_samples):
    """Hermans batch-hard (build-defined): [n,3] triplets, one per anchor, + count."""
    dist = _prep(dist)
    p, k = int(k_classes), int(k_samples)
    n = p * k
    trip = _new((n, 3), dist, torch.int32)
    count = _new((1,), dist, torch.int32)
    check(_lib.lib().embnet_batch_hard(ptr(dist), p, k, ptr(trip), ptr(count), stream()))
    return trip, count


# --------------------------------------------------------------------------- triplet hinge
class _TripletHinge(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y_pred, margin):
        y = _prep(y_pred)
        t, e3 = y.shape
        if e3 % 3:
            raise _lib.EmbnetError(f"triplet_loss: last dim {e3} is not 3*E")
        loss = _new((t,), y)
        check(_lib.lib().embnet_triplet_hinge_fwd(ptr(y), t, e3 // 3, f32(margin), ptr(loss), stream()))
        ctx.save_for_backward(y)
        ctx.margin = margin
        return loss

    @staticmethod
    def backward(ctx, dloss):
        (y,) = ctx.saved_tensors
        t, e3 = y.shape
        dy = torch.empty_like(y)
        check(_lib.lib().embnet_triplet_hinge_bwd(ptr(y), ptr(_prep(dloss)), t, e3 // 3, f32(ctx.margin),
                                                  ptr(dy), stream()))
        return dy, None


def triplet_hinge(y_pred, margin):
    return _TripletHinge.apply(y_pred, float(margin))


class _TripletGatherLoss(torch.autograd.Function):
    """mean_t max(|a-p|^2 - |a-n|^2 + m, 0) over the live triplets, rows gathered from emb."""

    @staticmethod
    def forward(ctx, emb, triplets, count, margin):
        emb = _prep(emb)
        n, e = emb.shape
        max_t = triplets.shape[0]
        loss = _new((max_t,), emb)
        act = _new((max_t,), emb)
        mean = _new((), emb)
        check(_lib.lib().embnet_triplet_gather_fwd(ptr(emb), n, e, ptr(triplets), ptr(count), max_t,
                                                   f32(margin), ptr(loss), ptr(act), ptr(mean), stream()))
        ctx.save_for_backward(emb, triplets, count, act)
        ctx.mark_non_differentiable(loss)
        return mean, loss

    @staticmethod
    def backward(ctx, dmean, _dloss):
        emb, triplets, count, act = ctx.saved_tensors
        n, e = emb.shape
        demb = torch.empty_like(emb)
        up = _prep(dmean)
        check(_lib.lib().embnet_triplet_gather_bwd(ptr(emb), n, e, ptr(triplets), ptr(count),
                                                   triplets.shape[0], ptr(act), ptr(up), ptr(demb), stream()))
        return demb, None, None, None


def triplet_gather_loss(emb, triplets, count, margin):
    """-> (mean loss scalar [autograd], per-triplet losses [max_t])."""
    return _TripletGatherLoss.apply(emb, triplets, count, float(margin))


_FUSED_WS = {}


def fused_loss_supported(k_classes, k_samples, e):
    return bool(_lib.lib().embnet_fused_loss_supported(int(k_classes), int(k_samples), int(e)))


class _FusedTripletLoss(torch.autograd.Function):
    """distance matrix + mining + gathered hinge + mean in one launch (embnet_fused_triplet_loss_fwd); the backward is
    the gather form's (embnet_triplet_gather_bwd), since the outputs are the same."""

    @staticmethod
    def forward(ctx, emb, p, k, margin, mode, seed, seed_dev=None):
        emb = _prep(emb)
        n, e = emb.shape
        lib = _lib.lib()
        code = 3 if mode == "batch_hard" else MINING_MODES[mode]
        rows = n if code == 3 else lib.embnet_mine_max_triplets(p, k)
        trip = _new((rows, 3), emb, torch.int32)
        count = _new((1,), emb, torch.int32)
        sel = _new((max(p * (k * (k - 1) // 2), 1),), emb, torch.int32)
        loss, act, mean = _new((rows,), emb), _new((rows,), emb), _new((), emb)
        key = (emb.device.index, stream(), p, k)
        ws = _FUSED_WS.get(key)
        if ws is None:                                      # zero-filled once; the kernel re-arms its counter itself
            ws = _FUSED_WS[key] = torch.zeros(max(lib.embnet_fused_loss_workspace_bytes(p, k) // 4, 8), device=emb.device)
        check(lib.embnet_fused_triplet_loss_fwd(ptr(emb), p, k, e, f32(margin), code, int(seed) & (2 ** 64 - 1),
                                                seed_dev, ptr(trip),
                                                ptr(count), ptr(sel), ptr(loss), ptr(act), ptr(mean), ptr(ws),
                                                ws.numel() * 4, stream()))
        ctx.save_for_backward(emb, trip, count, act)
        ctx.mark_non_differentiable(loss, trip, count)
        ctx.set_materialize_grads(False)                    # no zero tensors (three fill launches) for the outputs nobody differentiates
        return mean, loss, trip, count

    @staticmethod
    def backward(ctx, dmean, _dloss, _dtrip, _dcount):
        emb, trip, count, act = ctx.saved_tensors
        n, e = emb.shape
        demb = torch.empty_like(emb)
        check(_lib.lib().embnet_triplet_gather_bwd(ptr(emb), n, e, ptr(trip), ptr(count), trip.shape[0], ptr(act),
                                                   ptr(_prep(dmean)), ptr(demb), stream()))
        return demb, None, None, None, None, None, None


def fused_triplet_loss(emb, k_classes, k_samples, margin, mode, seed=0, seed_dev=None):
    """-> (mean loss [autograd], per-triplet losses, triplets [rows,3] int32, count [1] int32); one launch.
    seed_dev: device address of a uint64 seed that overrides `seed` (graph replays, see train_step.TripletTrainer)."""
    return _FusedTripletLoss.apply(emb, int(k_classes), int(k_samples), float(margin), mode, int(seed), seed_dev)


# --------------------------------------------------------------------------- contrastive / accuracy
class _Contrastive(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y_true, dist):
        y = _prep(y_true).reshape(-1)
        d = _prep(dist).reshape(-1)
        if y.numel() != d.numel():
            raise _lib.EmbnetError("contrastive_loss: y_true and y_pred sizes differ")
        out = _new((), d)
        check(_lib.lib().embnet_contrastive_fwd(ptr(y), ptr(d), d.numel(), ptr(out), stream()))
        ctx.save_for_backward(y, d)
        ctx.shape = dist.shape
        return out

    @staticmethod
    def backward(ctx, dout):
        y, d = ctx.saved_tensors
        dd = torch.empty_like(d)
        check(_lib.lib().embnet_contrastive_bwd(ptr(y), ptr(d), d.numel(), ptr(_prep(dout)), ptr(dd), stream()))
        return None, dd.reshape(ctx.shape)


def contrastive(y_true, dist):
    return _Contrastive.apply(y_true, dist)


def accuracy(y_true, dist):
    y = _prep(y_true.detach()).reshape(-1)
    d = _prep(dist.detach()).reshape(-1)
    out = _new((), d)
    check(_lib.lib().embnet_accuracy(ptr(y), ptr(d), d.numel(), ptr(out), stream()))
    return out


# --------------------------------------------------------------------------- embedding heads
class _L2Normalize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _prep(x)
        n, e = x.shape
        y = torch.empty_like(x)
        rn = _new((n,), x)
        check(_lib.lib().embnet_l2norm_fwd(ptr(x), n, e, ptr(y), ptr(rn), stream()))
        ctx.save_for_backward(y, rn)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, rn = ctx.saved_tensors
        n, e = y.shape
        dx = torch.empty_like(y)
        check(_lib.lib().embnet_l2norm_bwd(ptr(y), ptr(rn), ptr(_prep(dy)), n, e, ptr(dx), stream()))
        return dx


def l2_normalize(x):
    return _L2Normalize.apply(x)


class _PairDistance(torch.autograd.Function):
    @staticmethod
    def forward(ctx, e1, e2):
        e1, e2 = _prep(e1), _prep(e2)
        b, e = e1.shape
        d = _new((b, 1), e1)
        check(_lib.lib().embnet_pair_distance_fwd(ptr(e1), ptr(e2), b, e, ptr(d), stream()))
        ctx.save_for_backward(e1, e2, d)
        return d

    @staticmethod
    def backward(ctx, dd):
        e1, e2, d = ctx.saved_tensors
        b, e = e1.shape
        de1, de2 = torch.empty_like(e1), torch.empty_like(e2)
        check(_lib.lib().embnet_pair_distance_bwd(ptr(e1), ptr(e2), ptr(d), ptr(_prep(dd)), b, e,
                                                  ptr(de1), ptr(de2), stream()))
        return de1, de2


def pair_distance(e1, e2):
    """models.py:225: sqrt(max(sum (e1-e2)^2, 1e-7)), keepdims -> [b,1]."""
    return _PairDistance.apply(e1, e2)


# --------------------------------------------------------------------------- evaluation: kNN on embeddings
def cross_distances(q, x, squared=False):
    """[nq,e] x [n,e] -> [nq,n] Euclidean distances (sklearn euclidean_distances(Q, X))."""
    q, x = _prep(q.detach()), _prep(x.detach())
    nq, e = q.shape
    n = x.shape[0]
    if x.shape[1] != e:
        raise _lib.EmbnetError(f"cross_distances: widths differ ({e} vs {x.shape[1]})")
    lib = _lib.lib()
    ws = _new((max(lib.embnet_cross_dist_workspace_bytes(nq, n) // 4, 1),), q)
    d = _new((nq, n), q)
    check(lib.embnet_cross_dist_f32(ptr(q), nq, ptr(x), n, e, ptr(d), int(bool(squared)), ptr(ws), ws.numel() * 4,
                                    stream()))
    return d


def topk_smallest(dist, k):
    """-> (values [rows,k], indices [rows,k] int32), ascending, ties to the smaller column."""
    dist = _prep(dist)
    rows, n = dist.shape
    idx = _new((rows, k), dist, torch.int32)
    val = _new((rows, k), dist)
    check(_lib.lib().embnet_topk_smallest(ptr(dist), rows, n, int(k), ptr(idx), ptr(val), stream()))
    return val, idx


def knn_vote(idx, labels):
    """Majority label of each row's neighbours (labels int32 [n]); ties to the smallest label."""
    idx = idx.contiguous()
    labels = labels.to(torch.int32).contiguous()
    pred = _new((idx.shape[0],), idx, torch.int32)
    check(_lib.lib().embnet_knn_vote(ptr(idx), ptr(labels), idx.shape[0], idx.shape[1], ptr(pred), stream()))
    return pred


# --------------------------------------------------------------------------- softmax pre-training head
class _SoftmaxXent(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets):
        z, t = _prep(logits), _prep(targets)
        b, c = z.shape
        if t.shape != z.shape:
            raise _lib.EmbnetError(f"categorical_crossentropy: targets {tuple(t.shape)} vs logits {tuple(z.shape)}")
        prob, rows, corr = torch.empty_like(z), _new((b,), z), _new((b,), z)
        mean, acc = _new((), z), _new((), z)
        check(_lib.lib().embnet_softmax_xent_fwd(ptr(z), ptr(t), b, c, ptr(prob), ptr(rows), ptr(corr), ptr(mean),
                                                 ptr(acc), stream()))
        ctx.save_for_backward(prob, t)
        ctx.mark_non_differentiable(acc, prob)
        return mean, acc, prob

    @staticmethod
    def backward(ctx, dmean, _dacc, _dprob):
        prob, t = ctx.saved_tensors
        b, c = prob.shape
        dz = torch.empty_like(prob)
        check(_lib.lib().embnet_softmax_xent_bwd(ptr(prob), ptr(t), b, c, ptr(_prep(dmean)), ptr(dz), stream()))
        return dz, None


def softmax_cross_entropy(logits, targets):
    """Keras Dense(softmax) + 'categorical_crossentropy' + 'accuracy' from the logits:
    -> (mean loss [autograd], accuracy, probabilities)."""
    return _SoftmaxXent.apply(logits, targets)
