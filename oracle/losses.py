"""Oracle (test infrastructure): the reference's three loss/metric functions
restated in NumPy float64, plus their analytic gradients.

Follows /root/reference/embedding_net/losses_and_accuracies.py:
  contrastive_loss :4-11, triplet_loss :14-44, accuracy :47-50.
Pinned by tests/golden/triplet_loss.npz and siamese_losses.npz.
"""
import numpy as np


def triplet_loss(margin=0.5):
    """losses_and_accuracies.py:14 — returns loss_function(y_true, y_pred)."""

    def loss_function(y_true, y_pred):
        y = np.asarray(y_pred, np.float64)
        total = y.shape[-1]                                   # :27
        a = y[:, 0:int(total * 1 / 3)]                        # :29
        p = y[:, int(total * 1 / 3):int(total * 2 / 3)]       # :30
        n = y[:, int(total * 2 / 3):int(total * 3 / 3)]       # :31
        pos = np.sum(np.square(a - p), axis=1)                # :34
        neg = np.sum(np.square(a - n), axis=1)                # :37
        return np.maximum(pos - neg + margin, 0.0)            # :40-41, shape [T]

    return loss_function


def triplet_loss_grad(margin, y_pred, upstream):
    """d(sum_t upstream[t] * loss[t]) / d y_pred.  TF's `maximum` sends the
    gradient to its first argument when the two are equal, so a row is active
    iff pos - neg + margin >= 0."""
    y = np.asarray(y_pred, np.float64)
    e = y.shape[-1] // 3
    a, p, n = y[:, :e], y[:, e:2 * e], y[:, 2 * e:]
    basic = np.sum((a - p) ** 2, 1) - np.sum((a - n) ** 2, 1) + margin
    g = (np.asarray(upstream, np.float64) * (basic >= 0.0))[:, None]
    return np.concatenate([2 * (n - p) * g, 2 * (p - a) * g, 2 * (a - n) * g], axis=1)


def contrastive_loss(y_true, y_pred):
    """losses_and_accuracies.py:4 — margin hard-coded 1 (:8); y=1 same class."""
    y = np.asarray(y_true, np.float64)
    d = np.asarray(y_pred, np.float64)
    margin = 1
    return np.mean(y * np.square(d) + (1 - y) * np.square(np.maximum(margin - d, 0)))


def contrastive_loss_grad(y_true, y_pred):
    y = np.asarray(y_true, np.float64)
    d = np.asarray(y_pred, np.float64)
    return (2 * y * d - 2 * (1 - y) * np.maximum(1 - d, 0)) / d.size


def accuracy(y_true, y_pred):
    """losses_and_accuracies.py:47 — fixed 0.5 threshold on distances."""
    y = np.asarray(y_true)
    return np.mean(np.equal(y, (np.asarray(y_pred) < 0.5).astype(y.dtype)))


def softmax_cross_entropy(logits, targets):
    """Keras Dense(softmax) + loss='categorical_crossentropy' (reference backbones.py:146-151; TF computes it
    from the logits of the softmax op) and metric 'accuracy'.  Third-party arithmetic (TensorFlow 2.2):
    parity unpinned, standard definition.  Returns (mean loss, accuracy, probabilities, dlogits of the mean)."""
    z = np.asarray(logits, np.float64)
    t = np.asarray(targets, np.float64)
    zs = z - z.max(axis=1, keepdims=True)
    lse = np.log(np.exp(zs).sum(axis=1, keepdims=True))
    logp = zs - lse
    loss = -(t * logp).sum(axis=1)
    acc = np.mean(z.argmax(1) == t.argmax(1))
    prob = np.exp(logp)
    return loss.mean(), acc, prob, (prob * t.sum(1, keepdims=True) - t) / z.shape[0]
