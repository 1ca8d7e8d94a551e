"""Oracle (test infrastructure): the update rules of the optimizers the reference builds in
/root/reference/embedding_net/utils.py:143-153 — `optimizers.Adam(lr)`, `optimizers.RMSprop(lr)`,
`keras_radam.RAdam(lr)`, `optimizers.SGD(lr)` — restated in NumPy float64, one function per rule.

PARITY UNPINNED: tensorflow 2.2 (requirements.txt:2, pinned) and keras-rectified-adam
(requirements.txt:6, unpinned) are not installable here and the reference holds no optimizer vectors.
The rules follow the published implementations the reference's call sites reach with its arguments
(only `lr` is passed, everything else is the library default):

  tf.keras.optimizers.SGD      momentum 0:        w <- w - lr*g
  tf.keras.optimizers.Adam     b1 .9 b2 .999 eps 1e-7, amsgrad off (fused ResourceApplyAdam):
                                 lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m <- b1 m + (1-b1) g; v <- b2 v + (1-b2) g^2
                                 w <- w - lr_t * m / (sqrt(v) + eps)            (eps is NOT bias-corrected)
  tf.keras.optimizers.RMSprop  rho .9 momentum 0 eps 1e-7, not centered, rms slot starts at 0:
                                 rms <- rho rms + (1-rho) g^2;  w <- w - lr * g / (sqrt(rms) + eps)
  keras_radam.RAdam            b1 .9 b2 .999 eps K.epsilon()=1e-7, no weight decay, no warm-up (total_steps 0):
                                 m, v as Adam; m^ = m/(1-b1^t); v^ = sqrt(v/(1-b2^t))
                                 sma_inf = 2/(1-b2) - 1; sma_t = sma_inf - 2 t b2^t/(1-b2^t)
                                 r_t = sqrt((sma_t-4)/(sma_inf-4) * (sma_t-2)/(sma_inf-2) * sma_inf/sma_t)
                                 w <- w - lr * (r_t * m^/(v^ + eps) if sma_t >= 5 else m^)
t counts from 1.  Each class keeps its slots per parameter index and updates a list of arrays in place.
"""
import numpy as np


class _Base:
    def __init__(self, lr):
        self.lr, self.t, self.slots = float(lr), 0, {}

    def slot(self, i, name, like):
        return self.slots.setdefault((i, name), np.zeros_like(like, dtype=np.float64))

    def step(self, params, grads):
        """params, grads: lists of float64 arrays; params are updated in place."""
        self.t += 1
        for i, (w, g) in enumerate(zip(params, grads)):
            if g is not None:
                self.update(i, w, np.asarray(g, np.float64))


class SGD(_Base):
    def update(self, i, w, g):
        w -= self.lr * g


class Adam(_Base):
    b1, b2, eps = 0.9, 0.999, 1e-7

    def update(self, i, w, g):
        m, v = self.slot(i, "m", w), self.slot(i, "v", w)
        m[...] = self.b1 * m + (1 - self.b1) * g
        v[...] = self.b2 * v + (1 - self.b2) * g * g
        lr_t = self.lr * np.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        w -= lr_t * m / (np.sqrt(v) + self.eps)


class RMSprop(_Base):
    rho, eps = 0.9, 1e-7

    def update(self, i, w, g):
        rms = self.slot(i, "rms", w)
        rms[...] = self.rho * rms + (1 - self.rho) * g * g
        w -= self.lr * g / (np.sqrt(rms) + self.eps)


class RAdam(_Base):
    b1, b2, eps = 0.9, 0.999, 1e-7

    def update(self, i, w, g):
        m, v = self.slot(i, "m", w), self.slot(i, "v", w)
        t = self.t
        m[...] = self.b1 * m + (1 - self.b1) * g
        v[...] = self.b2 * v + (1 - self.b2) * g * g
        m_hat = m / (1 - self.b1 ** t)
        sma_inf = 2.0 / (1 - self.b2) - 1.0
        sma_t = sma_inf - 2.0 * t * self.b2 ** t / (1 - self.b2 ** t)
        if sma_t >= 5:
            v_hat = np.sqrt(v / (1 - self.b2 ** t))
            r_t = np.sqrt((sma_t - 4) / (sma_inf - 4) * (sma_t - 2) / (sma_inf - 2) * sma_inf / sma_t)
            w -= self.lr * r_t * m_hat / (v_hat + self.eps)
        else:
            w -= self.lr * m_hat


def get_optimizer(name, learning_rate):
    """utils.py:143-153 — same name dispatch ('adam', 'rms_prop', 'radam', anything else -> SGD)."""
    return {"adam": Adam, "rms_prop": RMSprop, "radam": RAdam}.get(name, SGD)(learning_rate)
