"""Drop-in for embedding_net/losses_and_accuracies.py (same names, arguments and
meaning), computing on the GPU through libembnet_hip.so.

  contrastive_loss(y_true, y_pred)          reference :4-11
  triplet_loss(margin)(y_true, y_pred)      reference :14-44  -> per-row [T] (caller means)
  accuracy(y_true, y_pred)                  reference :47-50
Inputs are torch CUDA tensors; outputs carry autograd.
"""
from . import ops


def contrastive_loss(y_true, y_pred):
    '''Contrastive loss (Hadsell et al. 2006), margin fixed to 1, y_true 1 = same class.'''
    return ops.contrastive(y_true, y_pred)


def triplet_loss(margin=0.5):
    """Returns loss_function(y_true, y_pred); y_pred is [T, 3E] = concat(anchor,
    positive, negative) on the last axis, y_true is ignored (Keras signature)."""

    def loss_function(y_true, y_pred):
        return ops.triplet_hinge(y_pred, margin)

    return loss_function


def accuracy(y_true, y_pred):
    '''Classification accuracy with a fixed 0.5 threshold on distances.'''
    return ops.accuracy(y_true, y_pred)
