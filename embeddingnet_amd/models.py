"""Drop-in for embedding_net/models.py: EmbeddingNet / TripletNet / SiameseNet with the
reference's constructor arguments and attributes (reference models.py:22-49,164-236).

Training-hot-path surface only: model assembly, `_generate_encodings`, save/load of weights.
The post-training kNN / encoding utilities (reference models.py:52-161) are outside the
hot path.
"""
import os

import numpy as np
import torch
from torch import nn

from . import layers as L
from . import ops
from .backbones import Model, get_backbone, keras_weights, load_keras_weights


class EmbeddingNet:

    def __init__(self, params):
        self.params_model = params['model']
        self.params_dataloader = params['dataloader']
        self.params_generator = params['generator']
        self.params_general = params['general']
        self.params_train = params['train']
        if 'softmax' in params:
            self.params_softmax = params['softmax']

        self.base_model = None
        self.backbone_model = None
        self.model = None

        self.workdir_path = os.path.join(self.params_general['work_dir'],
                                         self.params_general['project_name'])

        self.encoded_training_data = {}

    def _create_base_model(self):
        self.base_model, self.backbone_model = get_backbone(**self.params_model)
        # reference models.py:44-45: Dense(1, sigmoid) 'output_img' on the embedding.  Only its
        # pre-activation is on any loss path the reference can run, so the head stays linear here
        # and SiameseNet applies the sigmoid where it is consumed.
        e = self.params_model.get('encodings_len', 4096)
        dev = next(self.base_model.parameters()).device
        self.classification_model = Model(_ClsHead(self.base_model, e), name="classification_model").to(dev)

    def _generate_encodings(self, imgs):
        return self.base_model.predict(imgs)

    # -- weights ---------------------------------------------------------------------------
    def save_weights(self, path):
        np.savez(path, **{k: v.detach().cpu().numpy() for k, v in keras_weights(self.base_model).items()})

    def load_model(self, file_path):
        """Restore base-model weights saved by save_weights (reference: keras load_model :92-98)."""
        load_keras_weights(self.base_model, np.load(file_path))
        self.input_shape = list(self.params_model['input_shape'])

    def save_base_model(self, save_folder):
        os.makedirs(save_folder, exist_ok=True)
        self.save_weights(os.path.join(save_folder, "final_model.npz"))


class _ClsHead(nn.Module):
    def __init__(self, base_model, e):
        super().__init__()
        self.base_model = base_model
        self.output_img = L.Dense(e, 1)

    def forward(self, x):
        return self.output_img(self.base_model(x))


class _TripletGraph(nn.Module):
    """[a, p, n] -> concat(base(a), base(p), base(n)) on the last axis (reference models.py:176-186)."""

    def __init__(self, base_model):
        super().__init__()
        self.base_model = base_model

    def forward(self, inputs):
        a, p, n = inputs
        return torch.cat([self.base_model(a), self.base_model(p), self.base_model(n)], dim=-1)


class TripletNet(EmbeddingNet):

    def __init__(self, params, training=False):
        super().__init__(params)

        self.training = training

        if self.training:
            self._create_base_model()
            self._create_model_triplet()

    def _create_model_triplet(self):
        self.model = Model(_TripletGraph(self.base_model), name="triplet_model")


class _SiameseGraph(nn.Module):
    """[x1, x2] -> [distance, cls1, cls2] (reference models.py:203-230); 'l2' distance head."""

    def __init__(self, base_model, classification_model, distance_type, encodings_len):
        super().__init__()
        if distance_type not in ('l1', 'l2'):
            raise KeyError(f"distance_type '{distance_type}' (reference supports 'l1' and 'l2')")
        self.base_model, self.classification_model = base_model, classification_model
        self.distance_type = distance_type
        if distance_type == 'l1':
            self.output_siamese = L.Dense(encodings_len, 1)          # reference models.py:221

    def forward(self, inputs):
        x1, x2 = inputs
        e1, e2 = self.base_model(x1), self.base_model(x2)
        if self.distance_type == 'l1':                               # models.py:217-221
            out = L.sigmoid(self.output_siamese(L.abs_diff(e1, e2)))
        else:                                                        # models.py:223-228
            out = ops.pair_distance(e1, e2)
        # The two 'output_im*' classification outputs (models.py:211-215) carry no loss in train.py:118;
        # they are not evaluated here (each would be one more backbone forward).
        return [out, None, None]


class SiameseNet(EmbeddingNet):

    def __init__(self, params, training):
        super().__init__(params)

        self.training = training

        if self.training:
            self._create_base_model()
            self._create_model_siamese()

    def _create_model_siamese(self):
        dev = next(self.base_model.parameters()).device
        self.model = Model(_SiameseGraph(self.base_model, self.classification_model,
                                         self.params_model['distance_type'],
                                         self.params_model.get('encodings_len', 4096)), name="siamese_model").to(dev)
