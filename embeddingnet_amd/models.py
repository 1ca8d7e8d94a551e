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


PARAM_SECTIONS = ('model', 'dataloader', 'generator', 'general', 'train')     # -> self.params_<section>


class EmbeddingNet:
    """Attribute surface of the reference class (models.py:24-40): the params sub-dicts as `params_<section>`
    (`params_softmax` only when the config has SOFTMAX_PRETRAINING), `base_model`, `backbone_model`, `model`,
    `workdir_path` = <work_dir>/<project_name>, `encoded_training_data`."""

    def __init__(self, params):
        for section in PARAM_SECTIONS:
            setattr(self, 'params_' + section, params[section])
        if 'softmax' in params:
            self.params_softmax = params['softmax']
        self.base_model = self.backbone_model = self.model = None
        general = self.params_general
        self.workdir_path = os.path.join(general['work_dir'], general['project_name'])
        self.encoded_training_data = {}

    def _create_base_model(self):
        """reference models.py:42-45: the backbone pair plus `classification_model` = Dense(1, sigmoid,
        name 'output_img') on the embedding."""
        self.base_model, self.backbone_model = get_backbone(**self.params_model)
        e = self.params_model.get('encodings_len', 4096)
        dev = next(self.base_model.parameters()).device
        self.classification_model = Model(_ClsHead(self.base_model, e), name="classification_model").to(dev)

    def _generate_encodings(self, imgs):
        return self.base_model.predict(imgs)

    # -- weights ---------------------------------------------------------------------------
    def _whole_model(self):
        """What a checkpoint covers: the trained graph (reference ModelCheckpoint saves `model.model`: base model plus,
        for SiameseNet, the 'output_siamese' / 'output_img' heads); the base model alone before one is built."""
        return self.model if self.model is not None else self.base_model

    def save_weights(self, path):
        np.savez(path, **{k: v.detach().cpu().numpy() for k, v in keras_weights(self._whole_model()).items()})

    def load_model(self, file_path):
        """Restore weights saved by save_weights (reference: keras load_model :92-98).  Base-model weights are
        required; head weights are taken when the file has them (older checkpoints hold the base model only)."""
        weights = np.load(file_path)
        load_keras_weights(self.base_model, weights)
        if self.model is not None:
            load_keras_weights(self.model, weights, strict=False)
        self.input_shape = list(self.params_model['input_shape'])

    def save_base_model(self, save_folder):
        os.makedirs(save_folder, exist_ok=True)
        self.save_weights(os.path.join(save_folder, "final_model.npz"))

    # -- encodings + kNN evaluation (reference models.py:61-90,128-161) ------
    def generate_encodings(self, data_loader, max_n_samples=10, shuffle=True):
        """{'paths', 'labels', 'encodings'} over up to max_n_samples training items per class
        (reference :61-83).  In-memory datasets (arrays instead of file lists) record row ids as paths."""
        import random
        from .datagenerators import get_image
        data_paths, data_labels, data_encodings = [], [], []
        for class_name in data_loader.class_names:
            data_list = data_loader.train_data[class_name]
            idx = list(range(len(data_list)))
            if len(idx) > max_n_samples:
                if shuffle:
                    random.shuffle(idx)
                idx = idx[:max_n_samples]
            if isinstance(data_list, np.ndarray):
                imgs = data_list[idx]
                data_paths += [f"{class_name}:{i}" for i in idx]
            else:
                paths = [data_list[i] for i in idx]
                imgs = np.asarray([get_image(p, self.params_model['input_shape']) for p in paths], np.float32) / 255.
                data_paths += paths
            encods = self._generate_encodings(imgs)
            data_encodings.extend(list(encods))
            data_labels += [class_name] * len(encods)
        return {'paths': data_paths, 'labels': data_labels, 'encodings': np.squeeze(np.array(data_encodings))}

    def train_embeddings_classifier(self, data_loader, classification_model, max_n_samples=10, shuffle=True):
        """reference models.py:52-59: fit any scikit-learn style classifier (`.fit(X, y)`) on the training encodings."""
        encodings = self.generate_encodings(data_loader, max_n_samples=max_n_samples, shuffle=shuffle)
        classification_model.fit(encodings['encodings'], encodings['labels'])

    def save_encodings(self, encoded_training_data, save_folder='./', save_file_name='encodings.pkl'):
        import pickle
        data = {k: v for k, v in encoded_training_data.items() if k != 'knn_classifier'}
        with open(os.path.join(save_folder, save_file_name), "wb") as f:
            pickle.dump(data, f)

    def load_encodings(self, path_to_encodings, knn_k=1):
        import pickle
        with open(path_to_encodings, 'rb') as f:
            self.encoded_training_data = pickle.load(f)
        return self.fit_knn(knn_k)

    def fit_knn(self, knn_k=1, encoded_training_data=None):
        """Build encoded_training_data['knn_classifier'] — the entry predict_knn reads (reference :134-137);
        the reference leaves its construction to external code."""
        from .knn import KNNClassifier
        if encoded_training_data is not None:
            self.encoded_training_data = encoded_training_data
        d = self.encoded_training_data
        dev = next(self.base_model.parameters()).device if self.base_model is not None else None
        d['knn_classifier'] = KNNClassifier(n_neighbors=knn_k, device=dev).fit(d['encodings'], d['labels'])
        return d['knn_classifier']

    def predict_knn(self, image, with_top5=False):
        """image: path, HxWx3 uint8 array, or a float [H,W,3] array already in [0,1]."""
        from .datagenerators import get_image
        if type(image) is str:
            img = np.asarray(get_image(image, self.params_model['input_shape']), np.float32) / 255.
        else:
            img = np.asarray(image, np.float32)
            if img.max() > 1.5:
                img = img / 255.
        encoding = self.base_model.predict(np.expand_dims(img, axis=0))
        knn = self.encoded_training_data['knn_classifier']
        predicted_label = knn.predict(encoding)
        if with_top5:
            idx = knn.kneighbors(encoding, n_neighbors=5)[1]
            return predicted_label, [self.encoded_training_data['labels'][idx[0][i]] for i in range(5)]
        return predicted_label

    def calculate_prediction_accuracy(self, data_loader):
        """top-1 / top-5 kNN accuracy over data_loader.val_data (batched: one distance GEMM per class)."""
        from .datagenerators import get_image
        knn = self.encoded_training_data['knn_classifier']
        labels = self.encoded_training_data['labels']
        top1 = top5 = total = 0
        for class_name, items in data_loader.val_data.items():
            if len(items) == 0:
                continue
            if isinstance(items, np.ndarray):
                imgs = items
            else:
                imgs = np.asarray([get_image(p, self.params_model['input_shape']) for p in items], np.float32) / 255.
            enc = self.base_model.predict(imgs)
            pred = knn.predict(enc)
            idx = knn.kneighbors(enc, n_neighbors=min(5, len(labels)))[1]
            top1 += int(np.sum(pred == class_name))
            top5 += sum(class_name in [labels[j] for j in row] for row in idx)
            total += len(imgs)
        return {'top1': top1 / max(total, 1), 'top5': top5 / max(total, 1)}


class _ClsHead(nn.Module):
    def __init__(self, base_model, e):
        super().__init__()
        self.base_model = base_model
        self.output_img = L.Dense(e, 1)

    def head(self, emb):
        return L.sigmoid(self.output_img(emb))               # Dense(units=1, activation='sigmoid'), models.py:44

    def forward(self, x):
        return self.head(self.base_model(x))


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
        # 'output_im1' / 'output_im2' (models.py:211-215): classification_model on each input.  It shares every
        # layer with base_model, so its value is the head applied to the embedding already computed (the reference
        # graph runs the shared layers a second time to the same result); they carry no loss in train.py:118.
        head = self.classification_model.net.head
        return [out, head(e1), head(e2)]


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
