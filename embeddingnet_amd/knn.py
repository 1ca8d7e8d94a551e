"""k-nearest-neighbour classifier on embeddings, with the slice of scikit-learn's KNeighborsClassifier
interface the reference uses (models.py:15,128-142: `.predict(encoding)`, `.kneighbors(encoding,
n_neighbors=5)` on `encoded_training_data['knn_classifier']`): brute-force Euclidean, uniform weights.
Distances come from the MFMA distance GEMM, selection and voting from HIP kernels (ops.py)."""
import numpy as np
import torch

from . import ops


class KNNClassifier:
    def __init__(self, n_neighbors=1, device=None):
        self.n_neighbors = int(n_neighbors)
        self.device = device

    def fit(self, encodings, labels):
        """encodings [n,e] (NumPy or tensor); labels: list of class names (any hashable)."""
        from .backbones import default_device
        dev = self.device or (encodings.device if torch.is_tensor(encodings) else default_device())
        self._x = torch.as_tensor(np.asarray(encodings) if not torch.is_tensor(encodings) else encodings,
                                  dtype=torch.float32, device=dev).contiguous()
        self.classes_ = np.array(sorted(set(labels)))
        lookup = {c: i for i, c in enumerate(self.classes_)}
        self._y = torch.tensor([lookup[l] for l in labels], dtype=torch.int32, device=dev)
        return self

    def _q(self, q):
        return torch.as_tensor(np.asarray(q) if not torch.is_tensor(q) else q, dtype=torch.float32,
                               device=self._x.device).reshape(-1, self._x.shape[1])

    def kneighbors(self, q, n_neighbors=None, return_distance=True):
        k = int(n_neighbors or self.n_neighbors)
        val, idx = ops.topk_smallest(ops.cross_distances(self._q(q), self._x), k)
        if return_distance:
            return val.cpu().numpy(), idx.cpu().numpy().astype(np.int64)
        return idx.cpu().numpy().astype(np.int64)

    def predict(self, q):
        _, idx = ops.topk_smallest(ops.cross_distances(self._q(q), self._x), self.n_neighbors)
        return self.classes_[ops.knn_vote(idx, self._y).cpu().numpy()]
