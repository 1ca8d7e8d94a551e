"""Drop-in for the generator side of embedding_net/datagenerators.py (reference :16-378).

The hot-path piece is TripletsDataGenerator: the reference samples P classes x K images, embeds them
with P predict() calls, builds the distance matrix with scikit-learn and mines triplets in a Python
double loop (:201-258).  Here the same contract (`__getitem__ -> ([A,P,N], targets)`) is served by one
batched inference forward + the HIP distance/mining kernels; `sample_batch()` additionally hands the
class-contiguous batch to the fused trainer (train_step.TripletTrainer), which is the fast path.

Image sources: `class_files_paths[class]` may be a list of file paths (decoded with PIL — cv2 is not
installed here — resized to input_shape, BGR order, /255 as reference :13-21,:156) or an in-memory
float array [n,H,W,3] already in [0,1] (synthetic data).  File IO is outside the hot path.
"""
import os
import random

import numpy as np
import torch

from . import ops


from ._decode_worker import get_image      # noqa: E402,F401  (torch-free module: the decode worker processes import it)


class ENDataLoader():
    """class -> file list from a directory tree (reference :89-111) or a CSV (:60-87), with the
    per-class train/val split (:51-58)."""

    def __init__(self, dataset_path, train_csv_file=None, val_csv_file=None, image_id_column='image_id',
                 label_column='label', validate=True, val_ratio=0.1, is_google=False):
        self.dataset_path = dataset_path
        self.class_names = []
        if train_csv_file is not None:
            self.class_files_paths = self._load_from_dataframe(train_csv_file, image_id_column, label_column, is_google)
        else:
            self.class_files_paths = self._load_from_directory()
        self.n_classes = len(self.class_names)
        self.n_samples = {k: len(v) for k, v in self.class_files_paths.items()}
        self.validate, self.val_ratio = validate, val_ratio
        if self.validate:
            if val_csv_file is not None:
                self.train_data = self.class_files_paths
                self.val_data = self._load_from_dataframe(val_csv_file, image_id_column, label_column, is_google)
            else:
                self.train_data, self.val_data = self.split_train_val(self.val_ratio)
        else:
            self.train_data, self.val_data = self.class_files_paths, {}

    def split_train_val(self, val_ratio):
        from sklearn.model_selection import train_test_split
        train_data, val_data = {}, {}
        for k, v in self.class_files_paths.items():
            train_data[k], val_data[k] = train_test_split(v, test_size=val_ratio, random_state=42)
        return train_data, val_data

    def _load_from_dataframe(self, csv_file, image_id_column, label_column, is_google=False):
        """reference :60-87: classes in order of first appearance in the CSV; paths are dataset_path/<image_id>, or, for
        the Google-Landmarks layout (is_google), dataset_path/<id[0]>/<id[1]>/<id[2]>/<id>.jpg.  (The reference also
        caches the result in ./tmp/data.pickle and silently reuses it for ANY later csv; that cache is not kept.)"""
        import pandas as pd
        df = pd.read_csv(csv_file)
        names = [str(c) for c in df[label_column].unique()]
        for cl in names:
            if cl not in self.class_names:
                self.class_names.append(cl)
        out = {}
        for cl, raw in zip(names, df[label_column].unique()):
            ids = [str(f) for f in df.loc[df[label_column] == raw][image_id_column]]
            if is_google:
                out[cl] = [os.path.join(self.dataset_path, f'{f[0]}/{f[1]}/{f[2]}/', f + '.jpg') for f in ids]
            else:
                out[cl] = [os.path.join(self.dataset_path, f) for f in ids]
        return out

    @staticmethod
    def _is_image(name):
        """reference :100-102: `.jpg`, or `.png` not starting with '._' (operator precedence as written there)."""
        return name.endswith('.jpg') or (name.endswith('.png') and not name.startswith('._'))

    def _load_from_directory(self):
        """reference :89-111: one class per sub-directory; images directly inside it, or inside its own sub-directories
        when it has any.  Listing order is sorted here (os.scandir order in the reference is arbitrary)."""
        out = {}
        for cl in sorted(os.listdir(self.dataset_path)):
            d = os.path.join(self.dataset_path, cl)
            if not os.path.isdir(d):
                continue
            self.class_names.append(cl)
            subdirs = [os.path.join(d, s) for s in sorted(os.listdir(d)) if os.path.isdir(os.path.join(d, s))]
            files = []
            for folder in (subdirs if subdirs else [d]):
                files += [os.path.join(folder, f) for f in sorted(os.listdir(folder))
                          if os.path.isfile(os.path.join(folder, f)) and self._is_image(f)]
            out[cl] = files
        return out


class SyntheticDataLoader:
    """In-memory stand-in with the same attributes: n_classes class prototypes + noise, images in [0,1]."""

    def __init__(self, n_classes, n_per_class, input_shape, noise=0.15, validate=True, val_ratio=0.2, seed=0):
        rs = np.random.RandomState(seed)
        h, w = input_shape[0], input_shape[1]
        self.class_names = [f"class_{i:03d}" for i in range(n_classes)]
        proto = rs.rand(n_classes, h, w, 3)
        self.class_files_paths = {
            c: np.clip(proto[i] + noise * rs.randn(n_per_class, h, w, 3), 0, 1).astype(np.float32)
            for i, c in enumerate(self.class_names)}
        self.n_classes = n_classes
        self.n_samples = {k: len(v) for k, v in self.class_files_paths.items()}
        self.validate, self.val_ratio = validate, val_ratio
        n_val = max(1, int(round(n_per_class * val_ratio))) if validate else 0
        self.train_data = {k: v[: len(v) - n_val] for k, v in self.class_files_paths.items()}
        self.val_data = {k: v[len(v) - n_val:] for k, v in self.class_files_paths.items()} if validate else {}


class ENDataGenerator:
    def __init__(self, class_files_paths, class_names, val_gen=False, input_shape=None, batch_size=32,
                 n_batches=10, n_batches_val=10, augmentations=None):
        self.input_shape, self.augmentations = input_shape, augmentations
        self.batch_size, self.n_batches, self.n_batches_val, self.val_gen = batch_size, n_batches, n_batches_val, val_gen
        self.class_files_paths, self.class_names = class_files_paths, class_names
        self.n_classes = len(self.class_names)
        self.n_samples = {k: len(v) for k, v in self.class_files_paths.items()}

    def __len__(self):
        return self.n_batches_val if self.val_gen else self.n_batches

    def _get_images_set(self, clsss, idxs, with_aug=True):
        """reference :145-156 -> float array [n,H,W,3] in [0,1]."""
        if type(clsss) is not list:
            clsss = [clsss] * len(idxs)
        imgs = []
        for cl, idx in zip(clsss, idxs):
            src = self.class_files_paths[cl]
            if isinstance(src, np.ndarray):
                imgs.append(src[idx])
            else:
                img = get_image(src[idx], self.input_shape)
                if with_aug and self.augmentations is not None:
                    img = self.augmentations(image=img)['image']
                imgs.append(np.asarray(img, np.float32) / 255.)
        return np.asarray(imgs, np.float32)


class TripletsDataGenerator(ENDataGenerator):

    def __init__(self, embedding_model, class_files_paths, class_names, n_batches=10, input_shape=None,
                 batch_size=32, augmentations=None, k_classes=5, k_samples=5, margin=0.5,
                 negatives_selection_mode='semihard'):
        super().__init__(class_files_paths=class_files_paths, class_names=class_names, input_shape=input_shape,
                         batch_size=batch_size, n_batches=n_batches, augmentations=augmentations)
        if negatives_selection_mode not in ops.MINING_MODES:
            raise KeyError(negatives_selection_mode)
        self.embedding_model = embedding_model
        self.k_classes, self.k_samples, self.margin = k_classes, k_samples, margin
        self.mode = negatives_selection_mode
        self._calls = 0

    def sample_plan(self):
        """reference :202-205: WHICH images make the next batch — P classes without replacement, K image indices per class WITH
        replacement — drawn from the global np.random stream in the reference's order.  -> (class names, [K indices] per class)."""
        selected = np.random.choice(self.n_classes, size=self.k_classes, replace=False)
        classes = [self.class_names[c] for c in selected]
        idxs = [np.random.choice(self.n_samples[cl], size=self.k_samples, replace=True) for cl in classes]
        return classes, idxs

    def plan_paths(self, plan):
        """The planned batch's image files in row order (class-contiguous) — what the decode worker processes are handed."""
        classes, idxs = plan
        return [self.class_files_paths[cl][int(i)] for cl, ix in zip(classes, idxs) for i in ix]

    def load_plan_u8(self, plan, out=None):
        """The planned batch as DECODED uint8 [P*K,H,W,3] (BGR, resized; augmentations applied), class-contiguous — what the
        input pipeline's worker threads run (embeddingnet_amd/input_pipeline.py); file-backed datasets only.
        NB with augmentations the worker threads draw from the library's global random state concurrently with this thread's plan
        draws, `depth` batches ahead: the PLANS keep the reference's order (they are drawn on the training thread), the
        augmentation draws do not — a seeded run with augmentations is reproducible only with workers=1, depth=1."""
        classes, idxs = plan
        h, w = self.input_shape[1], self.input_shape[0]     # get_image resizes to (input_shape[0], input_shape[1]) = (width, height)
        if out is None:
            out = np.empty((self.k_classes * self.k_samples, h, w, 3), np.uint8)
        row = 0
        for cl, ix in zip(classes, idxs):
            src = self.class_files_paths[cl]
            if isinstance(src, np.ndarray):
                raise TypeError("load_plan_u8: in-memory float datasets are not uint8 images")
            for i in ix:
                img = get_image(src[int(i)], self.input_shape)
                if self.augmentations is not None:
                    img = self.augmentations(image=img)['image']
                    # (an augmentation that returns floats — Normalize, ToFloat — would be truncated by the uint8 store below:
                    # refuse it here; sample_batch() / load_plan() take the float path for such pipelines — ADVICE r05)
                    if getattr(img, "dtype", None) != np.uint8 or img.shape != out[row].shape:
                        raise TypeError(f"load_plan_u8: the augmentation pipeline returned {getattr(img, 'dtype', type(img))} "
                                        f"{getattr(img, 'shape', None)}; the uint8 input pipeline needs uint8 {out[row].shape} "
                                        "(EMBNET_IMAGE_STORE=0 and the Feeder's float path, or drop the float transform)")
                out[row] = img
                row += 1
        return out

    def load_plan(self, plan):
        """The planned batch as the reference delivers it: float32 [P*K,H,W,3] in [0,1] (reference :211-218)."""
        classes, idxs = plan
        return np.vstack([self._get_images_set(cl, ix, with_aug=self.augmentations) for cl, ix in zip(classes, idxs)])

    def sample_batch(self):
        """reference :202-205,211-218: P classes without replacement, K images per class WITH replacement,
        stacked class-contiguous.  Returns float32 [P*K,H,W,3]."""
        return self.load_plan(self.sample_plan())

    def feeder(self, device, depth=10, workers=None, log=None):
        """The batches of sample_batch() as device tensors, produced ahead of the step (input_pipeline.Feeder)."""
        from .input_pipeline import Feeder
        return Feeder(self, device, depth=depth, workers=workers, log=log)

    def get_batch_triplets_mining(self):
        """reference :201-258 with the embedding / distance / mining work on the GPU."""
        return self.mine_batch(self.sample_batch())

    def mine_batch(self, images):
        """reference :211-258 on a given class-contiguous batch [P*K,H,W,3] (NumPy or device tensor): predict() ->
        distance matrix -> negative selection -> ([A,P,N], ones[T]).  The mined row indices stay in
        `self.last_triplets` (int64 [T,3], device)."""
        dev = next(self.embedding_model.parameters()).device
        x = images.to(dev) if torch.is_tensor(images) else torch.from_numpy(np.asarray(images, np.float32)).to(dev)
        was = self.embedding_model.training
        self.embedding_model.eval()                         # predict(): inference-mode BN, no dropout
        with torch.no_grad():
            emb = self.embedding_model(x)
            dist = ops.pairwise_distances(emb)
            self._calls += 1
            trip, count, _ = ops.mine_triplets(dist, self.k_classes, self.k_samples, self.margin, self.mode,
                                               seed=np.random.randint(0, 2 ** 31 - 1))
        self.embedding_model.train(was)
        t = self.last_triplets = trip[: int(count.item())].long()
        triplets = [x[t[:, 0]], x[t[:, 1]], x[t[:, 2]]]
        targets = torch.ones(len(t), device=dev)
        return triplets, targets

    def __getitem__(self, index):
        return self.get_batch_triplets_mining()


class SimpleTripletsDataGenerator(ENDataGenerator):
    """Random (anchor, positive, negative) triplets without mining (reference :264-314); validation generator."""

    def __init__(self, class_files_paths, class_names, input_shape=None, batch_size=32, n_batches=10,
                 augmentations=None, **kwargs):
        super().__init__(class_files_paths=class_files_paths, class_names=class_names, input_shape=input_shape,
                         batch_size=batch_size, n_batches=n_batches, augmentations=augmentations)

    def get_batch_triplets(self):
        a, p, n = [], [], []
        for _ in range(self.batch_size):
            ci = random.randrange(0, self.n_classes)
            cl = self.class_names[ci]
            other = self.class_names[(ci + random.randrange(1, self.n_classes)) % self.n_classes]
            n_cl = self.n_samples[cl]
            i1 = random.randrange(0, n_cl)
            i2 = (i1 + random.randrange(1, n_cl)) % n_cl if n_cl > 1 else i1
            i3 = random.randrange(0, self.n_samples[other])
            imgs = self._get_images_set([cl, cl, other], [i1, i2, i3], with_aug=self.augmentations)
            a.append(imgs[0]); p.append(imgs[1]); n.append(imgs[2])
        return [np.asarray(a), np.asarray(p), np.asarray(n)], np.ones((self.batch_size,), np.float32)

    def __getitem__(self, index):
        return self.get_batch_triplets()


class SiameseDataGenerator(ENDataGenerator):
    """Half same-class pairs (target 1), half different-class pairs (target 0) (reference :317-378)."""

    def __init__(self, class_files_paths, class_names, val_gen=False, input_shape=None, batch_size=32, n_batches=10,
                 n_batches_val=10, augmentations=None, **kwargs):
        super().__init__(class_files_paths=class_files_paths, class_names=class_names, val_gen=val_gen,
                         input_shape=input_shape, batch_size=batch_size, n_batches=n_batches,
                         n_batches_val=n_batches_val, augmentations=augmentations)

    def get_batch_pairs(self):
        x1, x2 = [], []
        targets = np.zeros((self.batch_size,), np.float32)
        n_same = self.batch_size // 2
        ci = random.randrange(0, self.n_classes)
        cl = self.class_names[ci]
        n_cl = self.n_samples[cl]
        indxs = np.random.randint(n_cl, size=self.batch_size)
        for i in range(self.batch_size):
            i1 = int(indxs[i])
            if i < n_same:
                i2 = (i1 + random.randrange(1, n_cl)) % n_cl if n_cl > 1 else i1
                imgs = self._get_images_set([cl, cl], [i1, i2], with_aug=self.augmentations)
                targets[i] = 1
            else:
                other = self.class_names[(ci + random.randrange(1, self.n_classes)) % self.n_classes]
                imgs = self._get_images_set([cl, other], [i1, random.randrange(0, self.n_samples[other])],
                                            with_aug=self.augmentations)
            x1.append(imgs[0]); x2.append(imgs[1])
        return [np.asarray(x1), np.asarray(x2)], targets

    def __getitem__(self, index):
        return self.get_batch_pairs()


class SimpleDataGenerator(ENDataGenerator):
    """Random images with one-hot class targets (reference :381-418); feeds the softmax pre-training."""

    def __init__(self, class_files_paths, class_names, input_shape=None, batch_size=32, n_batches=10,
                 augmentations=None):
        super().__init__(class_files_paths=class_files_paths, class_names=class_names, input_shape=input_shape,
                         batch_size=batch_size, n_batches=n_batches, augmentations=augmentations)

    def get_batch(self):
        images, targets = [], np.zeros((self.batch_size, self.n_classes), np.float32)
        for i in range(self.batch_size):
            ci = random.randrange(0, self.n_classes)
            cl = self.class_names[ci]
            images.append(self._get_images_set([cl], [random.randrange(0, self.n_samples[cl])],
                                               with_aug=self.augmentations)[0])
            targets[i][ci] = 1
        return [np.asarray(images, np.float32)], targets

    def __getitem__(self, index):
        return self.get_batch()
