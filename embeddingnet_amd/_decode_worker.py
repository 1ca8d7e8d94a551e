"""Image decode for the input pipeline's worker PROCESSES (input_pipeline.BatchPrefetcher(processes=True)).

Kept free of torch (a spawned worker imports numpy and PIL only: ~0.2 s and ~40 MB each instead of torch's 1.5 s / 300 MB).
Workers write decoded uint8 images straight into a shared [slots, B, H, W, 3] array backed by a file in /dev/shm that the
parent maps too; nothing is pickled but file names and row numbers."""
import os

import numpy as np

_ARR = None


def get_image(img_path, input_shape=None):
    """reference utils.py:13-21 (cv2.imread + resize) on PIL: uint8 HxWx3, BGR channel order."""
    from PIL import Image
    if not os.path.exists(img_path):
        print('image is not exist ' + img_path)
        return None
    img = Image.open(img_path).convert("RGB")
    if input_shape and img.size != (input_shape[0], input_shape[1]):
        img = img.resize((input_shape[0], input_shape[1]), Image.BILINEAR)
    return np.asarray(img)[:, :, ::-1]


def attach(path, shape):
    """Pool initializer: map the parent's staging array."""
    global _ARR
    _ARR = np.memmap(path, dtype=np.uint8, mode="r+", shape=tuple(shape))


def decode_rows(slot, row0, paths, input_shape):
    """paths -> rows row0.. of staging slot `slot`.  Returns the number of rows written."""
    for j, p in enumerate(paths):
        img = get_image(p, input_shape)
        if img is None:
            raise FileNotFoundError(p)
        _ARR[slot, row0 + j] = img
    return len(paths)


def main(argv):
    """`python -m embeddingnet_amd._decode_worker <staging file> <slots,B,H,W,3>`: one JSON task per stdin line
    [slot, row0, [paths], [input_shape]] -> one reply line ("ok <rows>" or "err <message>")."""
    import json
    import sys
    # the reply channel is the ORIGINAL stdout; anything else this process prints (get_image's 'image is not exist') goes to stderr
    proto = os.fdopen(os.dup(1), "w", buffering=1)
    os.dup2(2, 1)
    sys.stdout = sys.stderr
    attach(argv[0], [int(v) for v in argv[1].split(",")])
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        try:
            slot, row0, paths, input_shape = json.loads(line)
            print("ok", decode_rows(slot, row0, paths, input_shape), file=proto, flush=True)
        except Exception as e:          # noqa: BLE001 — reported to the parent, which raises it on the consumer's thread
            print("err", type(e).__name__, str(e).replace("\n", " "), file=proto, flush=True)


if __name__ == "__main__":
    import sys
    main(sys.argv[1:])
