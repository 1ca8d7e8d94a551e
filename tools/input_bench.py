#!/usr/bin/env python3
"""Real-data throughput of the training step against the synthetic step.

Writes a JPEG tree (`--classes` x `--per-class` files at the config's resolution, quality 90) to a scratch directory, then times
the fused triplet step of a bench config (c1: simple2 64x64 8x4; c2: ResNet18 224x224 32x4) fed four ways:
  resident     one batch already in HBM, stepped on repeatedly (what bench.py times, minus its batch cycling)
  sequential   the round-4 path: sample_batch() (PIL decode + /255 on the training thread) -> from_numpy -> blocking copy
  prefetch     input_pipeline.BatchPrefetcher: worker PROCESSES decode `depth` batches ahead into a shared, page-locked uint8
               staging array, side-stream copy (prefetch_threads: the same with a thread pool — GIL-bound)
  store        input_pipeline.DeviceImageStore: dataset decoded once, uint8 resident in HBM, one gather+convert kernel per batch
Prints one JSON object (images/s each, the host core count, decode rate, store build time)."""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

CONFIGS = {"c1": dict(backbone="simple2", image=64, p=8, k=4, enc=256, mining="semihard"),
           "c2": dict(backbone="resnet18", image=224, p=32, k=4, enc=256, mining="hardest")}


def make_tree(root, classes, per_class, size, seed=0):
    from PIL import Image
    rs = np.random.RandomState(seed)
    proto = rs.rand(classes, size, size, 3)
    for c in range(classes):
        os.makedirs(os.path.join(root, f"class_{c:03d}"), exist_ok=True)
        for i in range(per_class):
            arr = np.clip(proto[c] + 0.15 * rs.randn(size, size, 3), 0, 1)
            Image.fromarray((arr * 255).astype(np.uint8)).save(os.path.join(root, f"class_{c:03d}", f"{i:04d}.jpg"), quality=90)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--classes", type=int, default=107)
    ap.add_argument("--per-class", type=int, default=50)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--depth", type=int, default=10)
    ap.add_argument("--workers", type=int, default=0)
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    from embeddingnet_amd import backbones as B
    from embeddingnet_amd import input_pipeline as IP
    from embeddingnet_amd.datagenerators import ENDataLoader, TripletsDataGenerator
    from embeddingnet_amd.train_step import TripletTrainer
    from embeddingnet_amd.utils import get_optimizer
    dev = torch.device("cuda:0")
    workers = args.workers or IP.default_workers()
    tmp = tempfile.mkdtemp(prefix="embnet_jpeg_")
    t0 = time.perf_counter()
    make_tree(tmp, args.classes, args.per_class, cfg["image"])
    t_tree = time.perf_counter() - t0
    dl = ENDataLoader(tmp, validate=False)
    shape = (cfg["image"], cfg["image"], 3)
    gen = TripletsDataGenerator(embedding_model=None, class_files_paths=dl.train_data, class_names=dl.class_names,
                                input_shape=list(shape), k_classes=cfg["p"], k_samples=cfg["k"], margin=0.5,
                                negatives_selection_mode=cfg["mining"])
    model, _ = B.get_backbone(shape, encodings_len=cfg["enc"], backbone_name=cfg["backbone"], backbone_weights=None, seed=0, device=dev)
    opt = get_optimizer("radam", 1e-4).build([p for p in model.parameters() if p.requires_grad])
    trainer = TripletTrainer(model, opt, cfg["p"], cfg["k"], margin=0.5, negatives_selection_mode=cfg["mining"], seed=0, graph="auto")
    n = cfg["p"] * cfg["k"]

    def timed(next_batch, steps=args.steps, warm=args.warmup):
        for _ in range(warm):
            trainer.step(next_batch())
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            trainer.step(next_batch())
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        return round(n * steps / dt, 1), round(1e3 * dt / steps, 3)

    out = {"config": args.config, "backbone": cfg["backbone"], "image": cfg["image"], "batch": n, "host_cores": len(os.sched_getaffinity(0)),
           "decode_workers": workers, "dataset": f"{args.classes} classes x {args.per_class} JPEG files (quality 90) at {cfg['image']}x{cfg['image']}",
           "tree_written_in_s": round(t_tree, 1)}
    # decode rate of one thread and of the pool (files are cached by the OS after the first pass)
    files = [f for v in dl.train_data.values() for f in v][:256]
    t = time.perf_counter()
    for f in files[:64]:
        IP.decode_u8(f, shape)
    out["decode_images_per_s_one_thread"] = round(64 / (time.perf_counter() - t), 1)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(workers) as pool:
        t = time.perf_counter()
        list(pool.map(lambda f: IP.decode_u8(f, shape), files))
        out["decode_images_per_s_pool"] = round(len(files) / (time.perf_counter() - t), 1)
    np.random.seed(0)
    xres = torch.from_numpy(gen.sample_batch()).to(dev)
    res_pool = [torch.from_numpy(gen.sample_batch()).to(dev) for _ in range(16)]
    tick = [0]

    def cyc():
        tick[0] += 1
        return res_pool[tick[0] % len(res_pool)]
    out["resident_images_per_s"], out["resident_ms_per_step"] = timed(cyc)
    seq_steps = max(6, min(args.steps, int(20 * out["decode_images_per_s_one_thread"] / n)))       # ~20 s of it at most
    out["sequential_images_per_s"], out["sequential_ms_per_step"] = timed(lambda: torch.from_numpy(gen.sample_batch()).to(dev),
                                                                          steps=seq_steps, warm=2)
    pf = IP.BatchPrefetcher(gen.sample_plan, gen.load_plan_u8, (n,) + shape, dev, depth=args.depth, workers=workers)
    out["prefetch_threads_images_per_s"], out["prefetch_threads_ms_per_step"] = timed(pf.next, steps=max(10, seq_steps))
    pf.close()
    pf = IP.BatchPrefetcher(gen.sample_plan, gen.load_plan_u8, (n,) + shape, dev, depth=args.depth, workers=workers,
                            paths_fn=gen.plan_paths, input_shape=list(shape))
    out["prefetch_staging_pinned"] = bool(getattr(pf, "pinned", False))
    out["prefetch_images_per_s"], out["prefetch_ms_per_step"] = timed(pf.next)
    pf.close()
    msgs = []
    store = IP.DeviceImageStore(dl.train_data, dl.class_names, shape, dev, workers=workers, log=msgs.append)
    out["store_build_s"], out["store_mib"] = round(store.decode_seconds, 2), round(store.data.numel() / 2 ** 20, 1)
    out["store_images_per_s"], out["store_ms_per_step"] = timed(lambda: store.batch(gen.sample_plan()))
    for key in ("sequential", "prefetch", "store"):
        out[f"{key}_over_resident"] = round(out[f"{key}_images_per_s"] / out["resident_images_per_s"], 3)
    out["step_mode"] = "hip graph" if getattr(trainer, "_graph", None) is not None else "eager"
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
