"""Feeding the training step: decoded images reach the GPU as uint8 and ahead of the step that needs them.

The reference trains through Keras `fit_generator` (/root/reference/tools/train.py:172-177): its `Sequence` runs behind a
background enqueuer (queue depth 10) whose `__getitem__` decodes, resizes and divides by 255 on the host
(embedding_net/datagenerators.py:145-156) and hands float32 batches over.  With an 11 k images/s step a synchronous,
single-threaded decode on the training thread would bound the whole job, so:

  * a batch is described by a PLAN — which classes, which image of each — drawn on the TRAINING thread in exactly the order
    and from exactly the `np.random` stream the reference's sampler uses (datagenerators.py:202-205), several batches ahead;
  * `DeviceImageStore` (no augmentations, dataset fits the budget): every image is decoded ONCE by a thread pool (PIL's decode
    releases the GIL), stays resident in HBM as uint8 (a 224x224 image is 147 KB: 100 000 of them are 15 GB of the 288), and a
    step's batch is ONE kernel — gather by index, convert, divide (embnet_u8_to_f32).  Nothing but a few hundred index bytes
    crosses PCIe per step;
  * `BatchPrefetcher` (augmentations, or a dataset beyond the budget): worker threads decode the planned batches into a ring of
    PINNED uint8 buffers `depth` batches ahead; the consumer copies a ready buffer on a side stream (non-blocking, 1 byte per
    value instead of the reference's 4) and converts on the compute stream.

Both hand `TripletTrainer.step` the class-contiguous float32 [P*K, H, W, 3] tensor `sample_batch()` would have produced, value
for value (float32 `x / 255.`).
"""
import os
import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream


def default_workers():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    return max(1, min(32, n - 1))


def decode_u8(path, input_shape):
    """One image as the reference's `get_image` delivers it (utils.py:13-21): uint8 [H, W, 3], BGR, resized to input_shape."""
    from .datagenerators import get_image
    img = get_image(path, input_shape)
    if img is None:
        raise FileNotFoundError(path)
    return img


def u8_to_f32(src_u8, index, n, out=None, pad_to=None):
    """float32 [n, H, W, C'] = src_u8[index or :n] / 255 on the device (embnet_u8_to_f32)."""
    _, h, w, c = src_u8.shape
    c_out = pad_to or c
    if out is None:
        out = torch.empty((n, h, w, c_out), device=src_u8.device, dtype=torch.float32)
    check(_lib.lib().embnet_u8_to_f32(ptr(src_u8), ptr(index), n, h * w, c, c_out, 255.0, ptr(out), stream()))
    return out


class DeviceImageStore:
    """The training images of a `class_files_paths` dict, decoded once and resident in HBM as uint8 [N, H, W, 3]."""

    def __init__(self, class_files_paths, class_names, input_shape, device, workers=None, chunk=512, log=None):
        self.device = torch.device(device)
        # (rows, columns) of a decoded image: get_image resizes to (input_shape[0], input_shape[1]) = (width, height), the
        # reference's cv2.resize call (utils.py:13-21) — the same thing for the square shapes every shipped config uses
        self.input_shape = [int(input_shape[0]), int(input_shape[1]), 3]
        self.shape = (int(input_shape[1]), int(input_shape[0]), 3)
        self.first, files = {}, []
        for cl in class_names:
            self.first[cl] = len(files)
            src = class_files_paths[cl]
            if isinstance(src, np.ndarray):
                raise TypeError("DeviceImageStore holds decoded files; in-memory float datasets need no store")
            files += list(src)
        self.n = len(files)
        h, w, c = self.shape
        self.data = torch.empty((self.n, h, w, c), device=self.device, dtype=torch.uint8)
        workers = workers or default_workers()
        import time
        t0 = time.perf_counter()
        with ThreadPoolExecutor(workers) as pool:
            for lo in range(0, self.n, chunk):
                part = files[lo:lo + chunk]
                host = torch.empty((len(part), h, w, c), dtype=torch.uint8).pin_memory()
                view = host.numpy()

                def one(i, path=None):
                    view[i] = decode_u8(part[i], self.input_shape)
                list(pool.map(one, range(len(part))))
                self.data[lo:lo + len(part)].copy_(host, non_blocking=False)
        self.decode_seconds = time.perf_counter() - t0
        if log:
            log(f"DeviceImageStore: {self.n} images {h}x{w} decoded by {workers} threads in {self.decode_seconds:.1f} s "
                f"({self.n / max(self.decode_seconds, 1e-9):.0f} images/s), {self.data.numel() / 2 ** 20:.0f} MiB resident in HBM")

    @staticmethod
    def bytes_needed(class_files_paths, input_shape):
        return sum(len(v) for v in class_files_paths.values()) * int(input_shape[0]) * int(input_shape[1]) * 3

    def batch(self, plan, pad_to=None):
        """plan = (classes, idxs) as TripletsDataGenerator.sample_plan() draws it -> float32 [P*K, H, W, 3] on the device."""
        classes, idxs = plan
        rows = np.concatenate([self.first[cl] + np.asarray(ix, np.int64) for cl, ix in zip(classes, idxs)]).astype(np.int32)
        index = torch.from_numpy(rows).to(self.device, non_blocking=True)
        return u8_to_f32(self.data, index, len(rows), pad_to=pad_to)


class DecodeProcesses:
    """`n` worker processes (`python -m embeddingnet_amd._decode_worker`: numpy + PIL, no torch) that decode image files into a
    shared uint8 staging array [slots, B, H, W, 3] backed by a file in /dev/shm.  PIL's decode releases the GIL only inside
    libjpeg; open / convert / resize / array conversion hold it, so a THREAD pool tops out near one core's rate (measured:
    64x64 JPEGs 11 200 images/s on one thread, 4 800 on 32 threads; 224x224: 2 400 vs 2 900 — profiles/r05_input_bench_*.json).
    run() is called from the prefetcher's threads: it borrows an idle worker, writes one task line, blocks on the reply."""

    def __init__(self, n, slots_shape):
        import subprocess
        import sys
        import tempfile
        self.shape = tuple(int(v) for v in slots_shape)
        nbytes = int(np.prod(self.shape))
        fd, self.path = tempfile.mkstemp(prefix="embnet_stage_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        os.ftruncate(fd, nbytes)
        os.close(fd)
        self.array = np.memmap(self.path, dtype=np.uint8, mode="r+", shape=self.shape)
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
        self.procs, self.idle = [], queue.Queue()
        for _ in range(n):
            p = subprocess.Popen([sys.executable, "-m", "embeddingnet_amd._decode_worker", self.path, ",".join(map(str, self.shape))],
                                 stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, bufsize=1, env=env)
            self.procs.append(p)
            self.idle.put(p)

    def run(self, slot, row0, paths, input_shape):
        import json
        p = self.idle.get()
        try:
            p.stdin.write(json.dumps([int(slot), int(row0), list(paths), [int(v) for v in input_shape]]) + "\n")
            p.stdin.flush()
            reply = p.stdout.readline()
        finally:
            self.idle.put(p)
        if not reply.startswith("ok"):
            raise RuntimeError(f"decode worker: {reply.strip() or 'died'}")

    def close(self):
        for p in self.procs:
            try:
                p.stdin.close()
                p.terminate()
            except OSError:
                pass
        for p in self.procs:                 # reap them: a terminated child that nobody waits for stays a zombie
            try:
                p.wait(timeout=5)
            except Exception:                # noqa: BLE001
                try:
                    p.kill(); p.wait(timeout=5)
                except Exception:            # noqa: BLE001
                    pass
        self.procs = []
        try:
            os.unlink(self.path)
        except OSError:
            pass


class BatchPrefetcher:
    """Planned batches decoded `depth` ahead by worker threads into pinned uint8 buffers; `next()` returns the float32 device
    tensor of the oldest one.  `plan_fn()` is called on the consumer's thread (it draws from np.random — the reference's
    sampling stream stays single-threaded and in order); `load_fn(plan, out_u8)` fills a [B, H, W, 3] uint8 array from worker
    threads."""

    def __init__(self, plan_fn, load_fn, batch_shape, device, depth=10, workers=None, paths_fn=None, input_shape=None, rows_per_task=8):
        """paths_fn(plan) -> the batch's file paths in row order (with input_shape): the decode then runs in worker PROCESSES
        (DecodeProcesses) that fill the staging buffers directly; without it `load_fn(plan, out)` runs on worker threads."""
        self.plan_fn, self.load_fn = plan_fn, load_fn
        self.device = torch.device(device)
        self.depth = max(2, int(depth))
        self.shape = tuple(int(v) for v in batch_shape)                # (B, H, W, 3)
        workers = workers or default_workers()
        self.pool = ThreadPoolExecutor(workers)
        gpu = self.device.type == "cuda"
        self.paths_fn, self.input_shape, self.rows_per_task, self.procs = paths_fn, input_shape, int(rows_per_task), None
        if paths_fn is not None:
            self.procs = DecodeProcesses(workers, (self.depth + 1,) + self.shape)
            self.bufs = [torch.from_numpy(self.procs.array[i]) for i in range(self.depth + 1)]
            self.pinned = False
            if gpu:                                                    # page-lock the staging file's mapping: truly asynchronous copies
                try:
                    rc = torch.cuda.cudart().cudaHostRegister(self.procs.array.ctypes.data, self.procs.array.nbytes, 0)
                    self.pinned = int(rc) == 0
                except Exception:                                      # noqa: BLE001 — pageable staging still works (synchronous copies)
                    self.pinned = False
        else:
            self.bufs = [torch.empty(self.shape, dtype=torch.uint8).pin_memory() if gpu else torch.empty(self.shape, dtype=torch.uint8)
                         for _ in range(self.depth + 1)]
        self.free = queue.Queue()
        for i in range(len(self.bufs)):
            self.free.put(i)
        self.pending = queue.Queue()                                   # (buffer index, future) in plan order
        self.copy_stream = torch.cuda.Stream(device=self.device) if gpu else None
        # device side: NDEV uint8 staging tensors in rotation; the copy of batch j + 1 is issued (side stream) while step j
        # computes, ordered behind the convert kernel that last read its staging tensor by an event — never behind the whole
        # compute stream
        self.NDEV = 3
        self.staged = [None] * self.NDEV
        self.read_done = [None] * self.NDEV                            # event: the convert kernel that read staged[k] is done
        self.busy = [None] * len(self.bufs)                            # event: the H2D copy out of pinned buffer i has finished
        self._k = 0
        self._ahead = None                                             # (staging index, copy-done event) of the NEXT batch
        self._closed = False
        for _ in range(self.depth):
            self._schedule()

    def _schedule(self):
        i = self.free.get()
        if self.busy[i] is not None:                                   # the copy that read this pinned buffer must be done
            self.busy[i].synchronize()
            self.busy[i] = None
        plan = self.plan_fn()                                          # consumer thread: the sampling stream stays ordered
        if self.procs is not None:
            paths = self.paths_fn(plan)
            futs = [self.pool.submit(self.procs.run, i, r0, paths[r0:r0 + self.rows_per_task], self.input_shape)
                    for r0 in range(0, len(paths), self.rows_per_task)]
        else:
            futs = [self.pool.submit(self.load_fn, plan, self.bufs[i].numpy())]
        self.pending.put((i, futs))

    def next_u8(self):
        """The oldest planned batch as the (pinned) uint8 host tensor — host-side use and tests; schedules its successor.  The
        returned buffer is overwritten by a later batch: copy what must outlive the next call."""
        i, futs = self.pending.get()
        for f in futs:
            f.result()
        self.free.put(i)
        self._schedule()
        return self.bufs[i]

    def _issue_copy(self):
        """Oldest decoded batch -> a device staging tensor, on the side stream.  Returns (staging index, copy-done event)."""
        i, futs = self.pending.get()
        for f in futs:
            f.result()                                                 # decode finished (raises what a worker raised)
        k = self._k = (self._k + 1) % self.NDEV
        if self.staged[k] is None:
            self.staged[k] = torch.empty(self.shape, device=self.device, dtype=torch.uint8)
        with torch.cuda.stream(self.copy_stream):
            if self.read_done[k] is not None:
                self.copy_stream.wait_event(self.read_done[k])
            self.staged[k].copy_(self.bufs[i], non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.copy_stream)
        self.busy[i] = done
        self.free.put(i)
        self._schedule()
        return k, done

    def next(self, pad_to=None):
        if self.copy_stream is None:
            raise _lib.EmbnetError("BatchPrefetcher.next() delivers device tensors: the conversion kernel has no CPU form")
        k, done = self._ahead if self._ahead is not None else self._issue_copy()
        main = torch.cuda.current_stream(self.device)
        main.wait_event(done)
        out = u8_to_f32(self.staged[k], None, self.shape[0], pad_to=pad_to)
        ev = torch.cuda.Event()
        ev.record(main)
        self.read_done[k] = ev
        self._ahead = self._issue_copy()                               # the next batch crosses PCIe while this one is stepped on
        return out

    def close(self):
        if not getattr(self, "_closed", True):      # (a half-constructed object has no _closed: nothing to release)
            self._closed = True
            self.pool.shutdown(wait=True, cancel_futures=True)
            if self.procs is not None:
                if getattr(self, "pinned", False):
                    try:
                        torch.cuda.synchronize(self.device)
                        torch.cuda.cudart().cudaHostUnregister(self.procs.array.ctypes.data)
                    except Exception:                                  # noqa: BLE001
                        pass
                self.bufs = []
                self.procs.close()

    def __del__(self):
        self.close()


class Feeder:
    """What tools/train.py steps on: `.next()` -> the next class-contiguous float32 device batch of a TripletsDataGenerator.
    In-memory float datasets (synthetic): the generator's own batch, copied; files without augmentations that fit
    `store_budget` bytes: DeviceImageStore; otherwise: BatchPrefetcher."""

    def __init__(self, gen, device, depth=10, workers=None, store_budget=32 << 30, log=None):
        self.gen, self.device = gen, torch.device(device)
        any_src = next(iter(gen.class_files_paths.values()))
        self.kind, self.store, self.prefetch = "memory", None, None
        if not isinstance(any_src, np.ndarray):
            need = DeviceImageStore.bytes_needed(gen.class_files_paths, gen.input_shape)
            if gen.augmentations is None and need <= store_budget and os.environ.get("EMBNET_IMAGE_STORE", "1") != "0":
                self.kind = "store"
                self.store = DeviceImageStore(gen.class_files_paths, gen.class_names, gen.input_shape, device, workers, log=log)
            else:
                self.kind = "prefetch"
                b = gen.k_classes * gen.k_samples
                procs = gen.augmentations is None and os.environ.get("EMBNET_DECODE_PROCESSES", "1") != "0"
                self.kind = "prefetch (worker processes)" if procs else "prefetch (worker threads)"
                self.prefetch = BatchPrefetcher(gen.sample_plan, gen.load_plan_u8, (b, gen.input_shape[1], gen.input_shape[0], 3),
                                                device, depth, workers, paths_fn=gen.plan_paths if procs else None,
                                                input_shape=gen.input_shape)
        if log:
            log(f"input pipeline: {self.kind}")

    def next(self):
        if self.store is not None:
            return self.store.batch(self.gen.sample_plan())
        if self.prefetch is not None:
            return self.prefetch.next()
        return torch.from_numpy(self.gen.sample_batch()).to(self.device)

    def close(self):
        if self.prefetch is not None:
            self.prefetch.close()
