#!/usr/bin/env python3
"""Headline benchmark: training images/sec of the fused metric-learning step
(ResNet18, 224x224, triplet loss with online hardest-negative mining, local batch 32x4=128,
E=256, margin 0.5 — BASELINE.json configs[1]) on N MI355X GPUs of one node.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

A step = one pass of the hot path over one synthetic batch already resident in HBM: backbone
forward -> NxN distance matrix -> mine-and-select -> hinge -> backward -> (RCCL all-reduce of the
gradient) -> RAdam update.  fp32 throughout.  Rank 0 prints ONE JSON line; details go to stderr.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md, Peak FP32 (matrix)
# forward MACs per 224x224 image; training FLOP = 3 passes x 2 FLOP/MAC
FWD_GMAC = {"resnet18": 1.826, "resnet50": 4.1, "simple2": 0.136, "simple": 0.991}


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota
    (os.cpu_count() reports the whole machine and oversubscribes the thread pool)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(args):
    """The oracle's reference-structured step (oracle/step.py) timed on this box's host cores, on a
    bounded sample: same backbone / resolution / K, P reduced so that a step takes seconds."""
    import numpy as np
    from oracle.step import ReferenceStep
    cores = usable_cores()
    torch.set_num_threads(cores)
    p, k = args.cpu_classes, args.k_samples
    ref = ReferenceStep(args.backbone, (args.image, args.image, 3), args.encodings_len, p, k, args.margin,
                        args.mining if args.mining != "batch_hard" else "hardest", lr=args.lr, seed=0,
                        optimizer="radam")
    rs = np.random.RandomState(1234)
    x = rs.rand(p * k, args.image, args.image, 3).astype(np.float32)
    ref.step(x, rng=np.random.RandomState(0))                    # warm-up (allocator, thread pool)
    t0, n, trip = time.perf_counter(), 0, 0
    while n < 1 or (time.perf_counter() - t0 < args.cpu_seconds and n < 50):
        _, t, _ = ref.step(x, rng=np.random.RandomState(n))
        n += 1
        trip += t
    dt = (time.perf_counter() - t0) / n
    return {"value": round(p * k / dt, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"oracle/step.py ReferenceStep (P predict() calls + f64 distance matrix + Python mining loop + "
                      f"3-branch fwd/bwd + RAdam) on torch-CPU fp32, {args.backbone} {args.image}x{args.image}, "
                      f"batch {p}x{k}={p * k} unique images/step, {n} steps of {dt:.2f} s, "
                      f"mean {trip / n:.1f} triplets (= {3 * trip / n / dt:.2f} branch images/sec)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--backbone", default="resnet18")
    ap.add_argument("--image", type=int, default=224)
    ap.add_argument("--k-classes", type=int, default=32, help="classes per GPU (local batch = k_classes*k_samples)")
    ap.add_argument("--k-samples", type=int, default=4)
    ap.add_argument("--encodings-len", type=int, default=256)
    ap.add_argument("--margin", type=float, default=0.5)
    ap.add_argument("--mining", default="hardest", choices=["hardest", "semihard", "random_hard", "batch_hard"])
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--cpu-classes", type=int, default=4)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    from embeddingnet_amd import _lib
    from embeddingnet_amd import backbones as B
    from embeddingnet_amd import layers as L
    from embeddingnet_amd.parallel import GradReducer, init_distributed
    from embeddingnet_amd.train_step import TripletTrainer
    from embeddingnet_amd.utils import get_optimizer

    rank, world, local = init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    _lib.lib()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    base, _ = B.get_backbone((args.image, args.image, 3), encodings_len=args.encodings_len,
                             backbone_name=args.backbone, backbone_weights=None, seed=0, device=dev)
    opt = get_optimizer("radam", args.lr).build([p for p in base.parameters() if p.requires_grad])
    reducer = GradReducer(base.parameters()) if world > 1 else None
    trainer = TripletTrainer(base, opt, args.k_classes, args.k_samples, margin=args.margin,
                             negatives_selection_mode=args.mining, seed=rank, reducer=reducer)
    n_local = args.k_classes * args.k_samples
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    images = torch.rand((n_local, args.image, args.image, 3), generator=gen, device=dev)   # resident in HBM

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step(images)
    timer = None if args.no_kernel_timer else L.KernelTimer()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        # HIP-event brackets around the conv kernels on every 4th timed step: each bracket is two marker
        # packets on the stream, and ~120 of them per step cost ~8 % of the step if applied to all steps
        L.TIMER = timer if i % 4 == 0 else None
        loss = trainer.step(images)
    barrier()
    elapsed = time.perf_counter() - t0
    L.TIMER = None
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = n_local * world * args.steps / elapsed

    if rank != 0:
        return
    roofline = None
    if timer is not None:
        summ = timer.summary()
        for name, d in sorted(summ.items(), key=lambda kv: -kv[1]["ms_total"]):
            log(f"  {name:100s} launches {d['launches']:5d}  avg {1e3 * d['ms_total'] / d['launches']:9.1f} us  "
                f"{d['flops_total'] / d['ms_total'] / 1e9:7.1f} TFLOP/s")
        if os.environ.get("EMBNET_BENCH_DETAIL"):
            for name, d in sorted(timer.detail().items(), key=lambda kv: -kv[1]["ms_total"]):
                log(f"    {name:86s} x{d['launches']:3d}  avg {1e3 * d['ms_total'] / d['launches']:8.1f} us  "
                    f"{d['flops_total'] / d['ms_total'] / 1e9:6.1f} TF/s")
        conv_ms = sum(d["ms_total"] for d in summ.values())
        timed_steps = (args.steps + 3) // 4
        log(f"  conv kernels total {conv_ms / timed_steps:.2f} ms of {ms_per_step:.2f} ms per step "
            f"(brackets on {timed_steps} of {args.steps} steps)")
        name, d = max(summ.items(), key=lambda kv: kv[1]["ms_total"])
        achieved = d["flops_total"] / d["ms_total"] / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath):
            traffic = json.load(open(tpath)).get(name)
        roofline = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": MFMA_F32_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 4), "traffic": traffic,
                    "launches": d["launches"], "avg_us": round(1e3 * d["ms_total"] / d["launches"], 1),
                    "flop_per_launch": d["flops_total"] / d["launches"]}
        if args.backbone in FWD_GMAC and args.image == 224:
            roofline["end_to_end_frac"] = round(value / world * 6 * FWD_GMAC[args.backbone] * 1e9 /
                                                (MFMA_F32_PEAK_TFLOPS * 1e12), 4)
    out = {
        "metric": "images/sec training (ResNet18, 224², triplet batch-hard) @ 1/2/4/8 GPU",
        "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.backbone} {args.image}x{args.image}x3 synthetic U[0,1), 107-class P x K sampling, "
                               f"local batch {args.k_classes}x{args.k_samples}={n_local}, E={args.encodings_len}, "
                               f"margin {args.margin}, mining '{args.mining}' per local batch, RAdam, fp32",
                   "global_batch": n_local * world, "parallelism": f"dp{world}", "final_loss": float(loss.item())},
        "roofline": roofline,
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
    if dist.is_initialized():
        dist.destroy_process_group()
