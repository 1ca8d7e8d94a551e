#!/usr/bin/env python3
"""Headline benchmark: training images/sec of the metric-learning hot path on N MI355X GPUs of one node.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)

Default workload = BASELINE.json configs[1] (C2): ResNet18, 224x224, triplet loss with online hardest-negative mining,
local batch 32x4 = 128, E = 256, margin 0.5.  Other configs of BASELINE.json:
  --config c1   simple2 64x64, 8x4 = 32, semihard                      (configs[0])
  --config c1s  simple (the reference's default backbone) 105x105, 8x4 = 32, semihard — the other small backbone of
                configs[0]: it cannot run at 64x64, 105x105 is its design size
  --config c3   ResNet50 224x224 Siamese ('l2' head), contrastive loss, 256 pairs (configs[2])
  --config c5   EfficientNet-B0 224x224, semihard, E = 512, local batch 64x4 = 256 (one rank of configs[4])
(C4 = C2 at --gpus 8).  A step = one pass of the hot path over one synthetic batch already resident in HBM: backbone
forward -> NxN distance matrix -> mine-and-select -> hinge -> backward -> (RCCL all-reduce of the gradient) -> optimizer
(siamese: two branch forwards -> pair distance -> contrastive loss -> backward -> optimizer).  fp32 throughout.
Rank 0 prints ONE JSON line; details go to stderr.

roofline: every kernel launch of libembnet_hip.so is timed with HIP events on the launch stream (embnet_trace_*), on
one to three of the timed steps (by the length of the timed region); the kernel with the largest total time is reported against the roofline that bounds it — for the
convolution kernels the 16-bit MFMA peak (16 x 157.3 TFLOP/s) with the 16-bit FLOP they execute (each fp32 product = 3 fp16
MFMA terms of a two-piece split in the kernels that read pre-split planes or know their operands' ranges (conv_*_h_kernel), 6 bf16
terms of an exact three-way split in the others:
3 or 6 x the algorithmic 2*M*N*K; the fp32-equivalent rate is given beside it), MFMA
fp32 (157.3 TFLOP/s) for the distance / dense GEMMs with their algorithmic FLOP, HBM (8.0 TB/s spec; 6.29 TB/s measured
copy rate also given) for the streaming kernels with their algorithmic bytes.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)



def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start one child per GPU BEFORE anything here touches the GPU (a
    process that has initialised HIP must never re-execute itself) and watch them (embeddingnet_amd/launch.py): the first
    rank that fails ends the world at once with its exit code; rank 0 prints the JSON line to the inherited stdout."""
    from embeddingnet_amd import launch
    return launch.spawn(n, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], timeout_s=3600)


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ:
    _n = next((int(a.split("=")[1]) if "=" in a else int(sys.argv[i + 2]) for i, a in enumerate(sys.argv[1:])
               if a == "--gpus" or a.startswith("--gpus=")), 1)
    if _n > 1:
        sys.exit(spawn_ranks(_n))

if __name__ == "__main__":
    # before anything touches the GPU: this rank's threads onto the cores of its GPU's NUMA node (no numactl / taskset hop)
    from embeddingnet_amd import launch as _launch
    _PIN = _launch.pin_to_gpu_numa(int(os.environ.get("LOCAL_RANK", "0"))) if int(os.environ.get("WORLD_SIZE", "1")) > 1 \
        else "affinity unchanged (one rank)"
    if os.environ.get("EMBNET_TEST_FAIL_RANK") == os.environ.get("RANK", "0") and "WORLD_SIZE" in os.environ:
        sys.exit(3)                      # test hook: this rank dies in start-up (tests/test_host_cpu.py)
else:
    _PIN = "affinity unchanged (imported)"

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

MFMA_F32_PEAK_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md, Peak FP32 (matrix)
MFMA_BF16_PEAK_TFLOPS = 16 * 157.3    # same guide, § Matrix cores: bf16 dense = 16 x the f32-input rate (~2.5 PF)
HBM_PEAK_GBPS, HBM_COPY_GBPS = 8000.0, 6290.0     # same guide: HBM3E spec / measured float4 copy
# forward MACs per 224x224 image; training FLOP = 3 passes x 2 FLOP/MAC
FWD_GMAC = {"resnet18": 1.826, "resnet50": 4.1, "efficientnet-b0": 0.39}
CONFIGS = {   # name -> argument defaults
    "c1": dict(mode="triplet", backbone="simple2", image=64, k_classes=8, k_samples=4, encodings_len=256, mining="semihard"),
    "c1s": dict(mode="triplet", backbone="simple", image=105, k_classes=8, k_samples=4, encodings_len=256, mining="semihard"),
    "c2": dict(mode="triplet", backbone="resnet18", image=224, k_classes=32, k_samples=4, encodings_len=256, mining="hardest"),
    "c3": dict(mode="siamese", backbone="resnet50", image=224, pairs=256, encodings_len=256),
    "c5": dict(mode="triplet", backbone="efficientnet-b0", image=224, k_classes=64, k_samples=4, encodings_len=512,
               mining="semihard"),
}


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


def cpu_classes_for(args):
    """Classes per CPU-baseline step: the GPU step's own P x K where the host can hold the reference step's
    activations — it trains THREE branches of up to P K (K - 1) / 2 mined triplets each (models.py:181-185), ~30 MB of fp32
    activations per 224x224 ResNet18 image — else the largest P that fits half of the free memory; --cpu-classes overrides."""
    if args.cpu_classes:
        return args.cpu_classes
    per_image = 30e6 * (args.image / 224.0) ** 2 * {"resnet50": 4.0, "efficientnet-b0": 2.5}.get(args.backbone, 1.0)
    try:
        import psutil
        free = psutil.virtual_memory().available
    except Exception:
        free = 16e9
    p = args.k_classes
    while p > 2 and 3 * p * args.k_samples * (args.k_samples - 1) / 2 * per_image > 0.5 * free:
        p //= 2
    return p


def cpu_baseline(args):
    """The oracle's reference-structured step (oracle/step.py) timed on this box's host cores, on a bounded sample:
    same backbone / resolution / K, batch reduced so that a step takes seconds."""
    import numpy as np
    from oracle.step import ReferenceStep, SiameseReferenceStep
    cores = usable_cores()
    torch.set_num_threads(cores)
    rs = np.random.RandomState(1234)
    shape = (args.image, args.image, 3)
    if args.mode == "siamese":
        b = args.cpu_pairs
        ref = SiameseReferenceStep(args.backbone, shape, args.encodings_len, lr=args.lr, optimizer=args.optimizer)
        x1, x2 = rs.rand(b, *shape).astype(np.float32), rs.rand(b, *shape).astype(np.float32)
        y = (np.arange(b) < b // 2).astype(np.float32)
        run, unique = (lambda i: ref.step(x1, x2, y)), 2 * b
        what = (f"oracle/step.py SiameseReferenceStep (2 branch fwd/bwd + contrastive + {args.optimizer}) on torch-CPU fp32, "
                f"{args.backbone} {args.image}x{args.image}, {b} pairs = {unique} images/step")
    else:
        p, k = cpu_classes_for(args), args.k_samples
        ref = ReferenceStep(args.backbone, shape, args.encodings_len, p, k, args.margin,
                            args.mining if args.mining != "batch_hard" else "hardest", lr=args.lr, optimizer=args.optimizer)
        x = rs.rand(p * k, *shape).astype(np.float32)
        trip = []
        run, unique = (lambda i: trip.append(ref.step(x, rng=np.random.RandomState(i))[1])), p * k
        what = (f"oracle/step.py ReferenceStep (P predict() calls + f64 distance matrix + Python mining loop + 3-branch "
                f"fwd/bwd + {args.optimizer}) on torch-CPU fp32, {args.backbone} {args.image}x{args.image}, batch {p}x{k}={unique} "
                f"unique images/step (the timed GPU step has {args.k_classes}x{k})")
    if args.mode != "siamese" and p > 4:
        # warm-up (allocator, thread pool) on a 4-class batch: a full-size warm-up step is 20 s of a run that should take minutes
        warm = ReferenceStep(args.backbone, shape, args.encodings_len, 4, k, args.margin,
                             args.mining if args.mining != "batch_hard" else "hardest", lr=args.lr, optimizer=args.optimizer)
        warm.step(x[:4 * k], rng=np.random.RandomState(0))
        del warm
    else:
        run(0)                                                    # warm-up (allocator, thread pool)
    t0, n = time.perf_counter(), 0
    while n < 2 or (time.perf_counter() - t0 < args.cpu_seconds and n < 50):     # at least two timed steps
        n += 1
        run(n)
    dt = (time.perf_counter() - t0) / n
    return {"value": round(unique / dt, 3), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"{what}, {n} steps of {dt:.2f} s"}


def roofline_from_trace(records, traced_steps, ms_per_step, workload, conv_terms=1, planes_terms=None):
    """records: [(kernel, ms, work, unit, bytes)] over `traced_steps` steps -> (roofline dict, per-kernel table)."""
    by = {}
    detail = os.environ.get("EMBNET_BENCH_DETAIL")       # substring of a kernel name: list its launches one by one
    if detail:
        for name, ms, work, unit, nbytes in records[: len(records) // max(traced_steps, 1)]:
            if detail in name:
                log(f"    {name[:70]:70s} {1e3 * ms:9.1f} us  {work / 1e6:10.1f} {'MFLOP' if unit == 0 else 'MB'}  "
                    f"{work / ms / (1e9 if unit == 0 else 1e6):8.1f} {'TFLOP/s' if unit == 0 else 'GB/s'}")
    for name, ms, work, unit, nbytes in records:
        d = by.setdefault(name, dict(launches=0, ms=0.0, work=0.0, bytes=0.0, unit=unit))
        d["launches"] += 1; d["ms"] += ms; d["work"] += work; d["bytes"] += nbytes
    if not by:
        return None, by
    table = sorted(by.items(), key=lambda kv: -kv[1]["ms"])
    for name, d in table[:int(os.environ.get("EMBNET_BENCH_ROWS", "14"))]:
        rate = d["work"] / d["ms"] / 1e9 if d["unit"] == 0 else d["work"] / d["ms"] / 1e6
        log(f"  {name[:104]:104s} x{d['launches'] // traced_steps:3d}/step  avg {1e3 * d['ms'] / d['launches']:8.1f} us  "
            f"{rate:8.1f} {'TFLOP/s' if d['unit'] == 0 else 'GB/s'}  {d['ms'] / traced_steps:6.3f} ms/step")
    total_ms = sum(d["ms"] for d in by.values()) / traced_steps
    log(f"  traced kernels total {total_ms:.2f} ms per traced step; an untraced step takes {ms_per_step:.2f} ms ({traced_steps} traced steps, "
        f"{sum(d['launches'] for d in by.values()) // max(traced_steps, 1)} library launches per step)")
    name, d = table[0]
    avg_us = 1e3 * d["ms"] / d["launches"]
    if d["unit"] == 0:
        achieved = d["work"] / d["ms"] / 1e9              # algorithmic (fp32) FLOP: 2 * M * N * K per launch
        terms = conv_terms if "conv_" in name else 1
        planes = planes_terms is not None and ("conv_patch_kernel" in name or "conv_wgrad_planes_kernel" in name)
        if planes:
            terms = planes_terms
        hform = "_h_kernel" in name          # gather convs given both operands' ranges (embnet_conv2d_ranges): the planes' arithmetic
        if hform:
            planes, terms = True, 3
        if terms > 1:
            # the conv kernels form every fp32 product from `terms` 16-bit MFMA terms of an operand split (include/embnet.h): the
            # matrix pipe executes terms x the algorithmic FLOP, on the bf16 / fp16 instruction -> priced against that peak
            how = ("3 v_mfma_f32_32x32x16_f16 terms of a two-piece fp16 split (22 mantissa bits, a power-of-two scale per tensor)"
                   if planes and terms == 3 else f"{terms} v_mfma_f32_32x32x16_bf16 terms of an exact three-way bf16 split")
            roof = {"bound": "mfma", "kernel": name, "achieved": round(terms * achieved, 1), "peak": MFMA_BF16_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(terms * achieved / MFMA_BF16_PEAK_TFLOPS, 4),
                    "arithmetic": f"fp32 operands and accumulation; each product = {how}; `achieved` counts the executed 16-bit MFMA FLOP",
                    "fp32_equivalent_tflops": round(achieved, 2),
                    "fp32_equivalent_over_f32_mfma_peak": round(achieved / MFMA_F32_PEAK_TFLOPS, 4),
                    "flop_per_launch": d["work"] / d["launches"]}
        else:
            roof = {"bound": "mfma", "kernel": name, "achieved": round(achieved, 2), "peak": MFMA_F32_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 4),
                    "flop_per_launch": d["work"] / d["launches"]}
    else:
        achieved = d["work"] / d["ms"] / 1e6
        roof = {"bound": "hbm", "kernel": name, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "frac_of_measured_copy_rate": round(achieved / HBM_COPY_GBPS, 4)}
    roof.update(launches=d["launches"], avg_us=round(avg_us, 1), traffic_algorithmic=round(d["bytes"] / d["launches"]),
                share_of_traced_kernel_time=round(d["ms"] / traced_steps / total_ms, 4),
                traced_kernel_ms_per_step=round(total_ms, 3))
    # measured HBM bytes per launch (PMC counters, separate rocprofv3 passes): only from a profile of THIS workload
    roof["traffic"] = None
    for tfile in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"):      # newest profile that has this workload and kernel
        tpath = os.path.join(ROOT, "profiles", tfile)
        if not os.path.exists(tpath):
            continue
        prof = json.load(open(tpath)).get(workload, {})
        # rocprofv3 prints a plain kernel with its argument list ("embnet::bn_bwd_apply4_kernel(float const*, ...)"), the
        # library's trace names it without; template instantiations carry the same text in both
        key = name if name in prof.get("kernels", {}) else next((k for k in prof.get("kernels", {}) if k.startswith(name + "(")), None)
        src = f"profiles/{tfile} [{workload}] (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this bench command)"
        if key is None and prof.get("detail"):
            # a trace label that stands for several template instantiations ("embnet::dwconv_row4_kernel" = every
            # dwconv_row4x2_kernel<KS, ST, FLIP, TW> the step launches): launch-weighted mean over the instantiations
            stem = name.split("::")[-1].split("<")[0].replace("_kernel", "")
            inst = {k: v for k, v in prof["detail"].items()
                    if k.split("(")[0].split("<")[0].split("::")[-1].startswith(stem) and v.get("launches")}
            if inst:
                n_l = sum(v["launches"] for v in inst.values())
                roof["traffic"] = round(sum(prof["kernels"][k] * v["launches"] for k, v in inst.items()) / n_l)
                roof["traffic_source"] = src + f", launch-weighted over {len(inst)} instantiations of {stem}*"
                roof["traffic_over_algorithmic"] = round(roof["traffic"] / max(roof["traffic_algorithmic"], 1), 2)
                break
        if key is not None:
            roof["traffic"] = prof["kernels"][key]
            roof["traffic_source"] = src
            roof["traffic_over_algorithmic"] = round(roof["traffic"] / max(roof["traffic_algorithmic"], 1), 2)
            break
    return roof, by


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default=os.environ.get("BCFG", "c2"), choices=sorted(CONFIGS))
    ap.add_argument("--mode", default=None, choices=["triplet", "siamese"])
    ap.add_argument("--backbone", default=None)
    ap.add_argument("--image", type=int, default=None)
    ap.add_argument("--k-classes", type=int, default=None, help="classes per GPU (local batch = k_classes*k_samples)")
    ap.add_argument("--k-samples", type=int, default=None)
    ap.add_argument("--pairs", type=int, default=None, help="siamese: pairs per GPU (first half same class)")
    ap.add_argument("--encodings-len", type=int, default=None)
    ap.add_argument("--margin", type=float, default=0.5)
    ap.add_argument("--mining", default=None, choices=["hardest", "semihard", "random_hard", "batch_hard"])
    ap.add_argument("--optimizer", default="radam", choices=["radam", "adam", "rms_prop", "sgd"])
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--no-batch-hard", action="store_true", help="skip the secondary batch-hard timing behind a 'hardest' run")
    ap.add_argument("--no-graph", action="store_true", help="triplet step: do not capture the step into a HIP graph (N = 1)")
    ap.add_argument("--force-graph", action="store_true", help="triplet step: capture the step whatever the probe would decide (A/B)")
    ap.add_argument("--cpu-classes", type=int, default=0, help="classes per CPU-baseline step (0: the GPU step's own, memory permitting)")
    ap.add_argument("--cpu-pairs", type=int, default=8)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--pool", type=int, default=0,
                    help="resident synthetic batches cycled step by step (0 = as many as fit ~10 GB of HBM, 8..1024): one "
                         "batch stepped on for hundreds of steps is memorised, mining then finds nothing and every "
                         "gradient is exactly zero — a dead problem, not a training step")
    ap.add_argument("--allow-dead", action="store_true", help="print the line even if the timed region saw loss 0 / the fallback triplet")
    ap.add_argument("--sustain-seconds", type=float, default=3.0,
                    help="after the timed region: keep stepping this long (untimed for `value`) and report the steady-state ms/step")
    args = ap.parse_args()
    for key, val in {**dict(k_classes=32, k_samples=4, pairs=256, mining="hardest"), **CONFIGS[args.config]}.items():
        if getattr(args, key, None) is None:
            setattr(args, key, val)

    from embeddingnet_amd import _lib
    from embeddingnet_amd import backbones as B
    from embeddingnet_amd.parallel import GradReducer, broadcast_model, init_distributed
    from embeddingnet_amd.train_step import TripletTrainer
    from embeddingnet_amd.utils import get_optimizer

    def pool_size(n_images):
        per = 4 * n_images * args.image * args.image * 3
        return args.pool if args.pool > 0 else int(min(1024, max(8, (10 << 30) // per)))

    rank, world, local = init_distributed()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback)")
    _lib.lib()
    dev = torch.device("cuda", local % max(torch.cuda.device_count(), 1))     # (more ranks than GPUs only in the gloo debug mode)
    torch.cuda.set_device(dev)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    shape = (args.image, args.image, 3)

    if args.mode == "siamese":
        from embeddingnet_amd.losses_and_accuracies import contrastive_loss
        from embeddingnet_amd.models import SiameseNet
        net = SiameseNet({"model": dict(input_shape=list(shape), encodings_len=args.encodings_len, mode="siamese",
                                        distance_type="l2", backbone_name=args.backbone, backbone_weights=None,
                                        freeze_backbone=False, embeddings_normalization=True, device=dev, seed=0),
                          "dataloader": {}, "generator": {}, "train": {},
                          "general": {"work_dir": "work_dirs/", "project_name": "bench"}}, training=True)
        model = net.model
        params = [p for p in net.base_model.parameters() if p.requires_grad]      # the 'l2' head has no weights of its own
        n_local = 2 * args.pairs
        pool = pool_size(n_local)
        x1s = [torch.rand((args.pairs,) + shape, generator=gen, device=dev) for _ in range(pool)]       # resident in HBM
        x2s = [torch.rand((args.pairs,) + shape, generator=gen, device=dev) for _ in range(pool)]
        y = (torch.arange(args.pairs, device=dev) < args.pairs // 2).float().reshape(-1, 1)   # datagenerators.py:345-374
    else:
        model, _ = B.get_backbone(shape, encodings_len=args.encodings_len, backbone_name=args.backbone,
                                  backbone_weights=None, seed=0, device=dev)
        params = [p for p in model.parameters() if p.requires_grad]
        n_local = args.k_classes * args.k_samples
        pool = pool_size(n_local)
        batches = [torch.rand((n_local,) + shape, generator=gen, device=dev) for _ in range(pool)]     # resident in HBM
    broadcast_model(model)                                    # identical start on every rank (parameters + BN buffers)
    opt = get_optimizer(args.optimizer, args.lr).build(params)
    reducer = GradReducer(params) if world > 1 else None

    tick = [0]                       # steps taken so far: step i runs on resident batch i mod pool
    graph = (False if args.no_graph else True if args.force_graph else "auto")
    if args.mode == "siamese":
        from embeddingnet_amd.train_step import SiameseTrainer
        model.train()
        # the step as a trainer (its own step context, one-launch optimizer + planes refresh, HIP graph where the probe says the
        # host cannot keep up: ResNet50's 900 launches take the host ~70-80 ms to enqueue)
        trainer = SiameseTrainer(model, opt, contrastive_loss, seed=rank, reducer=reducer, graph=graph)

        def step():
            x1, x2 = x1s[tick[0] % pool], x2s[tick[0] % pool]
            tick[0] += 1
            return trainer.step(x1, x2, y)
    else:
        # one GPU: the step is captured into a HIP graph after its warm-up steps (the host needs 2-4 ms to enqueue a step;
        # the small configs are otherwise host-bound); steps whose kernels are being timed run eagerly
        trainer = TripletTrainer(model, opt, args.k_classes, args.k_samples, margin=args.margin,
                                 negatives_selection_mode=args.mining, seed=rank, reducer=reducer,
                                 graph=graph)     # N > 1: two graphs around the all-reduce, every
        # rank takes the same decision (TripletTrainer._agree)
        def step():
            tick[0] += 1
            return trainer.step(batches[(tick[0] - 1) % pool])

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    n_warm = args.warmup
    if getattr(trainer, "graph_mode", False):
        n_warm = max(n_warm, trainer.GRAPH_WARMUP + 8)     # the probe + capture (once, after GRAPH_WARMUP eager steps) stay untimed
    for _ in range(n_warm):
        step()
    # (round 3 had an untimed "settle" phase here: it was the single resident batch being memorised — loss 0, all-zero
    # gradients, a cooler chip — not a clock ramp.  On live gradients it changes nothing: profiles/r04_ab_bench_c2_live*.json,
    # 11.065 ms/step with it, 11.05 without.  Removed.)
    trace = rank == 0 and not args.no_kernel_timer
    # kernels are timed on a FEW of the timed steps: a traced step carries two HIP events per launch (~600 marker packets on
    # a ResNet18 step, each a small bubble on the queue) and runs eagerly even where the step is a HIP graph, so it is slower
    # than a plain one — at every 4th step that cost 12 % of `value` (r03: 11.65 ms timed vs 10.19 ms in the untraced leg).
    # A traced C2 step costs ~1.4 ms more than a plain one (13 121 images/s timed with three of 50 steps traced vs 13 235 in
    # the untraced leg), so the number of traced steps follows the length of the timed region: 1 below 40 steps, 2 below
    # 80, else 3 — spread evenly, never the first step.
    # Round 6: never fewer than three traced steps in all (a per-kernel average over ONE step is one box's one moment)
    # — but a traced step is slower than a plain one (+ ~1.4 ms on C2; on a graph-replayed config it runs eagerly: 2.0 vs 0.9 ms on
    # C1), so only as many of them sit inside the timed region as its length carries (1 below 40 steps, 2 below 80, else 3) and the
    # rest are taken in the sustained leg behind it, which is not part of `value`.
    n_traced = min(1 if args.steps < 40 else 2 if args.steps < 80 else 3, max(args.steps - 1, 1))
    traced_at = {(j + 1) * args.steps // (n_traced + 1) for j in range(n_traced)}
    n_traced_sustained = max(0, 3 - n_traced) if args.sustain_seconds > 0 else 0
    if trace:
        _lib.trace_reset()
    # liveness record of the timed region, kept on the device (no host sync inside a step): every step's loss and number
    # of mined triplets — two 4-byte device copies per step
    loss_log = torch.zeros(args.steps, device=dev)
    count_log = torch.zeros(args.steps, device=dev, dtype=torch.int32)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if trace:
            _lib.trace_enable(i in traced_at)
        loss = step()
        loss_log[i].copy_(loss)
        if args.mode != "siamese":
            count_log[i].copy_(trainer.last_triplets[1][0])
    host_ms = 1e3 * (time.perf_counter() - t0) / args.steps       # time to ENQUEUE a step (the host runs ahead of the GPU)
    barrier()
    elapsed = time.perf_counter() - t0
    _lib.trace_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / args.steps
    value = n_local * world * args.steps / elapsed
    # sustained leg (not part of `value`): the chip lowers its clock under sustained matrix load, so a 0.2-0.6 s timed
    # region is a short-burst number; keep stepping and report what the step time settles at
    sustained, sus_traced = None, set()
    if args.sustain_seconds > 0:
        n_sus = max(args.steps, int(args.sustain_seconds * 1e3 / max(ms_per_step, 1e-3)) + 1)
        sus_traced = {(j + 1) * n_sus // (n_traced_sustained + 1) for j in range(n_traced_sustained)} if trace else set()
        barrier()
        s0 = time.perf_counter()
        for i in range(n_sus):
            if sus_traced:
                _lib.trace_enable(i in sus_traced)
            step()
        _lib.trace_enable(False)
        barrier()
        sus = time.perf_counter() - s0
        if world > 1:
            t = torch.tensor([sus], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            sus = float(t.item())
        sustained = {"steps": n_sus, "seconds": round(sus, 3), "ms_per_step": round(1e3 * sus / n_sus, 3),
                     "images_per_sec": round(n_local * world * n_sus / sus, 2)}
    # BASELINE.json's metric string says "triplet batch-hard": the headline line runs the reference's own rule for it ('hardest',
    # datagenerators.py:188-190 — SURVEY's C2); the Hermans batch-hard rule (README.md:112 of the reference cites the paper) is
    # timed here in the SAME process on the same model and batches, as a secondary value — never `value`.
    batch_hard = None
    if args.mode == "triplet" and args.mining == "hardest" and not args.no_batch_hard:
        tr2 = TripletTrainer(model, opt, args.k_classes, args.k_samples, margin=args.margin, negatives_selection_mode="batch_hard",
                             seed=rank, reducer=reducer, graph=graph)
        n2 = max(args.steps, 20)
        for _ in range(max(5, (tr2.GRAPH_WARMUP + 8) if tr2.graph_mode else 0)):
            tick[0] += 1
            tr2.step(batches[(tick[0] - 1) % pool])
        barrier()
        b0 = time.perf_counter()
        for _ in range(n2):
            tick[0] += 1
            bh_loss = tr2.step(batches[(tick[0] - 1) % pool])
        barrier()
        bh = time.perf_counter() - b0
        if world > 1:
            t = torch.tensor([bh], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            bh = float(t.item())
        batch_hard = {"mining": "batch_hard (Hermans et al.: hardest positive and hardest negative per anchor)", "steps": n2,
                      "ms_per_step": round(1e3 * bh / n2, 3), "images_per_sec": round(n_local * world * n2 / bh, 2),
                      "step_mode": "hip graph" if getattr(tr2, "_graph", None) is not None else "eager",
                      "last_loss": float(bh_loss.item()), "active_triplets_last": int(tr2.last_triplets[1][0].item())}
    host_all = [host_ms]
    if world > 1:                                             # every rank's host enqueue time (8 Python launch threads share the node's cores)
        t = torch.zeros(world, device=dev, dtype=torch.float64)
        t[rank] = host_ms
        dist.all_reduce(t)
        host_all = [round(v, 3) for v in t.tolist()]
    # a dead problem (hinge 0 on the reference's fallback triplet, datagenerators.py:246-250: every gradient exactly zero)
    # is not a training step: refuse to report a throughput for it — on every rank (each mines its own shard)
    losses, counts = loss_log.tolist(), count_log.tolist()
    # (isolated dead steps — semihard mining on a batch where no negative falls inside the margin band — happen in live training
    # too and run the same kernels: the run is refused when more than one step in twenty is dead)
    dead_steps = sum(1 for i, l in enumerate(losses) if l <= 0.0 or (args.mode != "siamese" and counts[i] <= 1))
    live = dict(loss_first_timed=losses[0], loss_last_timed=losses[-1], loss_min_timed=min(losses),
                active_triplets_min=(min(counts) if args.mode != "siamese" else None),
                active_triplets_mean=(round(sum(counts) / len(counts), 1) if args.mode != "siamese" else None),
                dead_steps=dead_steps, resident_batches=pool)
    dead = dead_steps > len(losses) // 20
    if world > 1:
        t = torch.tensor([1.0 if dead else 0.0], device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dead = t.item() > 0.5
    # per-rank record for rank 0's log: device, core affinity, step mode, how the gradient mean is taken
    mine = {"rank": rank, "device": f"{torch.cuda.get_device_name(dev)} [{dev.index}]", "affinity": _PIN,
            "step_mode": ("hip graph" if getattr(trainer, "_graph", None) is not None else "eager"),
            "gradient_mean": reducer.mean_mode if reducer is not None else None, "host_enqueue_ms": round(host_ms, 3)}
    ranks_info = [mine]
    if world > 1:
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, mine)
    if dead and not args.allow_dead:
        raise SystemExit(f"bench: the timed region ran on a dead problem ({live}): loss 0 / fallback triplet means all-zero "
                         "gradients; no throughput is reported for it (raise --pool, or --allow-dead for an A/B)")
    if rank != 0:
        return

    if world > 1:
        for r in ranks_info:
            log(f"  rank {r['rank']}: {r['device']}, {r['affinity']}, step {r['step_mode']}, gradient mean: {r['gradient_mean']}, "
                f"host enqueue {r['host_enqueue_ms']} ms/step")
    log(f"  step mode: {'HIP graph replay' if getattr(trainer, '_graph', None) is not None else 'eager'}"
        + (f" (capture failed: {trainer._graph_error})" if getattr(trainer, "_graph_failed", False) else "")
        + (f"  probe: {trainer.graph_probe}" if getattr(trainer, "graph_probe", None) else ""))
    # the enqueue loop blocks once the HIP queue is full, so on a GPU-bound step its time approaches the step time; the host
    # WORK per step is what the trainer's probe measured on an idle queue (graph_probe['host_ms'], triplet mode)
    replayed = getattr(trainer, "_graph", None) is not None      # a replayed step has no per-kernel host work
    probe_host = (getattr(trainer, "graph_probe", None) or {}).get("host_ms") if not replayed else None
    host_work = probe_host if probe_host is not None else host_ms
    log(f"  enqueue loop {host_ms:.2f} ms/step" + (f", host work {probe_host:.2f} ms/step (probe, idle queue)" if probe_host is not None else "")
        + f", step {ms_per_step:.2f} ms ({'GPU' if host_work < 0.9 * ms_per_step else 'host'}-bound)")
    roofline = None
    if trace:
        roofline, _ = roofline_from_trace(_lib.trace_records(), len(traced_at) + (len(sus_traced) if sustained else 0), ms_per_step, args.config,
                                          _lib.lib().embnet_conv_mfma_terms(), _lib.lib().embnet_conv_planes_mfma_terms())
        if roofline is not None:
            roofline["traced_steps"] = {"in_timed_region": len(traced_at), "in_sustained_leg": len(sus_traced) if sustained else 0}
        if roofline is not None and args.backbone in FWD_GMAC and args.image == 224:
            # the whole step's algorithmic FLOP against (a) the fp32-MFMA ceiling SURVEY 8(d) names — it can exceed 1: the convs
            # execute as 16-bit MFMA terms — and (b) the peak of the instruction they execute on, x the terms executed per product
            # (3 on the ResNets' default path): the fraction of the chip's matrix rate the STEP uses
            e2e = value / world * 6 * FWD_GMAC[args.backbone] * 1e9
            roofline["end_to_end_frac_of_mfma_peak"] = round(e2e / (MFMA_F32_PEAK_TFLOPS * 1e12), 4)
            terms = _lib.lib().embnet_conv_planes_mfma_terms() if args.backbone.startswith("resnet") else _lib.lib().embnet_conv_mfma_terms()
            roofline["end_to_end_executed_terms"] = terms
            roofline["end_to_end_frac_of_executed_peak"] = round(terms * e2e / (MFMA_BF16_PEAK_TFLOPS * 1e12), 4)
    SPLIT_NOTE = ("3x3 stride-1 convs and their weight gradients, and the ResNets' other convs where both operands' ranges are known: two fp16 "
                  "pieces + a per-tensor power-of-two scale, 3 terms; other convs: "
                  "three bf16 pieces, 6 terms" if _lib.lib().embnet_conv_planes_mfma_terms() == 3 else "three exact bf16 pieces, 6 terms")
    if args.mode == "siamese":
        metric = f"images/sec training ({args.backbone}, {args.image}², Siamese contrastive) @ 1/2/4/8 GPU"
        workload = (f"{args.backbone} {args.image}x{args.image}x3 synthetic U[0,1), SiameseNet 'l2' head + contrastive_loss, "
                    f"{args.pairs} pairs per GPU (first half same class) = {n_local} images, {pool} resident batches cycled, E={args.encodings_len}, "
                    f"{args.optimizer}, fp32 tensors (conv products on 16-bit MFMA from operand splits, fp32 accumulate: " + SPLIT_NOTE + ")")
    else:
        label = {"resnet18": "ResNet18", "resnet50": "ResNet50", "efficientnet-b0": "EfficientNet-B0"}.get(args.backbone, args.backbone)
        # (BASELINE.json's metric string says "batch-hard"; the reference's own rule for it is 'hardest' — config.mining and
        # config.workload name the rule that actually ran)
        mining = "batch-hard" if (args.mining in ("hardest", "batch_hard")) else args.mining
        metric = f"images/sec training ({label}, {args.image}², triplet {mining}) @ 1/2/4/8 GPU"
        workload = (f"{args.backbone} {args.image}x{args.image}x3 synthetic U[0,1), 107-class P x K sampling, local batch "
                    f"{args.k_classes}x{args.k_samples}={n_local} ({pool} resident batches cycled), E={args.encodings_len}, margin {args.margin}, mining "
                    f"'{args.mining}' per local batch, {args.optimizer}, fp32 tensors (conv products on 16-bit MFMA from operand splits, fp32 accumulate: " + SPLIT_NOTE + ")")
    out = {
        "metric": metric, "value": round(value, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": workload, "baseline_config": args.config, "global_batch": n_local * world,
                   "parallelism": f"dp{world}", "final_loss": float(loss.item()), **live,
                   "host_enqueue_ms_per_step": round(host_ms, 3), "host_enqueue_ms_per_step_by_rank": host_all,
                   "step_mode": ("hip graph" if getattr(trainer, "_graph", None) is not None else "eager"),
                   "host_work_ms_per_step": (round(probe_host, 3) if probe_host is not None else None),
                   "graph_probe": getattr(trainer, "graph_probe", None),
                   "mining": getattr(args, "mining", None) if args.mode != "siamese" else None,
                   "ranks": ranks_info if world > 1 else None,
                   "sustained": sustained,
                   "batch_hard": batch_hard,
                   "sustained_ms_per_step": sustained["ms_per_step"] if sustained else None,
                   "sustained_over_timed": round(sustained["ms_per_step"] / ms_per_step, 4) if sustained else None},
        "roofline": roofline,
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
    if dist.is_initialized():
        dist.destroy_process_group()
