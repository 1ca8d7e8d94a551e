#!/usr/bin/env python3
"""Two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of one bench command -> HBM bytes per launch per kernel, merged
into profiles/rNN_pmc_traffic.json under the bench config's name ({"c2": {"command": ..., "kernels": {name: bytes}}}),
with the gfx950 corrections of MI355X_MICROARCH.md §HBM: the counters are in KiB; FETCH_SIZE tallies a 128-B request
of a wide (16 B/lane) streaming read as 64 B, so it is doubled for the kernels whose loads are all 16 B/lane — the VEC
conv / GEMM / dense kernels and every `*4_kernel` (BatchNorm, pooling, GAP, depthwise, SE: float4 lanes).  WRITE_SIZE
is exact for 16-B-per-lane stores.  Other access widths are uncalibrated (left as counted).

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/r02_pmc_traffic.json c2 "<command>"
"""
import collections
import csv
import glob
import json
import os
import re
import sys


def per_kernel(dirname, counter):
    out = collections.defaultdict(list)
    for f in glob.glob(f"{dirname}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                out[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in out.items()}, {k: len(v) for k, v in out.items()}


def wide_loads(kernel):
    """True for kernels whose global loads are all 16 bytes per lane."""
    if re.search(r"embnet::(conv_|pairwise|cross_dist|dense_)", kernel):
        return ", false" not in kernel                    # the non-VEC instantiations load dwords
    return bool(re.search(r"embnet::\w*4(_sq)?_kernel", kernel)) or "slab_reduce" in kernel or "opt_step" in kernel \
        or "tail_fixup" in kernel or "affine_act_kernel" in kernel or "embnet::patch::" in kernel or "_planes_kernel" in kernel


def main():
    fetch_dir, write_dir, out_path, workload = sys.argv[1:5]
    command = sys.argv[5] if len(sys.argv) > 5 else ""
    fetch, nf = per_kernel(fetch_dir, "FETCH_SIZE")
    write, _ = per_kernel(write_dir, "WRITE_SIZE")
    kernels, detail = {}, {}
    for k in sorted(set(fetch) | set(write)):
        wide = wide_loads(k)
        f = fetch.get(k, 0.0) * 1024 * (2 if wide else 1)
        w = write.get(k, 0.0) * 1024
        kernels[k] = round(f + w)
        detail[k] = dict(fetch_bytes=round(f), write_bytes=round(w), fetch_doubled=wide, launches=nf.get(k, 0))
    merged = json.load(open(out_path)) if os.path.exists(out_path) else {}
    merged[workload] = dict(command=command, kernels=kernels, detail=detail)
    json.dump(merged, open(out_path, "w"), indent=1)
    for k, v in sorted(kernels.items(), key=lambda kv: -kv[1])[:12]:
        print(f"{v / 1e6:10.1f} MB/launch  {k[:110]}")


if __name__ == "__main__":
    main()
