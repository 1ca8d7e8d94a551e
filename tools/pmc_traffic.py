#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) into profiles/pmc_traffic.json:
{kernel name: HBM bytes per launch}, with the gfx950 corrections of MI355X_MICROARCH.md §HBM:
FETCH_SIZE counts 64 B per 128-B request of a wide (16 B/lane) streaming read -> doubled for the
kernels whose operand loads are all 16 B/lane (the VEC conv/GEMM kernels); counter unit is KiB.

  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py ...
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py ...
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/pmc_traffic.json
"""
import collections
import csv
import glob
import json
import sys


def per_kernel(dirname, counter):
    out = collections.defaultdict(list)
    for f in glob.glob(f"{dirname}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                out[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in out.items()}, {k: len(v) for k, v in out.items()}


def main():
    fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
    write, _ = per_kernel(sys.argv[2], "WRITE_SIZE")
    res, detail = {}, {}
    for k in sorted(set(fetch) | set(write)):
        wide = ("embnet::conv_" in k or "embnet::pairwise" in k or "embnet::dense" in k) and ", false" not in k
        f = fetch.get(k, 0.0) * 1024 * (2 if wide else 1)
        w = write.get(k, 0.0) * 1024
        res[k] = round(f + w)
        detail[k] = dict(fetch_bytes=round(f), write_bytes=round(w), fetch_doubled=wide, launches=nf.get(k, 0))
    json.dump(res, open(sys.argv[3], "w"), indent=1)
    json.dump(detail, open(sys.argv[3].replace(".json", "_detail.json"), "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1])[:12]:
        print(f"{v / 1e6:10.1f} MB/launch  {k[:110]}")


if __name__ == "__main__":
    main()
