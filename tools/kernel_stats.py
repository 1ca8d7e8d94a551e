#!/usr/bin/env python3
"""rocprofv3 --kernel-trace output (…_kernel_trace.csv) -> per-kernel table (markdown + csv).

  python tools/kernel_stats.py gpurun_out/prof/…_kernel_trace.csv STEPS out_prefix "title line"

STEPS = bench steps in the trace (warm-up + timed), used for the per-step columns."""
import collections
import csv
import sys


def main():
    path, steps, prefix = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    title = sys.argv[4] if len(sys.argv) > 4 else path
    agg = collections.OrderedDict()
    skipped = 0.0
    for r in csv.DictReader(open(path)):
        if "distribution_elementwise" in r["Kernel_Name"]:       # torch.rand filling the resident batch pool at start-up: not a step
            skipped += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
            continue
        d = agg.setdefault(r["Kernel_Name"], [0, 0.0])
        d[0] += 1
        d[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    total = sum(v[1] for _, v in rows)
    with open(prefix + ".csv", "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "pct"])
        for k, (n, us) in rows:
            w.writerow([k, n, round(us, 1), round(us / n, 2), round(100 * us / total, 2)])
    with open(prefix + ".md", "w") as f:
        f.write(f"# {title}\n\nSum of kernel time: {total / steps / 1e3:.3f} ms per step over {steps} steps"
                f" (start-up torch.rand kernels of the resident batch pool left out: {skipped / 1e3:.2f} ms in all).\n\n")
        f.write("| kernel | calls/step | avg us | ms/step | % |\n|---|---|---|---|---|\n")
        for k, (n, us) in rows[:40]:
            f.write(f"| `{k[:110]}` | {n / steps:.4g} | {us / n:.1f} | {us / steps / 1e3:.3f} | {100 * us / total:.1f} |\n")
    print(open(prefix + ".md").read()[:6000])


if __name__ == "__main__":
    main()
