#!/usr/bin/env python3
"""rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace of a bench command ->
per-kernel shader clock and matrix-pipe utilisation (markdown).  Units (MI355X_MICROARCH.md): GRBM_GUI_ACTIVE is summed
over the 8 XCDs, so clock = GRBM_GUI_ACTIVE / 8 / duration; SQ_VALU_MFMA_BUSY_CYCLES counts cycles a SIMD's matrix pipe is
busy, summed over the 1024 SIMDs, so MFMA utilisation = that / 1024 / (GRBM_GUI_ACTIVE / 8).

  python tools/pmc_mfma_clock.py gpurun_out/pmc_mfma profiles/r02_pmc_mfma_clock.md "title"
"""
import collections
import csv
import glob
import sys


def main():
    d, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else d
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                agg[k]["dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    rows = []
    for k, v in agg.items():
        if "GRBM_GUI_ACTIVE" not in v or not v["dur_ns"]:
            continue
        n = len(v["dur_ns"])
        g, dur = sum(v["GRBM_GUI_ACTIVE"]) / n, sum(v["dur_ns"]) / n
        m = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])) / max(len(v.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])), 1)
        rows.append((dur * n, k, n, dur, g / 8 / dur, m / 1024 / max(g / 8, 1)))
    rows.sort(reverse=True)
    with open(out, "w") as f:
        f.write(f"# {title}\n\nclock = GRBM_GUI_ACTIVE / 8 XCDs / duration (reads high on launches shorter than ~0.3 ms: the counter "
                "window is wider than the kernel); MFMA util = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / cycles.\n\n"
                "| kernel | launches | avg us | clock GHz | MFMA pipe busy |\n|---|---|---|---|---|\n")
        for _, k, n, dur, clk, util in rows[:30]:
            f.write(f"| `{k[:110]}` | {n} | {dur / 1e3:.1f} | {clk:.2f} | {util:.1%} |\n")
    print(open(out).read()[:4000])


if __name__ == "__main__":
    main()
