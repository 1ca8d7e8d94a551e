"""Sum PMC counters per kernel name from rocprofv3 counter_collection csv dirs: python pmc_sum.py <dir> <substr>"""
import csv, glob, sys, collections
d, sub = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(list)
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("  ".join(f"{k}={sum(v)/len(v):.4g}" for k, v in sorted(agg.items())))
