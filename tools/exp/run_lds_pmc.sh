cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/pmc_lds
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --kernel-trace --output-format csv -d gpurun_out/pmc_lds -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timer > /dev/null 2> gpurun_out/pmc_lds.err
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/pmc_lds/*/*counter_collection.csv")
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for fn in f:
    for r in csv.DictReader(open(fn)):
        k = r["Kernel_Name"]
        if "conv_" not in k: continue
        agg[k[:90]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, d in sorted(agg.items()):
    act = d.get("SQ_LDS_IDX_ACTIVE", 0); bc = d.get("SQ_LDS_BANK_CONFLICT", 0)
    print(f"{k:90s} LDS active {act:.3e}  bank conflict {bc:.3e}  ({100*bc/max(act,1):.1f} %)  addr conflict {d.get('SQ_LDS_ADDR_CONFLICT',0):.3e}")
PY
rm -rf gpurun_out/pmc_lds
