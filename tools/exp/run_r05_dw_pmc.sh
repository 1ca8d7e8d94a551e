# where do the depthwise kernels spend their wave cycles?  SQ counters per kernel over tools/exp/time_dw.py (batch 256)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/r05_dw_pmc; rm -rf $out; mkdir -p $out
rocprofv3 -L 2>/dev/null | grep -oE "\b(TCP|TA|TD|TCC)_[A-Z0-9_]+" | sort -u > $out/counters_mem.txt
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VALU --kernel-trace --output-format csv -d $out/sq -- python3 tools/exp/time_dw.py 256 > $out/time_dw.txt 2> $out/sq.err
python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob("gpurun_out/r05_dw_pmc/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
with open("gpurun_out/r05_dw_pmc/sq_summary.txt", "w") as o:
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:14]:
        wc = max(v.get("SQ_WAVE_CYCLES", 1), 1)
        line = (f"{k:70s} n={cnt[k]:4d} wait_any {v.get('SQ_WAIT_ANY',0)/wc:5.2f} wait_inst {v.get('SQ_WAIT_INST_ANY',0)/wc:5.2f} "
                f"act_valu {v.get('SQ_ACTIVE_INST_VALU',0)/wc:5.2f} act_vmem {v.get('SQ_ACTIVE_INST_VMEM',0)/wc:5.2f} "
                f"valu/vmem_rd insts {v.get('SQ_INSTS_VALU',0)/max(v.get('SQ_INSTS_VMEM_RD',1),1):6.1f}")
        print(line); o.write(line + "\n")
PY
rm -rf $out/sq
tail -14 $out/time_dw.txt; head -60 $out/counters_mem.txt | tr '\n' ' '
