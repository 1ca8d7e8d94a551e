cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/sep -- python3 $GRAFT_REPO_ROOT/tools/exp/se_mlp_bench.py > /dev/null 2>&1
python3 - <<'PY'
import csv,glob,collections
fs=glob.glob("/tmp/sep/**/*kernel_trace.csv",recursive=True)
rows=list(csv.DictReader(open(fs[0])))
agg=collections.OrderedDict()
for r in rows:
    nm=r["Kernel_Name"]
    if "semlp" not in nm: continue
    key=(nm.split("(")[0].split("::")[-1][:22], r.get("Grid_Size_X") or r.get("Workgroup_Size_X"))
    d=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
    agg.setdefault(key,[]).append(d)
for k,v in agg.items():
    v=sorted(v); print(k, len(v), "median", round(v[len(v)//2],1))
PY
