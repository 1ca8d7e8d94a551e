import csv, glob, sys, collections
d = sys.argv[1]
cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(cc)):
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    agg[k]["dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in agg.items():
    if "GRBM_GUI_ACTIVE" not in v or max(v["dur_ns"]) < 1e6: continue
    g = sum(v["GRBM_GUI_ACTIVE"][2:]) / len(v["GRBM_GUI_ACTIVE"][2:]); dur = sum(v["dur_ns"][2:]) / len(v["dur_ns"][2:])
    m = sum(v["SQ_VALU_MFMA_BUSY_CYCLES"][2:]) / len(v["SQ_VALU_MFMA_BUSY_CYCLES"][2:])
    clk = g / 8 / dur
    print(f"{k:70s} dur {dur/1e3:9.1f} us  clock {clk:5.2f} GHz  MFMA busy {m / 1024 / (g / 8):6.1%} of cycles")
