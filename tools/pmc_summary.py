"""Summarise rocprofv3 --pmc counter_collection.csv files for the trace kernel (largest dispatches)."""
import csv, glob, sys, collections
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        per = collections.defaultdict(dict)
        for r in rows:
            if "trace_kernel" in r["Kernel_Name"] or "render_kernel" in r["Kernel_Name"]:
                k = r["Dispatch_Id"]
                per[k][r["Counter_Name"]] = float(r["Counter_Value"])
                per[k]["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
                per[k]["vgpr"] = r["VGPR_Count"]; per[k]["grid"] = r["Grid_Size"]
        if not per: continue
        best = max(per.values(), key=lambda v: v["ms"])
        print(f, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in best.items()})
