"""Summary of tools/pmc_quick.sh: per arm, the counters of the LONG dispatches of the trace kernels (mean per launch)."""
import collections, csv, glob, json, os, sys
out = sys.argv[1]
res = {}
for arm in sorted(glob.glob(f"{out}/arm*")):
    if not os.path.isdir(arm): continue
    per = collections.defaultdict(dict)
    for f in glob.glob(f"{arm}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "trace_" in r["Kernel_Name"] and "Lb1E" not in r["Kernel_Name"] and "<true" not in r["Kernel_Name"]:
                e = per[(f, r["Dispatch_Id"])]
                e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                e["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
                e["kernel"] = r["Kernel_Name"][:60]; e["vgpr"] = r.get("VGPR_Count"); e["lds"] = r.get("LDS_Block_Size")
    ds = list(per.values())
    if not ds: continue
    longest = max(x["ms"] for x in ds)
    ds = [x for x in ds if x["ms"] >= 0.5 * longest]
    agg = collections.defaultdict(list)
    for x in ds:
        for k, v in x.items():
            if isinstance(v, float): agg[k].append(v)
    m = {k: sum(v) / len(v) for k, v in agg.items()}
    m["kernel"] = ds[0]["kernel"]; m["vgpr"] = ds[0]["vgpr"]; m["lds"] = ds[0]["lds"]
    m["opts"] = open(f"{arm}/opts.txt").read().strip() if os.path.exists(f"{arm}/opts.txt") else ""
    if "SQ_THREAD_CYCLES_VALU" in m and "SQ_INSTS_VALU" in m: m["lane_utilisation"] = m["SQ_THREAD_CYCLES_VALU"] / (64.0 * m["SQ_INSTS_VALU"])
    if "TCC_HIT_sum" in m: m["l2_hit_rate"] = m["TCC_HIT_sum"] / max(1.0, m["TCC_HIT_sum"] + m["TCC_MISS_sum"])
    if "SQ_WAIT_INST_ANY" in m and "SQ_WAVE_CYCLES" in m: m["wave_life_in_s_waitcnt"] = m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]
    if "FETCH_SIZE" in m: m["l2_miss_traffic_GB"] = (m["FETCH_SIZE"] * 2 * 64 + m.get("WRITE_SIZE", 0) * 64) / 1e9 if m["FETCH_SIZE"] > 1e3 else None
    res[os.path.basename(arm)] = m
keys = ["opts", "kernel", "vgpr", "lds", "ms", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "lane_utilisation", "wave_life_in_s_waitcnt",
        "TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum", "l2_hit_rate", "FETCH_SIZE", "WRITE_SIZE", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "TA_TA_BUSY_sum", "TCP_PENDING_STALL_CYCLES_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum", "GRBM_GUI_ACTIVE", "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum"]
for k in keys:
    print(f"{k:24s}", *[(f"{res[a].get(k):>16.4g}" if isinstance(res[a].get(k), float) else f"{str(res[a].get(k))[:16]:>16s}") for a in res])
json.dump(res, open(f"{out}/summary.json", "w"), indent=1)
