"""Turn the raw output of tools/profile_round.sh (gpurun_out/<dir>) into the tracked summaries under profiles/<dir>.

    python tools/profile_collect.py gpurun_out/r01b profiles/r01_final
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)


def dispatches(d, kernel_substr):
    """{dispatch id: {counter: value, ms, grid}} of the kernels whose name contains kernel_substr."""
    per = collections.defaultdict(dict)
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel_substr in r["Kernel_Name"]:
                e = per[(f, r["Dispatch_Id"])]
                e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                e["ms"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
                e["grid"] = int(r["Grid_Size"]); e["kernel"] = r["Kernel_Name"]
    return list(per.values())


def full_launches(ds):
    """the full-size launches = the long ones (not the 1-spp set-up launch or the short counters-enabled run); the grid
    size does not tell them apart: big launches run as a fixed number of persistent waves"""
    if not ds:
        return []
    longest = max(x["ms"] for x in ds)
    return [x for x in ds if x["ms"] >= 0.5 * longest]


for c in ("C1", "C2", "C3", "C4", "C5"):
    f = os.path.join(src, f"bench_{c}.json")
    if os.path.exists(f) and os.path.getsize(f):
        shutil.copy(f, os.path.join(dst, f"bench_{c}.json"))
for f in glob.glob(f"{src}/ktrace/**/*_kernel_stats.csv", recursive=True):
    shutil.copy(f, os.path.join(dst, "rocprofv3_kernel_stats_bench_C2.csv"))

# HBM traffic of the headline config: FETCH_SIZE (x2 on gfx950, MI355X_MICROARCH.md) + WRITE_SIZE, KiB units
tr = {}
for name, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
    for kern, key in (("trace_kernel<false", "trace"), ("accumulate_kernel", "accumulate")):
        ds = full_launches(dispatches(os.path.join(src, sub), kern))
        if ds:
            tr[(key, name)] = sum(x[name] for x in ds) / len(ds)
            tr[(key, "ms")] = sum(x["ms"] for x in ds) / len(ds)
            tr[(key, "n")] = len(ds)
            tr[(key, "kernel")] = ds[0]["kernel"]
if ("trace", "FETCH_SIZE") in tr and ("trace", "WRITE_SIZE") in tr:
    out = {
        "config": "C2", "spp": 256, "n_gpus": 1, "kernel": tr[("trace", "kernel")],
        "hbm_bytes_per_launch": int((2 * tr[("trace", "FETCH_SIZE")] + tr[("trace", "WRITE_SIZE")]) * 1024),
        "fetch_size_kib": tr[("trace", "FETCH_SIZE")], "write_size_kib": tr[("trace", "WRITE_SIZE")],
        "launches_averaged": tr[("trace", "n")], "kernel_ms_under_profiler": tr[("trace", "ms")],
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py --steps 3 --warmup 1 "
                  "--no-cpu-baseline`; FETCH_SIZE x2 per MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B), WRITE_SIZE as read",
        "accumulate_kernel": {"FETCH_SIZE_KiB": tr.get(("accumulate", "FETCH_SIZE")), "WRITE_SIZE_KiB": tr.get(("accumulate", "WRITE_SIZE")),
                              "ms": tr.get(("accumulate", "ms")), "launches": tr.get(("accumulate", "n"))},
    }
    json.dump(out, open(os.path.join(dst, "pmc_hbm_traffic_C2.json"), "w"), indent=1)
    json.dump(out, open(os.path.join(os.path.dirname(dst.rstrip("/")), "traffic_C2.json"), "w"), indent=1)
    print("traffic", out["hbm_bytes_per_launch"] / 1e9, "GB per launch")

# SQ counters of the full-size trace_kernel launches
for c in ("C2", "C4"):
    merged = {}
    for p in sorted(glob.glob(os.path.join(src, f"sq_{c}", "p*"))):
        ds = full_launches(dispatches(p, "trace_kernel<false"))
        for k in (ds[0] if ds else {}):
            if k not in ("grid", "kernel"):
                merged[k] = sum(x[k] for x in ds) / len(ds)
        if ds:
            merged["kernel"] = ds[0]["kernel"]; merged["grid_threads"] = ds[0]["grid"]
    if "SQ_INSTS_VALU" in merged:
        m = merged
        m["lane_utilisation"] = m["SQ_THREAD_CYCLES_VALU"] / (64.0 * m["SQ_INSTS_VALU"])
        m["wave_life_in_s_waitcnt"] = m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"]
        m["wave_life_issuing_valu"] = m["SQ_ACTIVE_INST_VALU"] / m["SQ_WAVE_CYCLES"]
        m["wave_life_waiting_to_issue"] = m["SQ_WAIT_INST_ANY"] / m["SQ_WAVE_CYCLES"]
        m["avg_waves_per_simd"] = m["SQ_WAVE_CYCLES"] * 4 / (m["ms"] * 1e-3 * 2.4e9) / 1024
        m["valu_pipe_busy"] = min(1.0, m["wave_life_issuing_valu"] * m["avg_waves_per_simd"])
        m["clock_assumed_GHz"] = 2.4        # avg_waves_per_simd and valu_pipe_busy scale with it (the chip clocks lower under load)
        m["units"] = "SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_* are in quad-cycles summed over waves; ms is the kernel time under the profiler"
        json.dump(m, open(os.path.join(dst, f"pmc_sq_trace_kernel_{c}.json"), "w"), indent=1)
        print(c, {k: (round(v, 3) if isinstance(v, float) else v) for k, v in m.items() if k[0].islower()})
