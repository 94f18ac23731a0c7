#!/usr/bin/env python3
"""Prints the measurement table of DESIGN.md section 7 from profiles/r03/ (bench lines, counter summaries)."""
import json
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "profiles/r03"


def load(name):
    p = os.path.join(d, name)
    if not os.path.exists(p):
        return None
    txt = open(p).read().strip()
    return json.loads(txt.splitlines()[-1]) if name.startswith("bench_") else json.loads(txt)


rows = []
print("| config | Msamples/s | ms / step | traced Msamples/s | bound | §8(d4) flops ÷ 157.3 TF | L2-miss traffic (counters) | L2 hit | lane util. (counters) | VALU pipe busy (class bounds) | node visits, triangle tests, shaded hits / sample |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for c in ("C1", "C2", "C3", "C4", "C5"):
    b = load(f"bench_{c}.json")
    if b is None:
        continue
    t = load(f"traffic_{c}.json")
    sq = load(f"pmc_sq_trace_kernel_{c}.json")
    r = b["roofline"]
    ps = r["per_sample"]
    traffic = "%.0f GB/s (%.1f %% of 8 TB/s)" % (t["hbm_GBps"], t["hbm_GBps"] / 80.0) if t else "—"
    l2 = "%.2f" % t["cache"]["l2_hit_rate"] if t and t.get("cache") else "—"
    lu = "%.2f" % sq["lane_utilisation"] if sq else "—"
    vb = "%.2f – %.2f" % (sq["valu_pipe_busy_low"], min(1.0, sq["valu_pipe_busy_high"])) if sq else "—"
    print("| %s %s | **%.0f** | %.3f | %.0f | %s | %.3f | %s | %s | %s | %s | %.1f, %.1f, %.2f |" % (
        c, b["config"]["workload"].split(",")[0], b["value"], b["ms_per_step"], b["traced_samples_per_s"], r["bound"], r["valu"]["frac"],
        traffic, l2, lu, vb, ps["node_visits"], ps["tri_tests"], ps["hits_shaded"]))
b = load("bench_C2.json")
if b and b.get("cpu_baseline"):
    cb = b["cpu_baseline"]
    print()
    print("CPU baseline (same box, `bench.py`'s bounded sample): %s %s on %s threads, kind `%s`; modes: %s" % (
        cb["value"], cb["unit"], cb["cores"], cb["kind"], json.dumps(cb.get("modes", {}))[:600]))
