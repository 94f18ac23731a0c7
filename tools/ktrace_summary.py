"""Summary of a rocprofv3 --kernel-trace CSV: per kernel count / mean duration, and the timeline of the LAST `n` dispatches
(start offsets, durations, gaps, overlap with the previous dispatch).  python3 tools/ktrace_summary.py <dir> [n] [json-out]"""
import collections, csv, glob, json, sys
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rows = []
for f in glob.glob(f"{d}/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", ""))))
rows.sort()
per = collections.defaultdict(list)
for s, e, k, q in rows: per[k].append((e - s) / 1e3)
out = {"kernels": {k[:90]: {"count": len(v), "mean_us": sum(v) / len(v), "min_us": min(v), "max_us": max(v)} for k, v in per.items()}}
for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["count"] * kv[1]["mean_us"]):
    print(f"{v['count']:6d} x {v['mean_us']:10.2f} us (min {v['min_us']:.2f} max {v['max_us']:.2f})  {k}")
tail = rows[-n:]
t0 = tail[0][0] if tail else 0
tl = []
prev_end = None
for s, e, k, q in tail:
    gap = None if prev_end is None else (s - prev_end) / 1e3
    tl.append({"start_us": (s - t0) / 1e3, "dur_us": (e - s) / 1e3, "gap_after_prev_end_us": gap, "queue": q, "kernel": k[:60]})
    print(f"  +{(s - t0) / 1e3:9.2f} us  dur {(e - s) / 1e3:9.2f}  gap {gap if gap is None else round(gap, 2)}  q{q}  {k[:70]}")
    prev_end = e if prev_end is None else max(prev_end, e)
out["timeline_tail"] = tl
if len(sys.argv) > 3: json.dump(out, open(sys.argv[3], "w"), indent=1)
