#!/usr/bin/env python3
"""Prints the tables of DESIGN.md section 7: the measurement table from profiles/<round>/ (bench lines, counter summaries) and the
kernels' register / LDS / spill figures read from the gfx950 code objects INSIDE the built pbrpathtracer_amd/libptk.so
(.hip_fatbin section -> clang offload bundles -> llvm-readelf --notes), so the document quotes the binary, not prose.

    python tools/design_table.py [profiles/r04]"""
import json
import os
import sys

d = sys.argv[1] if len(sys.argv) > 1 else "profiles/r04"


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


# ---- what the code objects in libptk.so say about the kernels -----------------------------------------------------------
def kernel_resources(lib):
    """[(kernel, vgprs, sgprs, vgpr spills, sgpr spills, LDS bytes, scratch bytes)] of every gfx950 kernel in the library"""
    import re
    import struct
    import subprocess
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fatbin")
        subprocess.run([f"{llvm}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib, os.devnull], check=True, capture_output=True)
        data = open(fat, "rb").read()
        magic, pos, n = b"__CLANG_OFFLOAD_BUNDLE__", 0, 0
        while True:
            i = data.find(magic, pos)
            if i < 0:
                break
            num = struct.unpack_from("<Q", data, i + 24)[0]
            off = i + 32
            for _ in range(num):
                o, sz, tl = struct.unpack_from("<QQQ", data, off); off += 24
                triple = data[off:off + tl].decode(); off += tl
                if "gfx950" in triple and sz:
                    co = os.path.join(tmp, f"co{n}.co"); n += 1
                    open(co, "wb").write(data[i + o:i + o + sz])
                    notes = subprocess.run([f"{llvm}/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
                    cur = {}
                    for ln in notes.splitlines():
                        m = re.match(r"\s+\.(\w+):\s+(\S+)", ln)
                        if not m:
                            continue
                        k, v = m.group(1), m.group(2)
                        if k == "group_segment_fixed_size" and "name" in cur:      # (first key of the next kernel's map)
                            out.append(cur); cur = {}
                        cur[k] = v
                    if "name" in cur:
                        out.append(cur)
            pos = i + 24
    rows = []
    for k in out:
        name = subprocess.run(["c++filt", k["name"]], capture_output=True, text=True).stdout.strip() or k["name"]
        name = re.sub(r"\(.*\)$", "", name)
        rows.append((name, int(k.get("vgpr_count", 0)), int(k.get("sgpr_count", 0)), int(k.get("vgpr_spill_count", 0)), int(k.get("sgpr_spill_count", 0)),
                     int(k.get("group_segment_fixed_size", 0)), int(k.get("private_segment_fixed_size", 0))))
    return rows


lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pbrpathtracer_amd", "libptk.so")
if os.path.exists(lib):
    try:
        rows = kernel_resources(lib)
        print()
        print("Kernel resources as built (gfx950 code objects inside `pbrpathtracer_amd/libptk.so`; waves / SIMD = min(8, 512 / VGPRs rounded up to 8)):")
        print()
        print("| kernel | VGPRs | SGPRs | VGPR spills | SGPR spills | LDS bytes / workgroup | scratch bytes / lane | waves / SIMD by registers |")
        print("|---|---|---|---|---|---|---|---|")
        for r in sorted(rows):
            if "trace" in r[0] or "accumulate" in r[0] or "level_bin" in r[0] or "collapse" in r[0]:
                alloc = (r[1] + 7) // 8 * 8
                print("| `%s` | %d | %d | %d | %d | %d | %d | %d |" % (r[0], r[1], r[2], r[3], r[4], r[5], r[6], min(8, 512 // max(8, alloc))))
    except Exception as e:          # (no llvm tools: the measurement table above still stands)
        print(f"(kernel resources not read: {type(e).__name__}: {e})")
