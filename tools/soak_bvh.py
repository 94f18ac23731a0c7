"""GPU box, one-off: the device BVH builder on adversarial geometry - validity of what lies in HBM (tests/bvh_check.py) and the
same closest hits as the host builder's tree.  python tools/soak_bvh.py [first_seed] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
from pbrpathtracer_amd import ptk
from bvh_check import check_bvh
from test_gpu_bvh_build import _scene

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
ctx = ptk.Context(0)
kinds = ["soup", "clustered", "identical", "line", "plane grid", "two far clusters", "huge and tiny", "duplicates", "big coordinates", "slivers"]
bad = 0; fell_back = 0
t0 = time.time()
for k in range(count):
    seed = first + k
    rng = np.random.default_rng(seed)
    kind = kinds[k % len(kinds)]
    n = int(rng.choice([4096, 5000, 12345, 40000]))
    c = rng.uniform(-1, 1, (n, 1, 3)); size = 0.02
    if kind == "clustered": c = c ** 5
    elif kind == "identical": c = np.tile(c[:1], (n, 1, 1))
    elif kind == "line": c = c * np.array([1.0, 0.0, 0.0]) + np.array([0.0, 0.3, -0.2])
    elif kind == "plane grid":
        g = int(np.ceil(np.sqrt(n))); ij = np.stack(np.meshgrid(np.arange(g), np.arange(g)), -1).reshape(-1, 2)[:n]
        c = np.concatenate([ij / g * 2 - 1, np.zeros((n, 1))], axis=1)[:, None, :]
    elif kind == "two far clusters": c = c * 1e-3 + np.where(rng.uniform(0, 1, (n, 1, 1)) < 0.5, -1e3, 1e3)
    elif kind == "big coordinates": c = c * 1e15; size = 1e13
    v = c + size * rng.uniform(-1, 1, (n, 3, 3))
    if kind == "identical": v = np.tile(v[:1], (n, 1, 1))
    if kind == "huge and tiny": v[: n // 50] = c[: n // 50] + 1.5 * rng.uniform(-1, 1, (n // 50, 3, 3))
    if kind == "duplicates": v[n // 2:] = v[: n - n // 2]
    if kind == "slivers": v[:, 2] = v[:, 0] + (v[:, 1] - v[:, 0]) * rng.uniform(0, 1, (n, 1)) + 1e-7 * rng.normal(0, 1, (n, 3))
    verts = v.astype(np.float32).reshape(n, 9)
    arrays = _scene(verts)
    ext = float(np.abs(verts).max())
    ro = (rng.uniform(-1.5, 1.5, (3000, 3)) * ext).astype(np.float32)
    tgt = verts.reshape(n, 3, 3)[rng.integers(0, n, 3000)].mean(axis=1)          # aim at triangles so that rays hit
    rd = (tgt - ro); rd /= np.maximum(np.linalg.norm(rd, axis=1, keepdims=True), 1e-30); rd = rd.astype(np.float32)
    rd[::40, 1] = 0.0
    try:
        ctx.set_option("device_build", 0); ctx.upload_scene(arrays); tri_h, tuv_h = ctx.probe_hits(ro, rd)
        ctx.set_option("device_build", 1); ctx.upload_scene(arrays)
        dev = ctx.upload_timing()["built_on_device"]
        fell_back += 0 if dev else 1
        nodes, order = ctx.download_bvh()
        info = check_bvh(nodes, order, verts)
        tri_d, tuv_d = ctx.probe_hits(ro, rd)
        ok = np.array_equal(tri_h, tri_d) and np.array_equal(tuv_h, tuv_d)
        if not ok: bad += 1
        print(f"seed {seed} {kind} n {n}: {'ok' if ok else 'HITS DIFFER'} device {dev} hits {(tri_h >= 0).mean():.2f} {info}  [{time.time() - t0:.0f} s]", flush=True)
    except (AssertionError, ptk.PtkError) as e:
        bad += 1
        print(f"seed {seed} {kind} n {n}: FAILED {type(e).__name__}: {str(e)[:200]}", flush=True)
print("bad:", bad, "fell back to the host builder:", fell_back)
sys.exit(1 if bad else 0)
