"""GPU box, one-off: the exact lens cull (live_mask_kernel: pixels none of whose lens rays can reach the scene's bounds are never
traced) against the cull-disabled kernel on random scenes and ADVERSARIAL cameras - inside the scene, far away, looking past it,
grazing its bounds, apertures from a pinhole's 1e-11 to wider than the scene, focal distances from far behind the scene to shorter
than the aperture, tiny and huge scene scales.  The two images must be bit-identical (every few cases also against the oracle).
    python tools/soak_lens_cull.py [first_seed] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
from pbrpathtracer_amd import ptk
from oracle import oracle_binding as OB
from test_gpu_random_scenes import random_scene

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 200
ctx = ptk.Context(0)
bad = 0; culled_cases = 0; started_sum = 0; total_sum = 0
t0 = time.time()
for k in range(count):
    seed = first + k
    rng = np.random.default_rng(seed * 31 + 5)
    n = [12, 17, 64, 500, 5000][k % 5]
    arrays, cam = random_scene(seed, n, bool(k & 1))
    sc = [1.0, 1.0, 0.02, 300.0][(k // 5) % 4]
    arrays["verts"] = (arrays["verts"] * np.float32(sc)).astype(np.float32)
    lo, hi = arrays["verts"].reshape(-1, 3).min(0), arrays["verts"].reshape(-1, 3).max(0)
    ctr, ext = (lo + hi) / 2, float((hi - lo).max())
    mode = k % 7
    if mode == 0:   pos = ctr + rng.normal(size=3) * ext * 0.2                       # inside the scene
    elif mode == 1: pos = ctr + rng.normal(size=3) * ext * 30.0                      # far away
    elif mode == 2: pos = hi + rng.uniform(0.0, 0.01, 3) * ext                       # on a corner of the bounds
    else:           pos = ctr + rng.normal(size=3) * ext * 2.0
    aim = ctr + rng.normal(size=3) * ext * [0.2, 1.0, 3.0, 0.0][(k // 3) % 4]        # at the scene, beside it, past it, dead centre
    d = aim - pos
    if not np.isfinite(d).all() or np.abs(d).max() == 0: d = np.array([0.0, 0.0, 1.0])
    d = d / np.linalg.norm(d)
    up = np.array([0.0, 1.0, 0.0]) if abs(d[1]) < 0.95 else np.array([1.0, 0.0, 0.0])
    dist = float(np.linalg.norm(ctr - pos))
    aperture = float([5e-11, 0.01 * ext, 0.3 * ext, 2.5 * ext, 1e-4 * ext][(k // 2) % 5])
    focal_dist = float([dist, 0.3 * dist, 4.0 * dist, 0.5 * aperture + 1e-6, dist][(k // 11) % 5])
    cam = dict(pos=pos.astype(np.float32), dir=d.astype(np.float32), up=up.astype(np.float32), focal=float(0.1 * sc), fovy=float([40, 75, 110, 160][k % 4]),
               focal_dist=focal_dist, aperture=aperture)
    W, H, D, spp = 64 + (seed % 3) * 16, 36 + (seed % 5) * 4, 3 + seed % 4, 3
    ctx.upload_scene(arrays); ctx.set_camera(**cam); ctx.set_frame(W, H, D); ctx.set_tile(0, 1)
    imgs = []
    for cull in (1, 0):
        ctx.set_option("lens_cull", cull); ctx.reset(); ctx.render(0, spp, seed)
        imgs.append((ctx.read_accum(), ctx.resolve_rgb8()))
    ctx.set_option("lens_cull", 1)
    st = ctx.collect_stats(0, spp, seed)
    started_sum += st["paths_started"]; total_sum += st["samples"]; culled_cases += st["paths_started"] < st["samples"]
    ok = np.array_equal(imgs[0][0], imgs[1][0], equal_nan=True) and np.array_equal(imgs[0][1], imgs[1][1])
    if ok and k % 2 == 0:
        o = OB.Oracle(arrays)
        ocam = OB.make_camera(cam["pos"], cam["dir"], cam["up"], cam["focal"], cam["fovy"], cam["focal_dist"], cam["aperture"], normalise=True)
        ref, _ = o.render(ocam, W, H, D, 0, spp, seed); o.close()
        ok = np.array_equal(ref, imgs[0][0], equal_nan=True)
    if not ok:
        bad += 1
        diff = int((imgs[0][0] != imgs[1][0]).any(axis=2).sum())
        print(f"MISMATCH seed {seed}: n {n} scale {sc} mode {mode} aperture {aperture:g} focal_dist {focal_dist:g} fovy {cam['fovy']}: {diff} pixels differ between cull on / off", flush=True)
    if k % 25 == 24:
        print(f"{k + 1} cases, {bad} mismatches, {culled_cases} with culled pixels, traced fraction {started_sum / max(1, total_sum):.3f}, {time.time() - t0:.0f} s", flush=True)
print(f"done: {count} cases, {bad} mismatches, {culled_cases} with culled pixels, traced fraction {started_sum / max(1, total_sum):.3f}")
sys.exit(1 if bad else 0)
