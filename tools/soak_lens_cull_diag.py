"""Diagnosis of one soak_lens_cull.py seed: GPU vs oracle, pixel counts per option set."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa
from pbrpathtracer_amd import ptk
from oracle import oracle_binding as OB
from test_gpu_random_scenes import random_scene
ctx = ptk.Context(0)
for seed in [int(x) for x in sys.argv[1:]]:
    k = seed - 1
    rng = np.random.default_rng(seed * 31 + 5)
    n = [12, 17, 64, 500, 5000][k % 5]
    arrays, cam = random_scene(seed, n, bool(k & 1))
    sc = [1.0, 1.0, 0.02, 300.0][(k // 5) % 4]
    arrays["verts"] = (arrays["verts"] * np.float32(sc)).astype(np.float32)
    lo, hi = arrays["verts"].reshape(-1, 3).min(0), arrays["verts"].reshape(-1, 3).max(0)
    ctr, ext = (lo + hi) / 2, float((hi - lo).max())
    mode = k % 7
    if mode == 0:   pos = ctr + rng.normal(size=3) * ext * 0.2
    elif mode == 1: pos = ctr + rng.normal(size=3) * ext * 30.0
    elif mode == 2: pos = hi + rng.uniform(0.0, 0.01, 3) * ext
    else:           pos = ctr + rng.normal(size=3) * ext * 2.0
    aim = ctr + rng.normal(size=3) * ext * [0.2, 1.0, 3.0, 0.0][(k // 3) % 4]
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
    o = OB.Oracle(arrays)
    ocam = OB.make_camera(cam["pos"], cam["dir"], cam["up"], cam["focal"], cam["fovy"], cam["focal_dist"], cam["aperture"], normalise=True)
    for d_, s_ in ((D, spp), (1, 1), (2, 1)):
        ref, _ = o.render(ocam, W, H, d_, 0, s_, seed)
        for opts in ({}, {"device_build": 0}, {"flat": 0}):
            for kk, vv in opts.items(): ctx.set_option(kk, vv)
            ctx.upload_scene(arrays); ctx.set_camera(**cam); ctx.set_frame(W, H, d_); ctx.set_tile(0, 1); ctx.reset(); ctx.render(0, s_, seed)
            got = ctx.read_accum()
            for kk in opts: ctx.set_option(kk, 1)
            bad = (ref != got) & ~(np.isnan(ref) & np.isnan(got))
            px = bad.any(axis=2)
            print(f"seed {seed} depth {d_} spp {s_} opts {opts}: {int(px.sum())} of {W*H} pixels differ; nan ref {int(np.isnan(ref).sum())} got {int(np.isnan(got).sum())}; max |d| {np.nanmax(np.abs(ref-got)):.3g}; first {np.argwhere(px)[:3].tolist()}", flush=True)
    o.close()

# ---- second stage (DIAG2=1): where do GPU and oracle part ways for the LAST seed above? -------------------------------
if os.environ.get("DIAG2"):
    o = OB.Oracle(arrays)
    ctx.upload_scene(arrays); ctx.set_camera(**cam); ctx.set_frame(W, H, 1); ctx.set_tile(0, 1)
    g = ctx.primary_dirs(); r = o.primary_dirs(ocam, W, H)
    print("primary dirs equal:", np.array_equal(g, r), "differing pixels:", int((g != r).any(axis=2).sum()))
    for ap in (cam["aperture"], 0.0):
        c2 = dict(cam); c2["aperture"] = ap
        oc2 = OB.make_camera(c2["pos"], c2["dir"], c2["up"], c2["focal"], c2["fovy"], c2["focal_dist"], c2["aperture"], normalise=True)
        for pc in (1, 0):
            ctx.set_option("primary_cache", pc)
            ctx.set_camera(**c2); ctx.reset(); ctx.render(0, 1, seed); got = ctx.read_accum()
            ref, _ = o.render(oc2, W, H, 1, 0, 1, seed)
            px = (ref != got).any(axis=2)
            print(f"aperture {ap:g} primary_cache {pc}: {int(px.sum())} pixels differ; e.g.", [(tuple(p), ref[tuple(p)].tolist(), got[tuple(p)].tolist()) for p in np.argwhere(px)[:2]])
        ctx.set_option("primary_cache", 1)
    # materials of the scene
    m = arrays["materials"]
    print("material types", np.unique(m["type"], return_counts=True), "lights", len(arrays["lights"]), "emissive max", float((m["emissive"] * m["emissive_intensity"][:, None]).max()))
