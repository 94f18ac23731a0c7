"""GPU box, one-off: many seeds of tests/test_gpu_random_scenes.py's generator (not part of the suite: minutes of oracle time).
python tools/soak_random_scenes.py [first_seed] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401  (first copy of the HIP runtime, as tests/conftest.py does)
from pbrpathtracer_amd import ptk
from oracle import oracle_binding as OB
from test_gpu_random_scenes import random_scene

def vary(arrays, cam, variant, seed):
    """stress variants on top of the generator: exact duplicates (ties on t), grid-snapped geometry (coplanar faces, shared
    edges, degenerate triangles), tiny and huge scales"""
    rng = np.random.default_rng(seed * 7 + variant)
    a = {k: v.copy() for k, v in arrays.items()}
    cam = dict(cam)
    per_tri = ("verts", "normals", "uvs", "tbn", "smoothing", "material")
    if variant == 1:
        n = len(a["verts"]); pick = rng.integers(0, n, max(1, n // 3))
        for k in per_tri: a[k] = np.concatenate([a[k], a[k][pick]])
        a["material"][n:] = rng.integers(0, len(a["materials"]), len(pick))
        perm = rng.permutation(len(a["verts"]))
        for k in per_tri: a[k] = np.ascontiguousarray(a[k][perm])
        m = a["materials"]; mat = a["material"]
        a["lights"] = np.nonzero((m["emissive"][mat] * m["emissive_intensity"][mat, None]).sum(axis=1) > 0)[0].astype(np.int32)
    elif variant == 2:
        a["verts"] = (np.round(a["verts"] * 4.0) / 4.0).astype(np.float32)
    elif variant in (3, 4):
        sc = np.float32(0.03 if variant == 3 else 1e3)
        a["verts"] = (a["verts"] * sc).astype(np.float32)
        cam["pos"] = (cam["pos"] * sc).astype(np.float32); cam["focal"] = float(cam["focal"] * sc); cam["focal_dist"] = float(cam["focal_dist"] * sc)
        cam["aperture"] = float(cam["aperture"] * sc)
    return a, cam


first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ctx = ptk.Context(0)
sizes = [int(x) for x in os.environ["SOAK_SIZES"].split(",")] if os.environ.get("SOAK_SIZES") else [5, 12, 16, 17, 64, 500, 3000, 4096, 5000, 20000]
bad = 0
t0 = time.time()
for k in range(count):
    seed = first + k
    n = sizes[k % len(sizes)]
    arrays, cam = random_scene(seed, n, bool(k & 1))
    variant = (seed // 10) % 5
    arrays, cam = vary(arrays, cam, variant, seed)
    if os.environ.get("SOAK_NAN"):
        # NaN soak: degenerate texture coordinates (NaN tangent frames, pathtracer.cpp:557-565 turns them into NaN normals wherever a normal map
        # sits), some zero-length normals on smoothed triangles (GetSmoothNormal normalises them to NaN): every comparison in the kernels must
        # send a NaN the way the oracle's does.  Images are compared with NaN == NaN, position by position.
        r2 = np.random.default_rng(seed + 7)
        arrays["uvs"] = np.where(r2.uniform(size=(len(arrays["uvs"]), 1)) < 0.5, 0.0, arrays["uvs"]).astype(np.float32)
        from pbrpathtracer_amd.pathtracer import lib as _hostlib
        tb = np.zeros((len(arrays["verts"]), 9), np.float32); Lh = _hostlib()
        for i_ in range(len(tb)):
            inp = np.concatenate([arrays["verts"][i_], arrays["uvs"][i_]]).astype(np.float32); o9 = np.zeros(9, np.float32)
            Lh.pth_triangle_init(inp.ctypes.data_as(Lh.pth_triangle_init.argtypes[0]), o9.ctypes.data_as(Lh.pth_triangle_init.argtypes[1])); tb[i_] = o9
        arrays["tbn"] = tb
        zero_n = r2.uniform(size=len(arrays["normals"])) < 0.1
        arrays["normals"] = np.where(zero_n[:, None], 0.0, arrays["normals"]).astype(np.float32)
    W, H, D, spp = 48 + (seed % 3) * 8, 32 + (seed % 5) * 3, 3 + seed % 6, 4
    if os.environ.get("SOAK_DEEP"): D, spp = 6 + seed % 7, 10          # long paths: glass chains, roulette, many light samples
    o = OB.Oracle(arrays)
    ocam = OB.make_camera(cam["pos"], cam["dir"], cam["up"], cam["focal"], cam["fovy"], cam["focal_dist"], cam["aperture"])
    ref, ref8 = o.render(ocam, W, H, D, 0, spp, seed)
    o.close()
    if not np.isfinite(ref).all() and not os.environ.get("SOAK_NAN"):
        print(f"seed {seed}: oracle image not finite (variant {variant}): skipped", flush=True)
        continue
    for dev in (0, 1):
        ctx.set_option("device_build", dev)
        ctx.upload_scene(arrays); ctx.set_camera(**cam); ctx.set_frame(W, H, D); ctx.set_tile(0, 1); ctx.reset()
        ctx.render(0, spp, seed)
        ok = np.array_equal(ref, ctx.read_accum(), equal_nan=True) and np.array_equal(ref8, ctx.resolve_rgb8())
        if not ok:
            bad += 1
            print(f"MISMATCH seed {seed} n {n} variant {variant} device_build {dev}", flush=True)
        # SOAK_OPTS="persistent=1;flat=0,chunk=3": further option sets, each on the same scene
        for opts in [x for x in os.environ.get("SOAK_OPTS", "").split(";") if x]:
            kv = [p.split("=") for p in opts.split(",")]
            for k_, v_ in kv: ctx.set_option(k_, float(v_))
            ctx.reset(); ctx.render(0, spp, seed)
            if not (np.array_equal(ref, ctx.read_accum(), equal_nan=True) and np.array_equal(ref8, ctx.resolve_rgb8())):
                bad += 1
                print(f"MISMATCH seed {seed} n {n} variant {variant} device_build {dev} opts {opts}", flush=True)
            for k_, v_ in kv: ctx.set_option(k_, {"persistent": -1, "flat": 1, "chunk": 0}.get(k_, 0))
    print(f"seed {seed} variant {variant}: {n} triangles {W}x{H} depth {D} ok, NaN pixels {int(np.isnan(ref).any(axis=2).sum())}, lit {(ref != 0).any(axis=2).mean():.2f}  [{time.time() - t0:.0f} s]", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
