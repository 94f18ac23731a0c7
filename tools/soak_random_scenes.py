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

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ctx = ptk.Context(0)
sizes = [5, 12, 16, 17, 64, 500, 3000, 4096, 5000, 20000]
bad = 0
t0 = time.time()
for k in range(count):
    seed = first + k
    n = sizes[k % len(sizes)]
    arrays, cam = random_scene(seed, n, bool(k & 1))
    W, H, D, spp = 48 + (seed % 3) * 8, 32 + (seed % 5) * 3, 3 + seed % 6, 4
    o = OB.Oracle(arrays)
    ocam = OB.make_camera(cam["pos"], cam["dir"], cam["up"], cam["focal"], cam["fovy"], cam["focal_dist"], cam["aperture"])
    ref, ref8 = o.render(ocam, W, H, D, 0, spp, seed)
    o.close()
    for dev in (0, 1):
        ctx.set_option("device_build", dev)
        ctx.upload_scene(arrays); ctx.set_camera(**cam); ctx.set_frame(W, H, D); ctx.set_tile(0, 1); ctx.reset()
        ctx.render(0, spp, seed)
        ok = np.array_equal(ref, ctx.read_accum()) and np.array_equal(ref8, ctx.resolve_rgb8())
        if not ok:
            bad += 1
            print(f"MISMATCH seed {seed} n {n} device_build {dev}", flush=True)
    print(f"seed {seed}: {n} triangles {W}x{H} depth {D} ok, lit {(ref != 0).any(axis=2).mean():.2f}  [{time.time() - t0:.0f} s]", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
