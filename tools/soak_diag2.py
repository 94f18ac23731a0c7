"""GPU box: for one soak seed / sample / pixel, the trace depth at which GPU and oracle part."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
from pbrpathtracer_amd import ptk
from oracle import oracle_binding as OB
from test_gpu_random_scenes import random_scene
sizes = [5, 12, 16, 17, 64, 500, 3000, 4096, 5000, 20000]
ctx = ptk.Context(0)
seed, s1, px, py = [int(x) for x in sys.argv[1:5]]
k = seed - 1000; n = sizes[k % 10]
arrays, cam = random_scene(seed, n, bool(k & 1))
W, H, Dmax = 48 + (seed % 3) * 8, 32 + (seed % 5) * 3, 3 + seed % 6
o = OB.Oracle(arrays)
ocam = OB.make_camera(cam["pos"], cam["dir"], cam["up"], cam["focal"], cam["fovy"], cam["focal_dist"], cam["aperture"])
ctx.upload_scene(arrays); ctx.set_camera(**cam); ctx.set_tile(0, 1)
m = arrays["materials"]
print("lights", arrays["lights"].tolist(), "material of tris", arrays["material"].tolist()[:20])
for D in range(0, Dmax + 3):
    ctx.set_frame(W, H, D)
    ref, _ = o.render(ocam, W, H, D, s1, 1, seed)
    for flat in ((1, 0) if n <= 16 else (0,)):
        ctx.set_option("flat", flat)
        ctx.reset(); ctx.render(s1, 1, seed); got = ctx.read_accum()
        print(f"D {D} flat {flat}: ref {ref[py, px]} got {got[py, px]} {'' if np.array_equal(ref[py, px], got[py, px]) else '  <-- differs'}   (whole frame differs in {int((ref != got).any(axis=2).sum())} px)")
o.close()
