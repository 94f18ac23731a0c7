"""GPU box: how the GPU image of some soak seeds differs from the oracle's (tools/soak_random_scenes.py's scenes)."""
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
for kv in os.environ.get("PTK_OPTS", "").split(","):
    if "=" in kv:
        k_, v_ = kv.split("="); ctx.set_option(k_, float(v_))
for seed in [int(x) for x in sys.argv[1:]]:
    k = seed - 1000; n = sizes[k % 10]
    arrays, cam = random_scene(seed, n, bool(k & 1))
    W, H, D, spp = 48 + (seed % 3) * 8, 32 + (seed % 5) * 3, 3 + seed % 6, 4
    o = OB.Oracle(arrays)
    ocam = OB.make_camera(cam["pos"], cam["dir"], cam["up"], cam["focal"], cam["fovy"], cam["focal_dist"], cam["aperture"])
    ctx.upload_scene(arrays); ctx.set_camera(**cam); ctx.set_frame(W, H, D); ctx.set_tile(0, 1)
    line = f"seed {seed} n {n} {W}x{H} D{D} aperture {cam['aperture']}:"
    for s1 in range(spp):                       # sample by sample: which sample of which pixel differs
        ref, _ = o.render(ocam, W, H, D, s1, 1, seed)
        ctx.reset(); ctx.render(s1, 1, seed); got = ctx.read_accum()
        d = (ref != got).any(axis=2)
        if d.any():
            ys, xs = np.nonzero(d)
            line += f" sample {s1}: {int(d.sum())} px, e.g. ({xs[0]},{ys[0]}) ref {ref[ys[0], xs[0]]} got {got[ys[0], xs[0]]};"
    o.close()
    print(line, flush=True)
