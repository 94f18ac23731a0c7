"""GPU box, one-off: the exchange step's pack / unpack kernels against the library's host-side layout on random frame sizes and
world sizes up to 64 (one GPU plays every rank).  python tools/soak_exchange.py [count]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
from pbrpathtracer_amd import ptk
from conftest import load_golden, scene_from_golden
count = int(sys.argv[1]) if len(sys.argv) > 1 else 200
z = load_golden("tier_s_cornell.npz")
ctx = ptk.Context(0)
ctx.upload_scene(scene_from_golden(z))
cam, proj = z["cam"], z["proj"]
ctx.set_camera(pos=cam[0:3], dir=cam[3:6], up=cam[6:9], focal=float(proj[0]), fovy=float(proj[1]), focal_dist=float(z["focal_dist"]), aperture=0.0)
rng = np.random.default_rng(5)
bad = 0
for k in range(count):
    W, H = int(rng.integers(1, 300)), int(rng.integers(1, 200))
    world = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 16, 33, 64]))
    ctx.set_frame(W, H, 2); ctx.set_tile(0, 1); ctx.reset()
    accum = rng.uniform(-1, 1, (H, W, 3)).astype(np.float32)
    ctx.write_accum(accum, 1)
    flat = accum.reshape(-1)
    packed = []; ok = True; covered = np.zeros(W * H * 3, np.int32)
    for r in range(world):
        lay = ptk.packed_layout(W, H, r, world)
        got = ctx.probe_pack(r, world)
        exp = np.where(lay >= 0, flat[np.maximum(lay, 0)], np.float32(0.0))
        ok = ok and np.array_equal(got, exp)
        np.add.at(covered, lay[lay >= 0], 1)
        packed.append(got)
    ok = ok and (covered == 1).all()                       # every float of the image belongs to exactly one rank
    image = ctx.probe_unpack(world, np.concatenate(packed))
    ok = ok and np.array_equal(image, accum)
    if not ok:
        bad += 1; print(f"MISMATCH {W}x{H} world {world}", flush=True)
print("cases", count, "mismatches:", bad)
sys.exit(1 if bad else 0)
