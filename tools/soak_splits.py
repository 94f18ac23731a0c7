"""GPU box, one-off: scheduling must never change a bit.  Random scenes (tests/test_gpu_random_scenes.py's generator), random
frame sizes, tile splits, sample batching, work-item sizes, pass sizes, persistent / one-item-per-wave launches, generations,
overlap on / off, primary cache on / off - every combination must give the oracle's accumulator.
python tools/soak_splits.py [first_seed] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
from pbrpathtracer_amd import ptk
from oracle import oracle_binding as OB
from test_gpu_random_scenes import random_scene

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ctx = ptk.Context(0)
bad = 0; t0 = time.time()
for k in range(count):
    seed = first + k
    rng = np.random.default_rng(seed)
    n = int(rng.choice([7, 16, 40, 900, 5000]))
    arrays, cam = random_scene(seed, n, bool(k & 1))
    W, H = int(rng.integers(1, 150)), int(rng.integers(1, 100))
    D, spp = int(rng.integers(1, 7)), int(rng.integers(1, 40))
    o = OB.Oracle(arrays)
    ocam = OB.make_camera(cam["pos"], cam["dir"], cam["up"], cam["focal"], cam["fovy"], cam["focal_dist"], cam["aperture"])
    ref, ref8 = o.render(ocam, W, H, D, 0, spp, seed)
    o.close()
    if not np.isfinite(ref).all():
        continue
    opts = dict(chunk=int(rng.choice([0, 1, 2, 3, 8])), persistent=int(rng.choice([-1, 0, 1])), generations=int(rng.choice([0, 1, 2, 3])),
                max_batch=int(rng.choice([1, 4])), overlap=int(rng.choice([0, 1])), primary_cache=int(rng.choice([0, 1])),
                pass_bytes=int(rng.choice([1 << 20, 1 << 22, 1 << 32])), device_build=int(rng.choice([0, 1])))
    for name, v in opts.items(): ctx.set_option(name, v)
    ctx.upload_scene(arrays); ctx.set_camera(**cam); ctx.set_frame(W, H, D)
    world = int(rng.integers(1, 10))
    total = np.zeros_like(ref)
    for rank in range(world):
        ctx.set_tile(rank, world); ctx.reset()
        done = 0
        while done < spp:                                   # the samples in random batches
            b = int(rng.integers(1, spp - done + 1)); ctx.render(done, b, seed); done += b
        part = ctx.read_accum()
        assert not (total != 0).any(axis=2)[(part != 0).any(axis=2)].any(), "two ranks wrote one pixel"
        total += part
    ok = np.array_equal(total, ref)
    if world == 1: ok = ok and np.array_equal(ctx.resolve_rgb8(), ref8)
    if not ok:
        bad += 1
        print(f"MISMATCH seed {seed} n {n} {W}x{H} D{D} spp {spp} world {world} {opts}: {int((total != ref).any(axis=2).sum())} px", flush=True)
    elif k % 25 == 0:
        print(f"seed {seed} ok ({n} tris {W}x{H} spp {spp} world {world} {opts}) [{time.time() - t0:.0f} s]", flush=True)
for name, v in dict(chunk=0, persistent=-1, generations=0, max_batch=1, overlap=1, primary_cache=1, pass_bytes=1 << 32, device_build=1).items(): ctx.set_option(name, v)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
