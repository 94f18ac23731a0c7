"""GPU box, one-off: a height field of many million triangles (C5's generator with a finer grid) through the whole path - .obj
ingest, device BVH build, render - against the oracle on the tiles of one rank of a wide split.  Index widths, leaf codes and
buffer sizes far beyond the BASELINE configs.   python tools/big_scene_check.py [nx nz spp world]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from pbrpathtracer_amd import scenes as S
from pbrpathtracer_amd import distributed as D
from pbrpathtracer_amd.pathtracer import PathTracer, camera_from_scene
from oracle import oracle_binding as OB
OB.build()
nx = int(sys.argv[1]) if len(sys.argv) > 1 else 2828
nz = int(sys.argv[2]) if len(sys.argv) > 2 else 1414
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 16
world = int(sys.argv[4]) if len(sys.argv) > 4 else 499
t0 = time.time()
pts, scene, _ = S.build_config("C5", tempfile.mkdtemp(), nx=nx, nz=nz, depth=6)
t_gen = time.time() - t0; t0 = time.time()
pt = PathTracer(0); pt.LoadSceneFile(pts); pt.SetSeed(5); pt.SetCameraAperture(0.0)
t_load = time.time() - t0
cam = camera_from_scene(scene); cam["aperture"] = 0.0
W, H = pt.GetResolution(); Dp = pt.GetTraceDepth()
print(f"{pt.GetTriangleCount()} triangles: scene files {t_gen:.1f} s, ingest + BVH + upload {t_load:.1f} s; bvh {pt.context().bvh_info()}", flush=True)
t0 = time.time(); pt.RenderFrames(spp); got = pt.ReadAccumulation(); t_r = time.time() - t0
assert pt.LastError() == "", pt.LastError()
print(f"render {W}x{H} depth {Dp} spp {spp}: {t_r * 1e3:.0f} ms -> {W * H * spp / t_r / 1e6:.0f} Msamples/s", flush=True)
t0 = time.time()
o = OB.Oracle(pt.StagedScene())
ocam = OB.make_camera(cam["pos"], cam["dir"], cam["up"], cam["focal"], cam["fovy"], cam["focal_dist"], cam["aperture"])
rank = world // 3
ref, _ = o.render(ocam, W, H, Dp, 0, spp, 5, rank=rank, world=world, want_rgb8=False)
mask = D.tile_owner_mask(W, H, rank, world)[::-1]
bad = int((ref[mask] != got[mask]).any(axis=1).sum())
print(f"oracle (build + {int(mask.sum())} pixels) {time.time() - t0:.1f} s; lit {float((ref[mask] != 0).any(axis=1).mean()):.2f}; differing pixels: {bad}", flush=True)
sys.exit(1 if bad else 0)
