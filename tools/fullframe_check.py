"""GPU box, one-off: EVERY pixel of a BASELINE config's frame against the oracle at a few samples per pixel (the test suite's
full-size check covers a spread of tiles at the full sample count).  python tools/fullframe_check.py C4 3 [seed]"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from pbrpathtracer_amd import scenes as S
from pbrpathtracer_amd.pathtracer import PathTracer, camera_from_scene
from oracle import oracle_binding as OB
OB.build()
cfg = sys.argv[1]; spp = int(sys.argv[2]) if len(sys.argv) > 2 else 2; seed = int(sys.argv[3]) if len(sys.argv) > 3 else 77
pts, scene, _ = S.build_config(cfg, tempfile.mkdtemp())
pt = PathTracer(0); pt.LoadSceneFile(pts); pt.SetSeed(seed)
cam = camera_from_scene(scene)
if scene.pinhole:
    pt.SetCameraAperture(0.0); cam["aperture"] = 0.0
W, H = pt.GetResolution(); D = pt.GetTraceDepth()
pt.RenderFrames(spp)
got = pt.ReadAccumulation()
t0 = time.time()
o = OB.Oracle(pt.StagedScene())
ocam = OB.make_camera(cam["pos"], cam["dir"], cam["up"], cam["focal"], cam["fovy"], cam["focal_dist"], cam["aperture"])
ref, _ = o.render(ocam, W, H, D, 0, spp, seed, want_rgb8=False)
d = (ref != got).any(axis=2)
print(f"{cfg} {W}x{H} depth {D} spp {spp}: {W * H * spp} samples, oracle {time.time() - t0:.1f} s, differing pixels: {int(d.sum())}", flush=True)
if d.any():
    ys, xs = np.nonzero(d)
    for y, x in list(zip(ys, xs))[:5]: print("  ", x, y, ref[y, x], got[y, x])
sys.exit(1 if d.any() else 0)
