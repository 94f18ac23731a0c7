"""GPU box: closest hits of many rays through the host-built and the device-built tree of a config's scene: any difference is
a box test that was not conservative for one of the two trees.  python tools/tree_independence_probe.py C5 4000000"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from pbrpathtracer_amd import scenes as S
from pbrpathtracer_amd.pathtracer import PathTracer
name = sys.argv[1] if len(sys.argv) > 1 else "C5"
nrays = int(sys.argv[2]) if len(sys.argv) > 2 else 4000000
pts, scene, _ = S.build_config(name, tempfile.mkdtemp())
rng = np.random.default_rng(1)
out = {}
for dev in (0, 1):
    pt = PathTracer(0); pt.context().set_option("device_build", dev); pt.LoadSceneFile(pts)
    ctx = pt.context()
    if dev == 0:
        # rays as a render produces them: camera rays, then bounce / shadow-like rays from the points they hit
        W, H = pt.GetResolution()
        pt.SetCameraAperture(0.0); pt.RenderFrames(1)
        d = ctx.primary_dirs().reshape(-1, 3)
        cam = np.array(pt.GetCameraPosition() if hasattr(pt, "GetCameraPosition") else scene.cam_pos, np.float32)
        sel = rng.integers(0, len(d), nrays // 4)
        ro0 = np.tile(cam, (len(sel), 1)).astype(np.float32); rd0 = d[sel].astype(np.float32)
        tri, tuv = ctx.probe_hits(ro0, rd0)
        hitp = (ro0 + rd0 * tuv[:, :1])[tri >= 0]
        k = nrays - len(ro0)
        src = hitp[rng.integers(0, len(hitp), k)].astype(np.float32)
        dirs = rng.normal(0, 1, (k, 3)).astype(np.float32); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        dirs[::97, 0] = 0.0; dirs[::89, 2] = 0.0
        ro = np.concatenate([ro0, src + dirs * np.float32(1e-4)]); rd = np.concatenate([rd0, dirs])
    out[dev] = ctx.probe_hits(ro, rd)
    print(name, "device_build", dev, "tree", ctx.bvh_info(), "hits", float((out[dev][0] >= 0).mean()), flush=True)
    pt.close()
dt = out[0][0] != out[1][0]
du = (out[0][1] != out[1][1]).any(axis=1) & ~dt
print(f"{name}: {len(ro)} rays, different triangle: {int(dt.sum())}, same triangle but different (t,u,v): {int(du.sum())}")
if dt.any():
    i = np.nonzero(dt)[0][:5]
    for j in i: print("  ray", j, "ro", ro[j], "rd", rd[j], "host", out[0][0][j], out[0][1][j], "device", out[1][0][j], out[1][1][j])
