"""GPU box, one-off: a very large frame (default 7680x4320, 33 M pixels) of the Cornell and the blob scenes at a few samples per pixel
against the oracle on the tiles of one rank of a wide split, plus the whole 8-bit hand-off image against the device's own resolve:
pixel / tile / sample-buffer indexing far beyond the BASELINE resolutions.   python tools/big_frame_check.py [W H spp world]"""
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
W = int(sys.argv[1]) if len(sys.argv) > 1 else 7680; H = int(sys.argv[2]) if len(sys.argv) > 2 else 4320
spp = int(sys.argv[3]) if len(sys.argv) > 3 else 3; world = int(sys.argv[4]) if len(sys.argv) > 4 else 9973
bad = 0
for cfg, kw in (("C1", {}), ("C4", dict(grid=24))):
    pts, scene, _ = S.build_config(cfg, tempfile.mkdtemp(), width=W, height=H, depth=4, **kw)
    pt = PathTracer(0); pt.LoadSceneFile(pts); pt.SetSeed(11); pt.SetCameraAperture(0.0)
    out = pt.AllocOutImage(); pt.SetOutImage(out)
    t0 = time.time(); pt.RenderFrames(spp); t_r = time.time() - t0
    assert pt.LastError() == "", pt.LastError()
    got = pt.ReadAccumulation()
    dev8 = pt.context().resolve_rgb8()
    same8 = bool(np.array_equal(np.asarray(out), dev8))
    cam = camera_from_scene(scene); cam["aperture"] = 0.0
    o = OB.Oracle(pt.StagedScene())
    ocam = OB.make_camera(cam["pos"], cam["dir"], cam["up"], cam["focal"], cam["fovy"], cam["focal_dist"], cam["aperture"])
    rank = world // 3
    ref, _ = o.render(ocam, W, H, pt.GetTraceDepth(), 0, spp, 11, rank=rank, world=world, want_rgb8=False)
    o.close()
    mask = D.tile_owner_mask(W, H, rank, world)[::-1]
    diff = int((ref[mask] != got[mask]).any(axis=1).sum())
    print(f"{cfg} {W}x{H} spp {spp}: {t_r * 1e3:.0f} ms ({W * H * spp / t_r / 1e6:.0f} Msamples/s); oracle on {int(mask.sum())} pixels of rank {rank}/{world}: differing {diff}; "
          f"hand-off image equals the device's resolve: {same8}; lit {float((got != 0).any(axis=2).mean()):.2f}", flush=True)
    bad += diff + (0 if same8 else 1)
    pt.SetOutImage(None); pt.close(); del out
sys.exit(1 if bad else 0)
