"""GPU box: what one rank of an N-way tile split costs per step on ONE GPU (no exchange): the strong-scaling ceiling of the
trace kernel itself.  python tools/rank_share_probe.py [C2] [steps]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pbrpathtracer_amd import scenes as S
from pbrpathtracer_amd.pathtracer import PathTracer
name = sys.argv[1] if len(sys.argv) > 1 else "C2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
pts, scene, spp = S.build_config(name, tempfile.mkdtemp())
pt = PathTracer(0); pt.LoadSceneFile(pts)
if scene.pinhole: pt.SetCameraAperture(0.0)
pt.SetSeed(1)
ctx = pt.context()
for kv in os.environ.get("PTK_OPTS", "").split(","):
    if "=" in kv:
        k, v = kv.split("="); ctx.set_option(k, float(v))
base = None
for world in (1, 2, 4, 8):
    worst = 0.0
    for rank in sorted({0, world // 2, world - 1}):
        pt.SetTile(rank, world); pt.ResetImage(); pt.RenderFrames(spp); ctx.synchronize()
        t0 = time.time()
        for _ in range(steps): pt.RenderFrames(spp)
        ctx.synchronize()
        worst = max(worst, (time.time() - t0) / steps * 1e3)
    base = base or worst
    print(f"{name} world {world}: slowest probed rank {worst:.3f} ms / step -> kernel-side scaling efficiency {base / (worst * world):.2f}", flush=True)
