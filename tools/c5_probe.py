"""Scratch probe (GPU box): per-pass timing and tail diagnostics of a config."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pbrpathtracer_amd import scenes as S
from pbrpathtracer_amd.pathtracer import PathTracer
name = sys.argv[1] if len(sys.argv) > 1 else "C5"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
pts, scene, _ = S.build_config(name, tempfile.mkdtemp())
pt = PathTracer(0); pt.LoadSceneFile(pts)
if scene.pinhole: pt.SetCameraAperture(0.0)
pt.SetSeed(1); pt.RenderFrames(1)
ctx = pt.context()
for kv in os.environ.get("PTK_OPTS", "").split(","):
    if "=" in kv:
        k, v = kv.split("="); ctx.set_option(k, float(v))
if any(k in os.environ.get("PTK_OPTS", "") for k in ("device_build", "bvh_")):
    pt.BuildBVH(); pt.RenderFrames(1)                 # builder options take effect at the next upload
print("bvh", ctx.bvh_info(), ctx.bvh_layout())
ctx.set_option("overlap", 0)
for rep in range(int(os.environ.get("PTK_PROBE_REPS", "3"))):
    ctx.reset(); t0 = time.time(); ctx.render(0, spp, 1); ctx.synchronize(); t1 = time.time()
    tm, am = ctx.last_kernel_ms()
    W, H = pt.GetResolution()
    print(f"{name} spp {spp}: wall {1e3*(t1-t0):.1f} ms trace {tm:.1f} acc {am:.2f} -> {W*H*spp/(t1-t0)/1e6:.0f} Msamples/s", flush=True)
st = ctx.collect_stats(0, min(spp, 16), 1)
s = st["samples"]
print({k: round(v / s, 3) for k, v in st.items() if k not in ("samples", "max_walk_nodes")}, "max nodes in one walk:", st["max_walk_nodes"])
