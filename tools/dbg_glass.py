import sys, os
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from conftest import load_golden, scene_from_golden
from pbrpathtracer_amd import ptk
kind = sys.argv[1] if len(sys.argv) > 1 else "glass"
z = load_golden(f"tier_s_{kind}.npz")
W, H, D, nref = int(z["width"]), int(z["height"]), int(z["depth"]), int(z["spp"])
c = ptk.Context(0); c.upload_scene(scene_from_golden(z))
cam, proj = z["cam"], z["proj"]
c.set_camera(cam[0:3], cam[3:6], cam[6:9], float(proj[0]), float(proj[1]), float(z["focal_dist"]), float(z["aperture"]))
c.set_frame(W, H, D)
def run(seed, spp):
    c.reset(); c.render(0, spp, seed); return c.read_accum() / spp
spp = 32768
g1, g2 = run(1, spp), run(2, spp)
ref = z["mean"]; h1, h2 = z["mean_half1"], z["mean_half2"]
rm = lambda a, b: float(np.sqrt(np.mean((a - b) ** 2)))
print("gpu1-gpu2", rm(g1, g2), " ref half1-half2", rm(h1, h2), "(each half", nref // 2, "spp)")
print("gpu1-ref", rm(g1, ref), "gpu2-ref", rm(g2, ref))
print("global means gpu", g1.reshape(-1, 3).mean(0), "ref", ref.reshape(-1, 3).mean(0))
d = (g1 - ref).mean(2)
np.set_printoptions(linewidth=250, precision=1, suppress=True)
print("diff x1000 (rows bottom-up), every 2nd pixel"); print((d[::2, ::2] * 1000).round(0).astype(int))
os.makedirs(os.path.join(R, "gpurun_out"), exist_ok=True)
np.save(os.path.join(R, "gpurun_out", f"{kind}_gpu_mean.npy"), g1)
