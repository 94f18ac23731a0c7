"""Scratch performance probe (GPU box): Cornell fixture scene at the C2 shape."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pbrpathtracer_amd import ptk
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
kind = sys.argv[1] if len(sys.argv) > 1 else "cornell"
W, H, D = int(sys.argv[2]) if len(sys.argv) > 2 else 1280, int(sys.argv[3]) if len(sys.argv) > 3 else 720, int(sys.argv[4]) if len(sys.argv) > 4 else 8
spp = int(sys.argv[5]) if len(sys.argv) > 5 else 64
z = np.load(os.path.join(G, f"tier_s_{kind}.npz"))
arr = {k[6:]: z[k] for k in z.files if k.startswith("scene_")}
c = ptk.Context(0)
c.upload_scene(arr)
cam = z["cam"]; proj = z["proj"]
ap = float(os.environ.get("PTK_APERTURE", float(z["aperture"])))
c.set_camera(cam[0:3], cam[3:6], cam[6:9], float(proj[0]), float(proj[1]), float(z["focal_dist"]), ap)
c.set_frame(W, H, D)
for kv in os.environ.get('PTK_OPTS','').split(','):
    if '=' in kv:
        k,v=kv.split('='); c.set_option(k, float(v))
c.reset()
c.render(0, 4, 1); c.synchronize()
for rep in range(3):
    c.reset()
    t0 = time.time(); c.render(0, spp, 1); c.synchronize(); t1 = time.time()
    ms, n = c.last_render_ms(); tm, am = c.last_kernel_ms()
    print(f"{kind} {W}x{H} D{D} spp{spp}: wall {1e3*(t1-t0):.1f} ms, kernels {ms:.1f} ms (trace {tm:.2f} + accumulate {am:.2f}) -> {W*H*spp/ms/1e3:.1f} Msamples/s")
st = c.collect_stats(0, min(spp, 32), 1)
print(st)
s = st["samples"]
print("per sample: rays %.2f shadow %.2f nodes %.1f tris %.1f shaded %.2f tex %.2f" % tuple(st[k]/s for k in ("rays","shadow_rays","node_visits","tri_tests","hits_shaded","tex_fetches")))
print("util: walk %.3f shade %.3f gen %.3f | per 64 samples: walk iters %.1f, shade execs %.2f, gen execs %.2f" % (
    st["walk_lane_iters"]/max(1,st["walk_wave_iters"])/64, st["shade_lanes"]/max(1,st["shade_wave_execs"])/64,
    st["gen_lanes"]/max(1,st["gen_wave_execs"])/64, st["walk_wave_iters"]*64/s, st["shade_wave_execs"]*64/s, st["gen_wave_execs"]*64/s))
print("bvh", c.bvh_info())
