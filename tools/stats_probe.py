"""Scratch: utilisation statistics + timing for a BASELINE config through the host API."""
import sys, os, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pbrpathtracer_amd import scenes as S
from pbrpathtracer_amd.pathtracer import PathTracer
cfg = sys.argv[1]; spp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
pts, scene, _ = S.build_config(cfg, tempfile.mkdtemp())
pt = PathTracer(0); pt.LoadSceneFile(pts)
if scene.pinhole: pt.SetCameraAperture(0.0)
pt.RenderFrames(1)
c = pt.context()
for kv in os.environ.get('PTK_OPTS','').split(','):
    if '=' in kv:
        k,v=kv.split('='); c.set_option(k, float(v))
W,H = pt.GetResolution()
if os.environ.get('PTK_TILE'):
    r,w = map(int, os.environ['PTK_TILE'].split(',')); c.set_tile(r,w); print('tile',r,w)
best=1e9
for rep in range(3):
    c.reset(); c.render(0, spp, 1); c.synchronize(); tm, am = c.last_kernel_ms(); best=min(best,tm+am)
st = c.collect_stats(0, int(os.environ.get("PTK_STATS_SPP", min(spp,16))), 1); s = st["samples"]
print(f"{cfg} spp{spp} opts[{os.environ.get('PTK_OPTS','')}]: {best:.1f} ms -> {W*H*spp/best/1e3:.0f} Msamples/s | util walk %.3f shade %.3f gen %.3f | per 64 samples: walk iters %.1f shade %.2f gen %.2f | nodes %.1f tris %.1f rays %.2f | tri arm util %.3f execs/64 %.1f" % (
    st["walk_lane_iters"]/max(1,st["walk_wave_iters"])/64, st["shade_lanes"]/max(1,st["shade_wave_execs"])/64, st["gen_lanes"]/max(1,st["gen_wave_execs"])/64,
    st["walk_wave_iters"]*64/s, st["shade_wave_execs"]*64/s, st["gen_wave_execs"]*64/s, st["node_visits"]/s, st["tri_tests"]/s, st["rays"]/s, st["tri_lanes"]/max(1,st["tri_wave_execs"])/64, st["tri_wave_execs"]*64/s))
