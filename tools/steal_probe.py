"""GPU box, one-off measurement: how many node-idle lanes of a walk iteration could test an interior entry taken from
ANOTHER lane's traversal stack.  Needs a scratch build of the library whose STATS kernel counts it in three hijacked counters
(tex_fetches, paths_started, hits_shaded - so only for scenes without textures):
   git archive HEAD | tar -x -C /tmp/t && patch -d /tmp/t -p0 < tools/experiments/steal_potential.patch   (paths: pbrpathtracer_amd/csrc/ptk_kernels.hip)
   make -C /tmp/t/pbrpathtracer_amd/csrc OUT=$PWD/pbrpathtracer_amd/libptk_S.so
   gpurun -- 'PTK_DEV_TOOLS=1 PTK_LIB_PATH=$PWD/pbrpathtracer_amd/libptk_S.so python3 tools/steal_probe.py C4 C3 C5'"""
import sys, os, tempfile
sys.path.insert(0, os.getcwd())
import torch
from pbrpathtracer_amd import scenes as S
from pbrpathtracer_amd.pathtracer import PathTracer
for cfg in sys.argv[1:]:
    pts, scene, _ = S.build_config(cfg, tempfile.mkdtemp())
    pt = PathTracer(0); pt.LoadSceneFile(pts)
    if scene.pinhole: pt.SetCameraAperture(0.0)
    pt.RenderFrames(1); c = pt.context()
    st = c.collect_stats(0, 8, 1)
    it = st["walk_wave_iters"]
    print(cfg, "walk iters", it, "node lanes/iter %.1f" % (st["node_visits"]/it), "walking lanes/iter %.1f" % (st["walk_lane_iters"]/it),
          "| extra node lanes/iter: ranked matching %.1f; fixed partner lane^32 %.1f, lane^1 %.1f, lane^8 %.1f (pinhole configs only)" % (st["tex_fetches"]/it, st["paths_started"]/it, st["hits_shaded"]/it, st["gen_lanes"]/it), flush=True)
    pt.close()
