"""Scratch probe: where the scene-load time of a config goes (host stages of LoadSceneFile)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pbrpathtracer_amd import scenes as S
from pbrpathtracer_amd.pathtracer import PathTracer
name = sys.argv[1] if len(sys.argv) > 1 else "C5"
t0 = time.time(); pts, scene, _ = S.build_config(name, tempfile.mkdtemp()); print(f"{name}: scene synthesis {time.time()-t0:.3f} s")
for rep in range(2):
    pt = PathTracer(0)
    t0 = time.time(); pt.LoadSceneFile(pts); t_all = time.time() - t0
    print(f"LoadSceneFile {t_all:.3f} s: upload stages {pt.context().upload_timing()}, triangles {pt.GetTriangleCount()}")
    pt.close()
pt = PathTracer(0)
t = {}
for o in scene.objects:
    t0 = time.time(); pt.LoadObject(o.obj_path); t[os.path.basename(o.obj_path)] = round(time.time() - t0, 3)
t0 = time.time(); pt.StagedScene(); t["flatten"] = round(time.time() - t0, 3)
t0 = time.time(); pt.BuildBVH(); t["BuildBVH(flatten+upload)"] = round(time.time() - t0, 3)
print("pieces", t, pt.context().upload_timing())
