"""Scratch probe: node visits / triangle tests per sample of a config with the host-built and the device-built tree."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pbrpathtracer_amd import scenes as S
from pbrpathtracer_amd.pathtracer import PathTracer
for name in sys.argv[1:] or ["C4"]:
    pts, scene, _ = S.build_config(name, tempfile.mkdtemp())
    for dev in (0, 1):
        pt = PathTracer(0); pt.context().set_option("device_build", dev)
        pt.LoadSceneFile(pts)
        if scene.pinhole: pt.SetCameraAperture(0.0)
        pt.SetSeed(1); pt.RenderFrames(1)
        c = pt.context()
        st = c.collect_stats(0, 16, 1)
        c.set_option("overlap", 0)
        ts = []
        for _ in range(3):
            c.reset(); c.render(0, 64, 1); c.synchronize(); ts.append(c.last_kernel_ms()[0])
        print(f"{name} {'device' if dev else 'host  '}: nodes {c.bvh_info()} stack {c.bvh_layout()[2]} visits/sample {st['node_visits']/st['samples']:.2f} tris {st['tri_tests']/st['samples']:.2f} trace {min(ts):.2f} ms {c.upload_timing()}")
        pt.close()
