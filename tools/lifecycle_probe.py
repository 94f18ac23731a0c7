"""GPU box, one-off: tracers created, loaded, rendered ASYNCHRONOUSLY and destroyed at once (renders still in flight), many times over,
for three scene sizes - no crash, no hang, and the device memory in use returns to where it started (no leak)."""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from pbrpathtracer_amd import scenes as S
from pbrpathtracer_amd.pathtracer import PathTracer
tmp = tempfile.mkdtemp()
cfgs = [("C1", dict(width=256, height=192)), ("C4", dict(width=320, height=200)), ("C3", dict(width=200, height=120))]
built = []
for name, kw in cfgs:
    d = os.path.join(tmp, name); os.makedirs(d); built.append(S.build_config(name, d, **kw)[0])
torch.cuda.init()
def used(): f, t = torch.cuda.mem_get_info(0); return (t - f) / 2**20
for b in built:                                                         # (first use of every path: code objects, the runtime's pools)
    pt = PathTracer(0); pt.LoadSceneFile(b); pt.RenderFrames(9); pt.SetResolution((150, 48)); pt.ResetImage(); pt.RenderFrames(2); pt.close()
base = used(); t0 = time.time()
for k in range(int(sys.argv[1]) if len(sys.argv) > 1 else 90):
    pt = PathTracer(0); pt.LoadSceneFile(built[k % 3]); pt.SetSeed(k)
    for _ in range(1 + k % 4): pt.RenderFrames(4 + k % 5)           # asynchronous: nothing waits
    if k % 7 == 3: pt.Exit()
    if k % 5 == 2: pt.SetResolution((64 + k, 48)); pt.ResetImage(); pt.RenderFrames(2)
    pt.close()                                                          # renders may still be running
    if k % 30 == 29: print(f"{k + 1} tracers, device memory in use {used():.0f} MiB (start {base:.0f}) [{time.time() - t0:.0f} s]", flush=True)
end = used()
print(f"memory in use: {base:.0f} -> {end:.0f} MiB")
sys.exit(0 if end - base < 64 else 1)
