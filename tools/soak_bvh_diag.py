"""GPU box: for one tools/soak_bvh.py seed, the rays whose closest hit differs between the two trees, judged by the oracle's brute-force loop."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
from pbrpathtracer_amd import ptk
from oracle import oracle_binding as OB
from test_gpu_bvh_build import _scene
seed = int(sys.argv[1]); k = seed - int(sys.argv[2]) if len(sys.argv) > 2 else seed
kinds = ["soup", "clustered", "identical", "line", "plane grid", "two far clusters", "huge and tiny", "duplicates", "big coordinates", "slivers"]
rng = np.random.default_rng(seed)
kind = kinds[k % len(kinds)]
n = int(rng.choice([4096, 5000, 12345, 40000]))
c = rng.uniform(-1, 1, (n, 1, 3)); size = 0.02
assert kind == "two far clusters"
c = c * 1e-3 + np.where(rng.uniform(0, 1, (n, 1, 1)) < 0.5, -1e3, 1e3)
v = c + size * rng.uniform(-1, 1, (n, 3, 3))
verts = v.astype(np.float32).reshape(n, 9)
arrays = _scene(verts)
ext = float(np.abs(verts).max())
ro = (rng.uniform(-1.5, 1.5, (3000, 3)) * ext).astype(np.float32)
tgt = verts.reshape(n, 3, 3)[rng.integers(0, n, 3000)].mean(axis=1)
rd = (tgt - ro); rd /= np.maximum(np.linalg.norm(rd, axis=1, keepdims=True), 1e-30); rd = rd.astype(np.float32)
rd[::40, 1] = 0.0
ctx = ptk.Context(0)
from bvh_check import check_bvh
ctx.set_option("device_build", 0); ctx.upload_scene(arrays); tri_h, tuv_h = ctx.probe_hits(ro, rd)
try:
    print("host tree:", check_bvh(*ctx.download_bvh(), verts))
except AssertionError as e:
    print("host tree INVALID:", e)
ctx.set_option("device_build", 1); ctx.upload_scene(arrays); tri_d, tuv_d = ctx.probe_hits(ro, rd)
try:
    print("device tree:", check_bvh(*ctx.download_bvh(), verts))
except AssertionError as e:
    print("device tree INVALID:", e)
o = OB.Oracle(arrays)
bad = np.nonzero((tri_h != tri_d) | (tuv_h != tuv_d).any(axis=1))[0]
print(kind, "n", n, "differing rays:", len(bad))
for j in bad[:12]:
    h, t, uvw = o.hit(ro[j], rd[j], brute=True)
    print(f"ray {j} rd {rd[j]} | host tri {tri_h[j]} t {tuv_h[j][0]!r} | device tri {tri_d[j]} t {tuv_d[j][0]!r} | brute force: hit {h} tri {t} t {uvw[0]!r}")
