"""GPU box: traversal microbenchmark behind DESIGN.md 5(g) - the product's 4-wide quantised tree against an 8-wide compressed one
built FROM it (children absorbed largest-surface-first until a node holds eight), same triangles, same incoherent closest-hit
rays, outside the megakernel (tools/microbench/bvh_width.hip).
    python tools/bvh_width_probe.py [C4|C5|C3] [million rays]"""
import json, os, struct, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pbrpathtracer_amd import scenes as S
from pbrpathtracer_amd.pathtracer import PathTracer

name = sys.argv[1] if len(sys.argv) > 1 else "C4"
mrays = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
t0 = time.time()
pts, scene, _ = S.build_config(name, tempfile.mkdtemp())
pt = PathTracer(0); pt.LoadSceneFile(pts); pt.RenderFrames(1)
ctx = pt.context()
nodes, order = ctx.download_bvh()
verts = np.ascontiguousarray(pt.StagedScene()["verts"], np.float32).reshape(-1, 3, 3)
n = len(verts); N = len(nodes)
print(f"{name}: {n} triangles, {N} 4-wide nodes ({time.time() - t0:.1f} s)", flush=True)

# ---- the 4-wide tree, decoded -----------------------------------------------------------------------------------------------
raw = nodes.view(np.uint32)
link = raw[:, 6:10].view(np.int32)
lo4 = np.stack([(raw[:, 10 + a][:, None] >> (8 * np.arange(4))) & 255 for a in range(3)], axis=-1)
hi4 = np.stack([(raw[:, 13 + a][:, None] >> (8 * np.arange(4))) & 255 for a in range(3)], axis=-1)
valid = (lo4 <= hi4).all(axis=2)
# true (padded) boxes, bottom-up: links point forward, so reverse index order sees children first
ext = float(np.abs(verts).max())
pad = np.float32(1e-5 * max(ext, 1.0))
tri_lo = verts.min(axis=1) - pad; tri_hi = verts.max(axis=1) + pad
tl = tri_lo[order]; th = tri_hi[order]                      # in leaf order
cb_lo = np.full((N, 4, 3), np.inf, np.float32); cb_hi = np.full((N, 4, 3), -np.inf, np.float32)     # child boxes
nb_lo = np.full((N, 3), np.inf, np.float32); nb_hi = np.full((N, 3), -np.inf, np.float32)           # node boxes
for i in range(N - 1, -1, -1):
    for k in range(4):
        if not valid[i, k]: continue
        lk = int(link[i, k])
        if lk >= 0: cb_lo[i, k] = nb_lo[lk]; cb_hi[i, k] = nb_hi[lk]
        else:
            code = ~lk; f, c = code >> 3, (code & 7) + 1
            cb_lo[i, k] = tl[f:f + c].min(axis=0); cb_hi[i, k] = th[f:f + c].max(axis=0)
    nb_lo[i] = cb_lo[i][valid[i]].min(axis=0); nb_hi[i] = cb_hi[i][valid[i]].max(axis=0)
print(f"boxes done ({time.time() - t0:.1f} s)", flush=True)


def area(lo, hi):
    d = np.maximum(hi - lo, 0.0)
    return float(d[0] * d[1] + d[1] * d[2] + d[2] * d[0])


def kids(i):
    """children of 4-wide node i: (is_leaf, link, lo, hi)"""
    return [(int(link[i, k]) < 0, int(link[i, k]), cb_lo[i, k], cb_hi[i, k]) for k in range(4) if valid[i, k]]


def bf16_up(x):
    """smallest bf16 value >= x (x > 0), as its 16 high bits"""
    b = np.float32(x).view(np.uint32)
    h = int(b) >> 16
    if (h << 16) < int(b): h += 1
    return h


# ---- collapse to 8 wide, breadth first: a node's interior children get consecutive indices, its leaves' triangles a contiguous range ----
recs = []                  # 20 dwords per node
new_order = []             # leaf order of the 8-wide tree -> index into the 4-wide tree's leaf order
queue = [(0, None)]        # (4-wide node whose children seed the 8-wide node, unused)
next_index = 1
depth_of = {0: 1}; max_depth = 1
qi = 0
while qi < len(queue):
    seed, _ = queue[qi]; my_index = qi; qi += 1
    ch = kids(seed)
    while len(ch) < 8:
        best, best_a = -1, -1.0
        for j, (leaf, lk, lo, hi) in enumerate(ch):
            if leaf: continue
            if len(ch) - 1 + int(valid[lk].sum()) > 8: continue
            a = area(lo, hi)
            if a > best_a: best_a, best = a, j
        if best < 0: break
        lk = ch[best][1]
        ch = ch[:best] + ch[best + 1:] + kids(lk)
    lo_u = np.min([c[2] for c in ch], axis=0); hi_u = np.max([c[3] for c in ch], axis=0)
    axis = int(np.argmax(hi_u - lo_u))
    ch.sort(key=lambda c: float(c[2][axis] + c[3][axis]))
    sc = []
    for a in range(3):
        s_ = max((float(hi_u[a]) - float(lo_u[a])) / 255.0 * (1 + 1e-6), 1e-30)
        hb = bf16_up(s_)
        while float(lo_u[a]) + 255.0 * float(np.uint32(hb << 16).view(np.float32)) < float(hi_u[a]): hb += 1
        sc.append(hb)
    scf = [float(np.uint32(h_ << 16).view(np.float32)) for h_ in sc]
    lo_b = np.full((3, 8), 255, np.uint32); hi_b = np.zeros((3, 8), np.uint32)
    imask = 0; cnt16 = 0
    child_base = next_index; tri_base = len(new_order)
    d = depth_of[my_index]
    for s_, (leaf, lk, lo, hi) in enumerate(ch):
        for a in range(3):
            o = float(lo_u[a]); q = scf[a]
            ql = int(np.floor((float(lo[a]) - o) / q)); qh = int(np.ceil((float(hi[a]) - o) / q))
            ql = min(max(ql, 0), 255); qh = min(max(qh, 0), 255)
            while ql > 0 and o + ql * q > float(lo[a]): ql -= 1
            while qh < 255 and o + qh * q < float(hi[a]): qh += 1
            lo_b[a, s_] = ql; hi_b[a, s_] = qh
        if leaf:
            code = ~lk; f, c = code >> 3, (code & 7) + 1
            assert c <= 4
            cnt16 |= (c - 1) << (2 * s_)
            new_order.extend(range(f, f + c))
        else:
            imask |= 1 << s_
            queue.append((lk, None)); depth_of[next_index] = d + 1; max_depth = max(max_depth, d + 1); next_index += 1
    w = np.zeros(20, np.uint32)
    w[0:3] = lo_u.astype(np.float32).view(np.uint32)
    w[3] = sc[0] | (sc[1] << 16)
    w[4] = sc[2] | (imask << 16) | (axis << 24)
    w[5] = child_base; w[6] = tri_base; w[7] = cnt16
    for a in range(3):
        w[8 + 2 * a] = sum(int(lo_b[a, k]) << (8 * k) for k in range(4)); w[9 + 2 * a] = sum(int(lo_b[a, 4 + k]) << (8 * k) for k in range(4))
        w[14 + 2 * a] = sum(int(hi_b[a, k]) << (8 * k) for k in range(4)); w[15 + 2 * a] = sum(int(hi_b[a, 4 + k]) << (8 * k) for k in range(4))
    recs.append(w)
N8 = len(recs)
assert sorted(new_order) == list(range(n)), "the 8-wide tree lost or duplicated triangles"
print(f"8-wide tree: {N8} nodes ({N / N8:.2f} x fewer), depth {max_depth} ({time.time() - t0:.1f} s)", flush=True)

# ---- triangle records (ptk_device.h) in both leaf orders ----------------------------------------------------------------------
def tri_records(idx):
    v = verts[idx]
    r = np.zeros((len(idx), 12), np.float32)
    r[:, 0:3] = v[:, 0]; r[:, 3:6] = v[:, 1] - v[:, 0]; r[:, 6:9] = v[:, 2] - v[:, 0]
    r[:, 9] = np.asarray(idx, np.int32).view(np.float32); r[:, 10] = np.int32(-1).view(np.float32)
    return r
rec4 = tri_records(order)
rec8 = tri_records(order[np.asarray(new_order)])

# ---- incoherent rays: from random surface points into the hemisphere (what bounce rays are) ----------------------------------
rng = np.random.default_rng(7)
R = int(mrays * 1e6)
t = rng.integers(0, n, R)
b = rng.uniform(size=(R, 2)).astype(np.float32); flip = b.sum(axis=1) > 1; b[flip] = 1 - b[flip]
v = verts[t]
p = v[:, 0] + (v[:, 1] - v[:, 0]) * b[:, :1] + (v[:, 2] - v[:, 0]) * b[:, 1:]
nrm = np.cross(v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]); nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-30)
d = rng.normal(size=(R, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
d = np.where((d * nrm).sum(axis=1, keepdims=True) < 0, -d, d)
rays = np.concatenate([p + nrm * np.float32(1e-4 * ext), d], axis=1).astype(np.float32)
scene_bound = np.float32(3.1 * (1.01 * ext + 1e-3))

path = os.path.join(tempfile.mkdtemp(), "scene.bin")
with open(path, "wb") as f:
    f.write(struct.pack("<8q", 0x3848564257, N, N8, n, R, max_depth, 0, 0))
    f.write(struct.pack("<f", float(scene_bound)) + b"\0" * 60)
    f.write(np.ascontiguousarray(nodes, np.float32).tobytes()); f.write(rec4.tobytes())
    f.write(np.stack(recs).astype(np.uint32).tobytes()); f.write(rec8.tobytes()); f.write(rays.tobytes())
exe = os.path.join(ROOT, "tools", "microbench", "bvh_width")
out = subprocess.run([exe, path], capture_output=True, text=True)
print(out.stdout.strip() or out.stderr[-800:])
os.remove(path)
if out.returncode == 0:
    r = json.loads(out.stdout.strip().splitlines()[-1]); r["config"] = name; r["triangles"] = n
    print(json.dumps(r))
sys.exit(out.returncode)
