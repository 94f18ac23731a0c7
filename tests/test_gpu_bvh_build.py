"""N2: the BVH built ON THE GPU (pbrpathtracer_amd/csrc/bvh_device.hip: level-synchronous binned SAH ->
4-wide quantised nodes -> device-side record packing) behind ptk_upload_scene, against the host builder as comparator.
Replaces BVHNode::Construct (reference mesh.cpp:169-211, pathtracer.cpp:260-274); closest hits do not depend on the tree,
so everything that reaches the image must be bit-identical between the two builders."""
import time

import numpy as np
import pytest

from bvh_check import check_bvh
from conftest import load_golden, scene_from_golden

pytestmark = pytest.mark.gpu


def _soup(n, size, seed, clustered=False):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-1, 1, (n, 1, 3))
    if clustered:
        c = c ** 3                                        # dense in the middle, sparse outside
    v = (c + size * rng.uniform(-1, 1, (n, 3, 3))).astype(np.float32).reshape(n, 9)
    return v


def _scene(verts):
    from pbrpathtracer_amd import ptk
    n = len(verts)
    mats = np.zeros(1, ptk.MATERIAL_DTYPE)
    mats["diffuse"] = 0.7; mats["specular"] = 1.0; mats["emissive_intensity"] = 1.0; mats["roughness"] = 1.0
    mats["translucency"] = 1.0; mats["ior"] = 1.5; mats["tex"] = -1
    e1 = verts[:, 3:6] - verts[:, 0:3]; e2 = verts[:, 6:9] - verts[:, 0:3]
    nrm = np.cross(e1, e2); nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-20)
    tbn = np.concatenate([nrm, np.zeros((n, 6))], axis=1).astype(np.float32)
    return dict(verts=verts, normals=np.tile(nrm, (1, 3)).astype(np.float32), uvs=np.zeros((n, 6), np.float32), tbn=tbn,
                smoothing=np.zeros(n, np.uint8), material=np.zeros(n, np.int32), materials=mats, lights=np.zeros(0, np.int32))


@pytest.fixture(scope="module")
def ctx():
    from pbrpathtracer_amd import ptk
    c = ptk.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("name,verts", [
    ("soup 50k", _soup(50000, 0.02, 1)),
    ("clustered soup 300k", _soup(300000, 0.004, 2, clustered=True)),
    ("identical 6000", np.tile(_soup(1, 0.3, 3), (6000, 1))),
    ("thin slab soup 20k", (_soup(20000, 0.05, 4) * np.array([1, 0.02, 1] * 3, np.float32))),
])
def test_device_built_tree_is_valid_and_gives_the_same_hits(ctx, name, verts):
    arrays = _scene(verts)
    rng = np.random.default_rng(9)
    ro = rng.uniform(-1.5, 1.5, (4000, 3)).astype(np.float32)
    rd = rng.normal(0, 1, (4000, 3)).astype(np.float32); rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    rd[::50, 0] = 0.0; rd[::70, 1] = 0.0                                    # axis-parallel rays too
    ctx.set_option("device_build", 0)
    t0 = time.time(); ctx.upload_scene(arrays); t_host = time.time() - t0
    assert not ctx.upload_timing()["built_on_device"]
    tri_h, tuv_h = ctx.probe_hits(ro, rd)
    info_h = check_bvh(*ctx.download_bvh(), verts)
    ctx.set_option("device_build", 1)
    t0 = time.time(); ctx.upload_scene(arrays); t_dev = time.time() - t0
    tm = ctx.upload_timing()
    assert tm["built_on_device"], "the device builder fell back to the host"
    nodes, order = ctx.download_bvh()
    info_d = check_bvh(nodes, order, verts)
    assert info_d["stack_need"] == ctx.bvh_layout()[2] and info_d["depth"] == ctx.bvh_info()[1]
    tri_d, tuv_d = ctx.probe_hits(ro, rd)
    assert (tri_h >= 0).sum() > (100 if "soup" in name else 0)
    assert np.array_equal(tri_h, tri_d) and np.array_equal(tuv_h, tuv_d)      # tree-independent closest hit, bit for bit
    # a second build gives the same tree (splits are decided from integer histograms: order-independent); only the
    # numbering of the nodes inside a level follows the order in which their parents' atomics landed
    ctx.upload_scene(arrays)
    nodes2, order2 = ctx.download_bvh()
    assert check_bvh(nodes2, order2, verts) == info_d
    assert sorted(map(bytes, nodes[:, [0, 1, 2, 3, 4, 5, 10, 11, 12, 13, 14, 15]].view(np.uint32))) == \
        sorted(map(bytes, nodes2[:, [0, 1, 2, 3, 4, 5, 10, 11, 12, 13, 14, 15]].view(np.uint32)))     # same boxes, links aside
    print(f"{name}: host {t_host*1e3:.0f} ms {info_h}, device {t_dev*1e3:.0f} ms {tm} {info_d}")


def test_device_build_renders_bit_identically_and_is_faster_at_1M(tmp_path):
    """C5's million-triangle scene through the drop-in API with either builder: same accumulator, bit for bit; the device
    build must not be slower than the host's, and its tree must cost a ray no more than 15 % more node visits."""
    from pbrpathtracer_amd import scenes as S
    from pbrpathtracer_amd.pathtracer import PathTracer
    pts, scene, _ = S.build_config("C5", str(tmp_path))
    out = {}
    for dev in (0, 1):
        pt = PathTracer(0)
        pt.context().set_option("device_build", dev)
        t0 = time.time(); pt.LoadSceneFile(pts); t_load = time.time() - t0
        pt.SetCameraAperture(0.0); pt.SetSeed(4)
        tm = pt.context().upload_timing()
        assert tm["built_on_device"] == bool(dev)
        pt.RenderFrames(8)
        assert pt.LastError() == ""
        st = pt.context().collect_stats(0, 8, 4)
        out[dev] = (pt.ReadAccumulation(), t_load, tm, st["node_visits"] / st["samples"], st["tri_tests"] / st["samples"])
        pt.close()
    assert np.array_equal(out[0][0], out[1][0])
    print("host  : load %.3f s %s, node visits / sample %.2f, triangle tests %.2f" % out[0][1:])
    print("device: load %.3f s %s, node visits / sample %.2f, triangle tests %.2f" % out[1][1:])
    assert out[1][2]["bvh_ms"] < out[0][2]["bvh_ms"]
    assert out[1][3] <= 1.15 * out[0][3]


@pytest.mark.parametrize("seed", [15, 295, 215])
def test_far_origins_and_tiny_nodes_give_tree_independent_hits(ctx, oracle_mod, seed):
    """Two clusters of 1e-3 at +-1e3, rays from up to 3000 units away aimed at 0.02-sized triangles (tools/soak_bvh.py, the
    cases that failed before): the slab distances' absolute error (~6e-8 x the distance between ray origin and node) exceeds
    the 8-bit grid step of the deep nodes, and Moeller-Trumbore's own error in t decides which of two nearly equidistant
    triangles is the closest.  The box test carries both as explicit slack (walk_step), so the host-built tree, the
    device-built tree and the oracle's brute-force loop over all triangles agree on every ray."""
    rng = np.random.default_rng(seed)
    n = int(rng.choice([4096, 5000, 12345, 40000]))
    c = rng.uniform(-1, 1, (n, 1, 3))
    c = c * 1e-3 + np.where(rng.uniform(0, 1, (n, 1, 1)) < 0.5, -1e3, 1e3)
    verts = (c + 0.02 * rng.uniform(-1, 1, (n, 3, 3))).astype(np.float32).reshape(n, 9)
    arrays = _scene(verts)
    ext = float(np.abs(verts).max())
    ro = (rng.uniform(-1.5, 1.5, (3000, 3)) * ext).astype(np.float32)
    tgt = verts.reshape(n, 3, 3)[rng.integers(0, n, 3000)].mean(axis=1)
    rd = tgt - ro; rd /= np.maximum(np.linalg.norm(rd, axis=1, keepdims=True), 1e-30); rd = rd.astype(np.float32)
    rd[::40, 1] = 0.0
    ctx.set_option("device_build", 0); ctx.upload_scene(arrays); tri_h, tuv_h = ctx.probe_hits(ro, rd)
    check_bvh(*ctx.download_bvh(), verts)
    ctx.set_option("device_build", 1); ctx.upload_scene(arrays); tri_d, tuv_d = ctx.probe_hits(ro, rd)
    check_bvh(*ctx.download_bvh(), verts)
    assert (tri_h >= 0).mean() > 0.9
    assert np.array_equal(tri_h, tri_d) and np.array_equal(tuv_h, tuv_d)
    o = oracle_mod.Oracle(arrays)
    for j in range(0, 3000, 7):
        h, t, v = o.hit(ro[j], rd[j], brute=True)
        assert (t if h else -1) == tri_d[j], j
        if h: assert np.array_equal(v, tuv_d[j]), j
    o.close()
