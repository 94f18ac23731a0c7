"""Seeded random scenes through every branch of PathTracer::Trace (pathtracer.cpp:551-727) at once: diffuse / mirror / rough-lobe
reflection, glass with and without roughness, translucency, emissive surfaces that are also lights, smoothed normals, normal /
diffuse / roughness / metalness / emissive / opacity textures, thin-lens and pinhole cameras, tiny to mid-size triangle counts
(FLAT kernel, host-built and device-built BVH).  HIP path vs the oracle, accumulators and RGB8 bit for bit.  The reference has
no tests of its own; the golden micro scenes pin the oracle to it, this widens the GPU <-> oracle side."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from pbrpathtracer_amd import ptk
    c = ptk.Context(0)
    yield c
    c.close()


def _unit(v):
    return v / np.maximum(np.linalg.norm(v, axis=-1, keepdims=True), 1e-20)


def random_scene(seed: int, n_tris: int, with_textures: bool):
    from pbrpathtracer_amd import ptk
    rng = np.random.default_rng(seed)
    # a back wall and a floor (so that every camera ray starts a path), big random triangles around, a soup of small ones inside
    n_room = min(12, n_tris)
    quad = lambda a, b, c, d: [[a, b, c], [a, c, d]]
    fixed = np.array(quad([-4, -3, -2.2], [4, -3, -2.2], [4, 3, -2.2], [-4, 3, -2.2]) + quad([-4, -1.6, -2.2], [-4, -1.6, 3.5], [4, -1.6, 3.5], [4, -1.6, -2.2]), np.float64)
    c_room = rng.uniform(-1.0, 1.0, (n_room, 1, 3)) * np.array([1.0, 1.0, 0.2]) + np.array([0.0, 0.0, -1.2]) * rng.integers(0, 2, (n_room, 1, 1))
    room = c_room + rng.uniform(-2.0, 2.0, (n_room, 3, 3))
    room[:min(4, n_room)] = fixed[:min(4, n_room)] + rng.uniform(-0.05, 0.05, (min(4, n_room), 3, 3))
    n_soup = n_tris - n_room
    soup = rng.uniform(-1.0, 1.0, (n_soup, 1, 3)) + rng.uniform(-1.0, 1.0, (n_soup, 3, 3)) * (0.9 / max(1.0, n_soup ** (1.0 / 3.0)))
    verts = np.concatenate([room, soup]).astype(np.float32)
    n = len(verts)
    e1, e2 = verts[:, 1] - verts[:, 0], verts[:, 2] - verts[:, 0]
    fn = _unit(np.cross(e1, e2)).astype(np.float32)
    tg = _unit(e1).astype(np.float32)
    bt = _unit(np.cross(fn, tg)).astype(np.float32)
    vn = _unit(fn[:, None, :] + 0.4 * rng.normal(0, 1, (n, 3, 3))).astype(np.float32)
    n_mats = 10
    mats = np.zeros(n_mats, ptk.MATERIAL_DTYPE)
    mats["type"] = rng.integers(0, 2, n_mats)
    mats["diffuse"] = rng.uniform(0.2, 0.95, (n_mats, 3))
    mats["specular"] = rng.uniform(0.3, 1.0, (n_mats, 3))
    mats["emissive"] = rng.uniform(0.0, 1.0, (n_mats, 3)) * (rng.uniform(0, 1, (n_mats, 1)) < 0.3)
    mats["emissive_intensity"] = rng.uniform(1.0, 6.0, n_mats)
    mats["roughness"] = rng.choice([0.0, 0.25, 0.7, 1.0], n_mats)
    mats["reflectiveness"] = rng.choice([0.0, 0.4, 1.0], n_mats)
    mats["translucency"] = rng.choice([0.0, 0.6, 1.0], n_mats)
    mats["ior"] = rng.uniform(1.1, 1.9, n_mats)
    mats["tex"] = -1
    mats[0]["type"] = 0; mats[0]["reflectiveness"] = 0.0; mats[0]["emissive"] = (1.0, 0.9, 0.8); mats[0]["emissive_intensity"] = 8.0   # a sure light
    textures = np.zeros(0, ptk.TEXTURE_DTYPE); texels = np.zeros(0, np.uint8)
    if with_textures:
        sizes = [(8, 8), (16, 4), (5, 7), (32, 32), (3, 3), (9, 2)]
        textures = np.zeros(len(sizes), ptk.TEXTURE_DTYPE); chunks = []; off = 0
        for k, (w, h) in enumerate(sizes):
            textures[k] = (w, h, off)
            t = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
            if k == 1: t[..., 2] = np.maximum(t[..., 2], 140)          # a normal map that mostly points outwards
            chunks.append(t.reshape(-1)); off += w * h * 4
        texels = np.concatenate(chunks)
        for m in range(1, n_mats):
            for slot in range(6):                                     # diffuse, normal, emissive, roughness, metalness, opacity
                if rng.uniform() < 0.35: mats[m]["tex"][slot] = int(rng.integers(0, len(sizes)))
    material = rng.integers(0, n_mats, n).astype(np.int32)
    material[0] = 0
    lights = np.nonzero((mats["emissive"][material] * mats["emissive_intensity"][material, None]).sum(axis=1) > 0)[0].astype(np.int32)
    arrays = dict(verts=verts.reshape(n, 9), normals=vn.reshape(n, 9), uvs=rng.uniform(-1.5, 2.5, (n, 6)).astype(np.float32),
                  tbn=np.concatenate([fn, tg, bt], axis=1).astype(np.float32), smoothing=(rng.uniform(0, 1, n) < 0.4).astype(np.uint8),
                  material=material, materials=mats, textures=textures, texels=texels, lights=lights)
    cam = dict(pos=np.array([0.1, -0.2, 3.2], np.float32), dir=_unit(np.array([-0.03, 0.05, -1.0], np.float32)), up=np.array([0.0, 1.0, 0.0], np.float32),
               focal=0.05, fovy=float(rng.uniform(35.0, 70.0)), focal_dist=3.0, aperture=float(rng.choice([0.0, 0.08])))
    return arrays, cam


@pytest.mark.parametrize("seed,n_tris,tex", [(11, 9, False), (12, 16, True), (13, 40, False), (14, 300, True), (15, 300, False),
                                             (16, 6000, True), (17, 6000, False), (18, 1500, True)])
def test_random_scene_matches_oracle(ctx, oracle_mod, seed, n_tris, tex):
    arrays, cam = random_scene(seed, n_tris, tex)
    W, H, D, spp = 56, 40, 7, 6
    o = oracle_mod.Oracle(arrays)
    ocam = oracle_mod.make_camera(cam["pos"], cam["dir"], cam["up"], cam["focal"], cam["fovy"], cam["focal_dist"], cam["aperture"])
    ref, ref8 = o.render(ocam, W, H, D, 0, spp, seed)
    o.close()
    for device_build in (0, 1):
        ctx.set_option("device_build", device_build)
        ctx.upload_scene(arrays); ctx.set_camera(**cam); ctx.set_frame(W, H, D); ctx.set_tile(0, 1); ctx.reset()
        ctx.render(0, spp, seed)
        got, got8 = ctx.read_accum(), ctx.resolve_rgb8()
        assert np.isfinite(ref).all()
        assert (ref != 0).any(axis=2).mean() > 0.5, "the scene is mostly black: a poor test"
        assert np.array_equal(ref, got), (seed, device_build, float(np.abs(ref - got).max()))
        assert np.array_equal(ref8, got8)
    ctx.set_option("device_build", 1)
    st = ctx.collect_stats(0, spp, seed)
    print(f"seed {seed}: {n_tris} triangles, lights {len(arrays['lights'])}, per sample: rays {st['rays'] / st['samples']:.2f}, "
          f"shaded {st['hits_shaded'] / st['samples']:.2f}, texture fetches {st['tex_fetches'] / st['samples']:.2f}")


@pytest.mark.parametrize("seed,n_tris,tex,W,H,D", [(1162, 16, False, 56, 38, 7), (1231, 12, True, 56, 35, 4), (1953, 17, True, 48, 41, 6),
                                                  (2964, 64, False, 48, 44, 3), (3841, 12, True, 56, 35, 4)])
def test_grazing_light_hits_decide_shadow_rays_like_the_reference(ctx, oracle_mod, seed, n_tris, tex, W, H, D):
    """Scenes of tools/soak_random_scenes.py in which round 1's shadow walk went wrong for one pixel-sample each: a shadow ray
    that grazes its own light triangle, where Moeller-Trumbore reports the hit far nearer than the light SAMPLE, so that
    "any hit nearer than 0.9999 x the distance to the sample" is not "something else is closest" (pathtracer.cpp:522-526).
    The walk now meets the light triangle first and ends on whatever the closest-hit rule then accepts: FLAT pass, host-built
    and device-built tree all give the oracle's image."""
    arrays, cam = random_scene(seed, n_tris, tex)
    o = oracle_mod.Oracle(arrays)
    ocam = oracle_mod.make_camera(cam["pos"], cam["dir"], cam["up"], cam["focal"], cam["fovy"], cam["focal_dist"], cam["aperture"])
    ref, ref8 = o.render(ocam, W, H, D, 0, 4, seed)
    o.close()
    for flat in ((1, 0) if n_tris <= 16 else (0,)):
        ctx.set_option("flat", flat)
        ctx.upload_scene(arrays); ctx.set_camera(**cam); ctx.set_frame(W, H, D); ctx.set_tile(0, 1); ctx.reset()
        ctx.render(0, 4, seed)
        assert np.array_equal(ref, ctx.read_accum()), (seed, flat)
        assert np.array_equal(ref8, ctx.resolve_rgb8())
    ctx.set_option("flat", 1)
