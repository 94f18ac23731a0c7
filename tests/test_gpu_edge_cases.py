"""Edge cases the reference's semantics imply (it has no tests of its own): empty / tiny / degenerate
scenes, no lights, extreme trace depths, ragged and 1-pixel frames, missing textures — HIP path vs oracle,
bit for bit, through the C-ABI."""
import numpy as np
import pytest

from conftest import load_golden, scene_from_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from pbrpathtracer_amd import ptk
    c = ptk.Context(0)
    yield c
    c.close()


def _cam(z, aperture=None):
    cam, proj = z["cam"], z["proj"]
    return dict(pos=cam[0:3], dir=cam[3:6], up=cam[6:9], focal=float(proj[0]), fovy=float(proj[1]),
                focal_dist=float(z["focal_dist"]), aperture=float(z["aperture"]) if aperture is None else aperture)


def _both(ctx, OB, arrays, cam, W, H, D, spp, seed=5):
    o = OB.Oracle(arrays)
    ocam = OB.make_camera(cam["pos"], cam["dir"], cam["up"], cam["focal"], cam["fovy"], cam["focal_dist"], cam["aperture"])
    ref, ref8 = o.render(ocam, W, H, D, 0, spp, seed)
    ctx.upload_scene(arrays); ctx.set_camera(**cam); ctx.set_frame(W, H, D); ctx.set_tile(0, 1); ctx.reset()
    ctx.render(0, spp, seed)
    return ref, ref8, ctx.read_accum(), ctx.resolve_rgb8()


def _empty_like(arrays, n=0):
    a = {k: v[:n].copy() if k in ("verts", "normals", "uvs", "tbn", "smoothing", "material") else v.copy() for k, v in arrays.items()}
    a["lights"] = np.zeros(0, np.int32)
    return a


def test_empty_scene_renders_black(ctx, oracle_mod):
    z = load_golden("tier_s_cornell.npz")
    arrays = _empty_like(scene_from_golden(z))
    ref, ref8, got, got8 = _both(ctx, oracle_mod, arrays, _cam(z), 33, 17, 4, 3)
    assert not got.any() and not got8.any() and np.array_equal(ref, got)
    assert ctx.bvh_info()[0] == 0


@pytest.mark.parametrize("n", [1, 2, 3, 5])
def test_few_triangles(ctx, oracle_mod, n):
    z = load_golden("tier_s_cornell.npz")
    full = scene_from_golden(z)
    keep = [int(full["lights"][0])] + [i for i in range(len(full["verts"])) if i != int(full["lights"][0])][: n - 1]
    a = {k: (v[keep].copy() if k in ("verts", "normals", "uvs", "tbn", "smoothing", "material") else v.copy()) for k, v in full.items()}
    a["lights"] = np.array([0], np.int32)
    ref, ref8, got, got8 = _both(ctx, oracle_mod, a, _cam(z), 40, 40, 4, 4)
    assert np.array_equal(ref, got) and np.array_equal(ref8, got8)


def test_no_lights_and_degenerate_triangles(ctx, oracle_mod):
    z = load_golden("tier_s_glass.npz")
    a = scene_from_golden(z)
    a["lights"] = np.zeros(0, np.int32)                       # DirectIllumimation returns 0 (pathtracer.cpp:506-507)
    a["materials"] = a["materials"].copy(); a["materials"]["emissive"][:] = 0
    a["verts"] = a["verts"].copy()
    a["verts"][3, 3:6] = a["verts"][3, 0:3]                   # zero-area triangle: |a| < EPS cull (pathtracer.cpp:387)
    a["verts"][5, 6:9] = a["verts"][5, 3:6]
    ref, ref8, got, got8 = _both(ctx, oracle_mod, a, _cam(z), 48, 32, 6, 3)
    assert np.array_equal(ref, got) and not got.any()          # nothing emits -> black, but every branch ran


@pytest.mark.parametrize("depth", [0, 1, 2, 12])
def test_trace_depth_extremes(ctx, oracle_mod, depth):
    z = load_golden("tier_s_glass.npz")
    ref, ref8, got, got8 = _both(ctx, oracle_mod, scene_from_golden(z), _cam(z), 40, 28, depth, 3)
    assert np.array_equal(ref, got) and np.array_equal(ref8, got8)
    if depth == 0:
        assert not got.any()                                   # `iter < mMaxDepth` never holds (pathtracer.cpp:571)


@pytest.mark.parametrize("wh", [(1, 1), (17, 1), (1, 19), (16, 16), (15, 33)])
def test_ragged_frames(ctx, oracle_mod, wh):
    z = load_golden("tier_s_opacity.npz")
    W, H = wh
    ref, ref8, got, got8 = _both(ctx, oracle_mod, scene_from_golden(z), _cam(z), W, H, 4, 5)
    assert got.shape == (H, W, 3) and np.array_equal(ref, got) and np.array_equal(ref8, got8)


def test_missing_texture_samples_as_zero(ctx, oracle_mod):
    """An Image whose file failed to load has no data and samples as 0 (image.cpp:65-66): staged as a
    zero-extent texture."""
    z = load_golden("tier_s_opacity.npz")
    a = scene_from_golden(z)
    a["textures"] = a["textures"].copy()
    a["textures"]["width"][0] = 0; a["textures"]["height"][0] = 0
    ref, ref8, got, got8 = _both(ctx, oracle_mod, a, _cam(z), 40, 30, 4, 4)
    assert np.array_equal(ref, got)
    # ... and a scene ALL of whose textures are missing files (a .pts moved to another machine): texture entries without a single
    # texel behind them still upload and render (found by tools/soak_api.py: the upload refused the empty atlas)
    a["textures"]["width"][:] = 0; a["textures"]["height"][:] = 0; a["textures"]["offset"][:] = 0
    a["texels"] = np.zeros(0, np.uint8)
    ref, ref8, got, got8 = _both(ctx, oracle_mod, a, _cam(z), 40, 30, 4, 4)
    assert np.array_equal(ref, got) and ref.any()


def test_many_spp_chunks_and_passes(ctx, oracle_mod):
    """spp not a multiple of the chunk, several passes through a tiny sample-buffer budget: identical."""
    z = load_golden("tier_s_cornell.npz")
    arrays = scene_from_golden(z); cam = _cam(z)
    ref, ref8, got, got8 = _both(ctx, oracle_mod, arrays, cam, 48, 32, 4, 37)
    assert np.array_equal(ref, got)
    ctx.set_option("pass_bytes", 1 << 20); ctx.set_option("chunk", 5)
    ctx.reset(); ctx.render(0, 37, 5)
    again = ctx.read_accum()
    ctx.set_option("pass_bytes", float(16 << 30)); ctx.set_option("chunk", 0)
    assert np.array_equal(again, got)


def test_bad_arguments_are_errors(ctx):
    from pbrpathtracer_amd import ptk
    z = load_golden("tier_s_cornell.npz")
    a = scene_from_golden(z)
    bad = dict(a); bad["material"] = a["material"].copy(); bad["material"][0] = 99
    with pytest.raises(ptk.PtkError):
        ctx.upload_scene(bad)
    bad = dict(a); bad["lights"] = np.array([12345], np.int32)
    with pytest.raises(ptk.PtkError):
        ctx.upload_scene(bad)
    with pytest.raises(ptk.PtkError):
        ctx.set_frame(0, 10, 3)
    with pytest.raises(ptk.PtkError):
        ctx.set_tile(2, 2)
    with pytest.raises(ptk.PtkError):
        ctx.set_option("no_such_option", 1)
    # coordinates the kernels' exact short reciprocal does not cover are refused at both doors: vertices ...
    bad = dict(a); bad["verts"] = a["verts"].copy(); bad["verts"][0, 0] = np.float32(3e18)
    with pytest.raises(ptk.PtkError, match="2\\^61"):
        ctx.upload_scene(bad)
    bad["verts"][0, 0] = np.float32(np.nan)
    with pytest.raises(ptk.PtkError):
        ctx.upload_scene(bad)
    # more triangles than the walk's 32-bit record offsets address: refused before any array is read (so the arrays here may be small)
    d = ptk.scene_desc(ptk.normalise_arrays(a)); d.num_triangles = 89_478_486
    assert ctx.L.ptk_upload_scene(ctx.h, ptk.C.byref(d)) == -4 and b"32-bit record offsets" in ctx.L.ptk_last_error(ctx.h)
    # ... and the camera position (ray origins): DESIGN.md, documented difference 7
    cam = _cam(z)
    for v in (np.float32(3e18), np.float32(np.inf), np.float32(np.nan)):
        far = dict(cam); far["pos"] = np.array([0.0, v, 0.0], np.float32)
        with pytest.raises(ptk.PtkError, match="camera position"):
            ctx.set_camera(**far)
    ctx.upload_scene(a)                                        # the context stays usable
    ctx.set_camera(**cam)


def test_flat_and_bvh_walks_agree(ctx, oracle_mod):
    """Scenes of <= 16 triangles take the FLAT kernel (every triangle, scalar loads, shadow + bounce ray in
    one pass); the BVH walk must give the same accumulator bit for bit."""
    z = load_golden("tier_s_cornell.npz")
    arrays = scene_from_golden(z); cam = _cam(z)
    ref, ref8, flat, flat8 = _both(ctx, oracle_mod, arrays, cam, 64, 48, 5, 9)
    ctx.set_option("flat", 0)
    ctx.reset(); ctx.render(0, 9, 5)
    walk = ctx.read_accum()
    ctx.set_option("flat", 1)
    assert np.array_equal(flat, ref) and np.array_equal(walk, ref)


@pytest.mark.parametrize("golden,aperture", [("tier_s_glass.npz", 0.0), ("tier_s_opacity.npz", None), ("tier_s_cornell.npz", 0.0)])
def test_work_distribution_modes_agree(ctx, oracle_mod, golden, aperture):
    """How work reaches the lanes never changes a bit: persistent waves pulling items from the queues, one item
    per wave, batched queue pops, several generations of waves, other chunk sizes, tile shares - all equal the
    oracle's accumulator (BVH and FLAT kernels, with and without the primary-hit cache / live-quadrant list)."""
    z = load_golden(golden)
    arrays = scene_from_golden(z); cam = _cam(z, aperture)
    W, H, D, spp = 150, 70, 5, 24
    ref, ref8, got, got8 = _both(ctx, oracle_mod, arrays, cam, W, H, D, spp)
    assert np.array_equal(ref, got) and np.array_equal(ref8, got8)
    defaults = {"persistent": -1, "generations": 0, "max_batch": 1, "chunk": 0, "tri_threshold": 4}
    for opts in ({"persistent": 1}, {"persistent": 0}, {"persistent": 1, "max_batch": 7}, {"persistent": 1, "generations": 3, "chunk": 2},
                 {"persistent": 1, "chunk": 24}, {"persistent": 0, "chunk": 3}, {"persistent": 1, "tri_threshold": 0}, {"persistent": 1, "tri_threshold": 64}):
        for k, v in {**defaults, **opts}.items():
            ctx.set_option(k, v)
        ctx.reset(); ctx.render(0, spp, 5)
        assert np.array_equal(ctx.read_accum(), got), opts
        assert np.array_equal(ctx.resolve_rgb8(), got8), opts
    # tile shares under persistent waves: three ranks' images are disjoint and sum to the whole
    ctx.set_option("persistent", 1); ctx.set_option("generations", 2); ctx.set_option("chunk", 0); ctx.set_option("tri_threshold", 4)
    total = np.zeros_like(got)
    for r in range(3):
        ctx.set_tile(r, 3); ctx.reset(); ctx.render(0, spp, 5)
        part = ctx.read_accum()
        assert not np.any((part != 0) & (total != 0))
        total += part
    assert np.array_equal(total, got)
    ctx.set_tile(0, 1)
    for k, v in defaults.items():
        ctx.set_option(k, v)


def test_bound_handoff_buffer_follows_another_accumulator_and_other_tiles(ctx, oracle_mod):
    """with a hand-off buffer bound, pixels that are black for every sample are skipped by the accumulate kernel unless the
    next frame is marked 'write everything'.  Switching to ANOTHER accumulator (ptk_bind_accum) or to other tiles (ptk_set_tile)
    must set that mark: the skipped pixels hold light in the old accumulator's frame that the new one does not have."""
    import torch
    from pbrpathtracer_amd import ptk
    z = load_golden("tier_s_cornell.npz")
    arrays = scene_from_golden(z); cam = _cam(z, aperture=0.0)
    W, H = 96, 64
    ctx.upload_scene(arrays); ctx.set_camera(**cam); ctx.set_frame(W, H, 3); ctx.set_tile(0, 1); ctx.reset()
    raw = ptk.load().ptk_host_alloc(W * H * 3)
    import ctypes as C
    out = np.ctypeslib.as_array(C.cast(raw, C.POINTER(C.c_uint8)), shape=(H, W, 3))
    try:
        ctx.bind_out_image(out)
        ctx.render(0, 2, 9); ctx.resolve_rgb8(out)
        # a camera move without a reset: pixels that now miss the box still hold light in the accumulator (they keep dimming)
        cam2 = dict(cam); cam2["pos"] = np.array(cam["pos"], np.float32) + np.array([0.8, 0.3, 0.0], np.float32)
        ctx.set_camera(**cam2); ctx.render(2, 2, 9); ctx.resolve_rgb8(out)
        dev = np.zeros((H, W, 3), np.uint8); ctx.L.ptk_resolve_rgb8(ctx.h, dev.ctypes.data)
        assert np.array_equal(out, dev)
        # another, zeroed accumulator: the frame is now only the NEW accumulator's - every pixel of the bound buffer must follow
        other = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda:0"); torch.cuda.synchronize()
        ctx.bind_accum(other.data_ptr())
        ctx.render(4, 1, 9); ctx.resolve_rgb8(out)
        ctx.L.ptk_resolve_rgb8(ctx.h, dev.ctypes.data)
        assert np.array_equal(out, dev)
        ctx.bind_accum(None)
        # other tiles: rank 1 of 2 after rank 0 of 2 (the accumulator keeps what rank 0's tiles hold; the frame shows both)
        ctx.reset(); ctx.set_tile(0, 2); ctx.render(0, 1, 9); ctx.resolve_rgb8(out)
        ctx.set_tile(1, 2); ctx.render(0, 1, 9); ctx.resolve_rgb8(out)
        ctx.L.ptk_resolve_rgb8(ctx.h, dev.ctypes.data)
        assert np.array_equal(out, dev) and out.any()
    finally:
        ctx.set_tile(0, 1); ctx.bind_accum(None); ctx.bind_out_image(None)
        ptk.load().ptk_host_free(raw)


def test_render_takes_more_passes_when_device_memory_is_short(oracle_mod):
    """the sample-buffer budget is 16 GiB per pass; on a GPU whose memory is mostly taken (a framework sharing it, a huge
    scene) a render must fall back to smaller passes instead of failing with an allocation error.  Most of the free memory is
    taken away with one torch allocation; a 1920 x 1080 render of 192 spp then needs more passes - and gives the image of the same
    samples rendered in comfortable 32-spp calls, bit for bit."""
    import torch
    from pbrpathtracer_amd import ptk
    z = load_golden("tier_s_cornell.npz")
    arrays = scene_from_golden(z); cam = _cam(z, aperture=0.0)
    c = ptk.Context(0)
    hog = None
    try:
        W, H = 1920, 1080
        c.upload_scene(arrays); c.set_camera(**cam); c.set_frame(W, H, 3); c.set_tile(0, 1); c.reset()
        for k in range(6):
            c.render(32 * k, 32, 3)
        want = c.read_accum()
        free, total = torch.cuda.mem_get_info(0)
        keep = 5 << 30                                   # leave ~5 GiB: 192 spp x 2 M pixels x 16 B = 6.4 GB per buffer, two of them, will not fit
        hog = torch.empty(max(0, free - keep), dtype=torch.uint8, device="cuda:0"); torch.cuda.synchronize()
        c.reset(); c.render(0, 192, 3)
        got = c.read_accum()
        ms, launches = c.last_render_ms()
        assert launches > 3                               # more than one pass (3 launches each)
        assert np.array_equal(got, want)
    finally:
        del hog
        torch.cuda.empty_cache()
        c.close()


def test_kernel_log_times_every_trace_launch(ctx):
    """ptk_kernel_log / ptk_kernel_log_read (bench.py's per-launch durations inside its timed region): one entry per trace launch,
    in order, overlapped renders included; reading empties the log; capacity bounds it."""
    z = load_golden("tier_s_cornell.npz")
    arrays = scene_from_golden(z); cam = _cam(z, aperture=0.0)
    ctx.upload_scene(arrays); ctx.set_camera(**cam); ctx.set_frame(128, 96, 3); ctx.set_tile(0, 1); ctx.reset()
    ctx.kernel_log(8)
    for k in range(5):
        ctx.render(4 * k, 4, 1)
    ms = ctx.kernel_log_read()
    assert len(ms) == 5 and all(0.0 < m < 1000.0 for m in ms)
    assert ctx.kernel_log_read() == []
    for k in range(11):
        ctx.render(20 + k, 1, 1)
    assert len(ctx.kernel_log_read()) == 8               # the log holds `capacity` launches
    ctx.kernel_log(0)
    ctx.render(40, 1, 1)
    assert ctx.kernel_log_read() == []
