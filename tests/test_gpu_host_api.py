"""GPU tests through the drop-in PathTracer API (C++ host layer -> C-ABI -> HIP kernels) on scenes
loaded from the reference's own formats (.obj + .pts), against the oracle fed with the same staged arrays."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_for(OB, pt, scene):
    from pbrpathtracer_amd.pathtracer import camera_from_scene
    o = OB.Oracle(pt.StagedScene())
    cam = camera_from_scene(scene)
    ocam = OB.make_camera(cam["pos"], cam["dir"], cam["up"], cam["focal"], cam["fovy"], cam["focal_dist"], cam["aperture"])
    return o, ocam


@pytest.mark.parametrize("cfg,kw,spp", [
    ("C1", dict(width=128, height=96), 8),
    ("C3", dict(width=160, height=90, nu=16, nv=8, tex_size=64), 6),
    ("C4", dict(width=96, height=54, grid=24), 4),
    ("C5", dict(width=96, height=54, nx=40, nz=20), 4),
])
def test_scene_file_render_matches_oracle(tmp_path, oracle_mod, cfg, kw, spp):
    from pbrpathtracer_amd import scenes as S
    from pbrpathtracer_amd.pathtracer import PathTracer
    pts, scene, _ = S.build_config(cfg, str(tmp_path), **kw)
    pt = PathTracer(0)
    pt.LoadSceneFile(pts)
    W, H = pt.GetResolution(); D = pt.GetTraceDepth()
    out = np.zeros((H, W, 3), np.uint8)
    pt.SetOutImage(out); pt.SetSeed(11)
    pt.RenderFrames(spp)
    assert pt.LastError() == "" and pt.GetSamples() == spp
    total = pt.ReadAccumulation()
    o, ocam = _oracle_for(oracle_mod, pt, scene)
    ref, ref8 = o.render(ocam, W, H, D, 0, spp, 11)
    rmse = float(np.sqrt(np.mean((total / spp - ref / spp) ** 2)))
    exact = float(np.mean(total == ref))
    print(f"{cfg}: rmse {rmse:.2e} exact {exact:.4f}")
    assert rmse <= 1e-3
    assert np.array_equal(total, ref)
    assert np.array_equal(out, ref8)
    pt.close()


def test_render_frame_semantics(tmp_path, oracle_mod):
    """RenderFrame() adds exactly one sample per pixel and refreshes the caller's RGB8 buffer;
    ResetImage() restarts; Exit() before a frame skips it (pathtracer.cpp:741-822)."""
    from pbrpathtracer_amd import scenes as S
    from pbrpathtracer_amd.pathtracer import PathTracer
    pts, scene, _ = S.build_config("C1", str(tmp_path), width=64, height=64)
    pt = PathTracer(0)
    pt.LoadSceneFile(pts)
    out = np.zeros((64, 64, 3), np.uint8)
    pt.SetOutImage(out)
    pt.RenderFrame()
    assert pt.GetSamples() == 1 and out.any()
    first = out.copy()
    pt.RenderFrame(); pt.RenderFrame()
    assert pt.GetSamples() == 3
    a3 = pt.ReadAccumulation()
    exp8 = (np.clip(a3 / np.float32(3), 0, 1) * np.float32(255)).astype(np.uint8)
    assert np.array_equal(out, exp8)                        # clamp(total/samples)*255 truncated, bottom-up
    pt.ResetImage(); pt.RenderFrame()
    assert pt.GetSamples() == 1 and np.array_equal(out, first)
    # rows are bottom-up: the ceiling light (top of the box) is in the upper rows of the buffer
    rows = out.astype(np.int32).sum(axis=(1, 2))
    assert rows[40:].sum() > 0 and a3[0].sum() == 0
    pt.Exit(); pt.RenderFrame()
    assert pt.GetSamples() == 2                              # mSamples still advances (pathtracer.cpp:753)
    assert np.array_equal(pt.ReadAccumulation(), pt.ReadAccumulation())
    pt.close()


def test_full_size_properties(tmp_path):
    """BASELINE config C2 at full resolution: size-independent properties instead of an oracle image —
    (a) splitting the spp budget over launches is invisible, (b) two ranks' tiles are disjoint and sum
    to the single-GPU accumulator bit for bit, (c) the accumulator is linear in the sample count."""
    from pbrpathtracer_amd import scenes as S
    from pbrpathtracer_amd.pathtracer import PathTracer
    pts, scene, _ = S.build_config("C2", str(tmp_path))
    pt = PathTracer(0); pt.LoadSceneFile(pts); pt.SetSeed(5)
    pt.RenderFrames(16)
    full = pt.ReadAccumulation()
    assert np.isfinite(full).all() and full.shape == (720, 1280, 3)
    pt.ResetImage(); pt.RenderFrames(5); pt.RenderFrames(11)
    assert np.array_equal(pt.ReadAccumulation(), full)
    parts = []
    for r in range(2):
        pt.SetTile(r, 2); pt.ResetImage(); pt.RenderFrames(16)
        parts.append(pt.ReadAccumulation())
    assert not np.logical_and(parts[0] != 0, parts[1] != 0).any()
    assert np.array_equal(parts[0] + parts[1], full)
    m = full.reshape(-1, 3).mean(0) / 16
    assert 0.03 < m.min() and m.max() < 0.2                  # the box covers ~22 % of the 16:9 frame
    pt.close()


@pytest.mark.parametrize("cfg,kw,spp", [("C1", dict(width=96, height=80), 8), ("C4", dict(width=96, height=54, grid=20), 5)])
def test_exact_pinhole_primary_cache(tmp_path, oracle_mod, cfg, kw, spp):
    """SetCameraAperture(0): the camera ray of a pixel is the same for every sample, so the kernel reuses
    its closest hit (primary-visibility cache) and skips pixels whose camera ray misses.  Must equal the
    oracle (which traces every camera ray) and the cache-disabled kernel bit for bit."""
    from pbrpathtracer_amd import scenes as S
    from pbrpathtracer_amd.pathtracer import PathTracer, camera_from_scene
    pts, scene, _ = S.build_config(cfg, str(tmp_path), **kw)
    pt = PathTracer(0)
    pt.LoadSceneFile(pts)
    pt.SetCameraAperture(0.0)
    pt.SetSeed(21)
    W, H = pt.GetResolution(); D = pt.GetTraceDepth()
    pt.RenderFrames(spp)
    cached = pt.ReadAccumulation()
    st = pt.context().collect_stats(0, spp, 21)
    assert st["rays"] < st["samples"] + st["shadow_rays"] + st["hits_shaded"]     # camera rays were not traversed
    ctx = pt.context(); ctx.set_option("primary_cache", 0)
    pt.ResetImage(); pt.RenderFrames(spp)
    plain = pt.ReadAccumulation()
    assert np.array_equal(cached, plain)
    o = oracle_mod.Oracle(pt.StagedScene())
    cam = camera_from_scene(scene); cam["aperture"] = 0.0
    ocam = oracle_mod.make_camera(cam["pos"], cam["dir"], cam["up"], cam["focal"], cam["fovy"], cam["focal_dist"], cam["aperture"])
    ref, _ = o.render(ocam, W, H, D, 0, spp, 21)
    assert np.array_equal(cached, ref)
    pt.close()


@pytest.mark.parametrize("cfg,kw,spp,aperture", [("C3", dict(width=160, height=90), 6, None), ("C3", dict(width=128, height=72), 4, 0.08),
                                                  ("C1", dict(width=120, height=48), 8, 0.02), ("C4", dict(width=96, height=54, grid=20), 5, 0.15)])
def test_lens_cull_is_exact(tmp_path, oracle_mod, cfg, kw, spp, aperture):
    """Thin-lens cameras (pathtracer.cpp:785-791): a pixel NONE of whose lens rays can reach the scene's bounding box is black for
    every sample, and live_mask_kernel leaves it out of the frame's work (round 4).  The cull is conservative,
    so the image must equal the oracle's - which shoots every ray - and the cull-disabled kernel's bit for bit, wide apertures
    (strongly defocused bundles) and camera moves included; and it must actually cull on the 16:9 framings."""
    from pbrpathtracer_amd import scenes as S
    from pbrpathtracer_amd.pathtracer import PathTracer, camera_from_scene
    pts, scene, _ = S.build_config(cfg, str(tmp_path), **kw)
    pt = PathTracer(0)
    pt.LoadSceneFile(pts)
    cam = camera_from_scene(scene)
    if aperture is not None:
        pt.SetCameraAperture(aperture); cam["aperture"] = aperture
    assert cam["aperture"] != 0.0
    pt.SetSeed(33)
    W, H = pt.GetResolution(); D = pt.GetTraceDepth()
    o = oracle_mod.Oracle(pt.StagedScene())
    for move in (None, ((0.9, 0.3, -3.2), (-0.25, -0.1, 1.0)), ((0.0, 0.0, -9.0), (0.45, 0.0, 1.0))):
        if move is not None:
            pt.SetCamera(move[0], move[1], (0, 1, 0)); pt.ResetImage()
            cam["pos"] = np.array(move[0], np.float32)
            cam["dir"] = np.array(move[1], np.float32); cam["up"] = np.array([0, 1, 0], np.float32)      # raw, as SetCamera gets them
        ctx = pt.context(); ctx.set_option("lens_cull", 1)
        pt.RenderFrames(spp)
        assert pt.LastError() == ""
        culled = pt.ReadAccumulation()
        st = ctx.collect_stats(0, spp, 33)
        ctx.set_option("lens_cull", 0)
        pt.ResetImage(); pt.RenderFrames(spp)
        plain = pt.ReadAccumulation()
        st0 = ctx.collect_stats(0, spp, 33)
        assert st0["paths_started"] == st0["samples"]                      # every pixel traced without the cull
        assert np.array_equal(culled, plain), move
        ocam = oracle_mod.make_camera(cam["pos"], cam["dir"], cam["up"], cam["focal"], cam["fovy"], cam["focal_dist"], cam["aperture"], normalise=move is not None)
        ref, _ = o.render(ocam, W, H, D, 0, spp, 33)
        assert np.array_equal(culled, ref), move
        if move is None and cfg in ("C3", "C1"):
            assert st["paths_started"] < 0.7 * st["samples"], (st["paths_started"], st["samples"])     # the box fills a fraction of a wide frame
        pt.ResetImage()
    pt.close()


def test_exit_from_another_thread(tmp_path):
    """Exit() / GetSamples() are called from the UI thread while the render thread is inside RenderFrame
    (main.cpp:1153, :2277-2324): must be safe.  Exit() cuts the render in flight - whole passes are skipped, an aborted
    pass adds NOTHING (the accumulator is a prefix of the passes, bit for bit), mSamples still advances - and the next
    RenderFrame() renders again without a ResetImage(), as the reference resets mExit on entry (pathtracer.cpp:742)."""
    import threading
    import time
    from pbrpathtracer_amd import scenes as S
    from pbrpathtracer_amd.pathtracer import PathTracer
    pts, scene, _ = S.build_config("C4", str(tmp_path), width=640, height=360, grid=60)
    pt = PathTracer(0); pt.LoadSceneFile(pts); pt.SetSeed(5)
    pt.RenderFrames(1)
    ctx = pt.context()
    # small passes so that the 2048-sample render is many launches: 640x360 -> 920 tiles x 4 x 64 x 16 B = 3.8 MB per sample
    ctx.set_option("pass_bytes", float(32 << 20))            # 8 samples per pass
    pt.ResetImage()
    seen = []
    stop = threading.Event()

    def ui():
        while not stop.is_set():
            seen.append(pt.GetSamples())
            time.sleep(0.0005)
    t = threading.Thread(target=ui); t.start()
    killer = threading.Timer(0.02, pt.Exit); killer.start()
    t0 = time.time()
    pt.RenderFrames(2048)                                   # long enough for Exit() to land mid-render
    dt = time.time() - t0
    stop.set(); t.join(); killer.join()
    assert pt.GetSamples() == 2048 and pt.LastError() == ""
    aborted = pt.ReadAccumulation()
    assert np.isfinite(aborted).all()
    # which prefix of the 8-sample passes made it?  render them again on a second tracer and compare bit for bit
    ref = PathTracer(0); ref.LoadSceneFile(pts); ref.SetSeed(5); ref.RenderFrames(1); ref.ResetImage()
    ref.context().set_option("pass_bytes", float(32 << 20))
    done, match = 0, np.array_equal(aborted, np.zeros_like(aborted))
    while not match and done < 2048:
        ref.RenderFrames(8); done += 8
        match = np.array_equal(ref.ReadAccumulation(), aborted)
    assert match, "the aborted render is not a whole number of passes"
    print(f"Exit() after 20 ms: {done} of 2048 samples were accumulated, render call took {dt*1e3:.1f} ms, UI polled {len(seen)} times")
    # ... and the next frame renders again, WITHOUT ResetImage: samples 2048.. are added on top
    pt.RenderFrames(8)
    assert pt.GetSamples() == 2056
    resumed = pt.ReadAccumulation()
    assert not np.array_equal(resumed, aborted) and np.isfinite(resumed).all()
    ref2 = PathTracer(0); ref2.LoadSceneFile(pts); ref2.SetSeed(5); ref2.RenderFrames(1); ref2.ResetImage()
    ref2.context().reset(); ref2.context().render(2048, 8, 5)
    assert np.array_equal((resumed - aborted)[aborted == 0], ref2.ReadAccumulation()[aborted == 0])   # exact where nothing was added before
    pt.close(); ref.close(); ref2.close()


def test_exit_cuts_every_render_in_flight(tmp_path):
    """ptk_render is asynchronous while no output image is bound, so several renders can be queued when Exit() arrives: it
    cuts ALL of them (kernels stand down when the named generation >= their own), not only the newest - an older render still
    executing would otherwise run to its end behind the UI's back.  Two renders of 1024 samples in 8-sample passes are queued
    and Exit() follows at once: what was accumulated is a whole number of passes (bit for bit a prefix) and FEWER than the
    first render's alone."""
    from pbrpathtracer_amd import scenes as S
    from pbrpathtracer_amd.pathtracer import PathTracer
    pts, scene, _ = S.build_config("C4", str(tmp_path), width=640, height=360, grid=60)
    pt = PathTracer(0); pt.LoadSceneFile(pts); pt.SetSeed(5)
    pt.RenderFrames(1)
    ctx = pt.context()
    ctx.set_option("pass_bytes", float(32 << 20))            # 8 samples per pass
    ctx.reset()
    ctx.render(0, 1024, 5); ctx.render(1024, 1024, 5)          # both only queued: nothing waits for them
    ctx.request_exit()
    ctx.synchronize()
    aborted = ctx.read_accum()
    assert np.isfinite(aborted).all()
    ref = PathTracer(0); ref.LoadSceneFile(pts); ref.SetSeed(5); ref.RenderFrames(1)
    rctx = ref.context(); rctx.set_option("pass_bytes", float(32 << 20)); rctx.reset()
    done, match = 0, not aborted.any()
    while not match and done < 1024:
        rctx.render(done, 8, 5); done += 8
        match = np.array_equal(rctx.read_accum(), aborted)
    assert match, "not a whole number of passes of the first render: an older render ran on after Exit()"
    assert done < 1024
    print(f"Exit() right after queueing 2 x 1024 samples: {done} samples were accumulated")
    # renders issued after the Exit() are not affected
    ctx.reset(); ctx.render(0, 8, 5)
    rctx.reset(); rctx.render(0, 8, 5)
    assert np.array_equal(ctx.read_accum(), rctx.read_accum())
    pt.close(); ref.close()


def test_material_edits_after_build_take_effect(tmp_path, oracle_mod):
    """SetMaterial after BuildBVH: the reference's triangles point into the loaded materials, so the next RenderFrame()
    uses the edited values without a rebuild (pathtracer.cpp:243-258) while the light list stays BuildBVH's.  Here the
    material table is rewritten in place (ptk_update_materials); geometry edits after BuildBVH are reported."""
    from pbrpathtracer_amd import scenes as S
    from pbrpathtracer_amd.pathtracer import PathTracer
    pts, scene, _ = S.build_config("C1", str(tmp_path), width=96, height=64, depth=4)
    pt = PathTracer(0); pt.LoadSceneFile(pts); pt.SetSeed(2)
    pt.RenderFrames(4)
    before = pt.ReadAccumulation()
    # paint the first element blue and make it half mirror: (type, diffuse, specular, emissive, I, roughness, reflectiveness, translucency, ior)
    pt.SetMaterial(0, 0, np.array([0, 0.1, 0.2, 0.9, 1, 1, 1, 0, 0, 0, 1, 0.0, 0.5, 1.0, 1.5], np.float32))
    pt.ResetImage(); pt.RenderFrames(4)
    assert pt.LastError() == ""
    edited = pt.ReadAccumulation()
    assert not np.array_equal(before, edited)
    # the same scene staged afresh with that material gives the same image: oracle on the re-staged arrays
    o, ocam = _oracle_for(oracle_mod, pt, scene)
    W, H = pt.GetResolution()
    ref, _ = o.render(ocam, W, H, pt.GetTraceDepth(), 0, 4, 2)
    assert np.array_equal(edited, ref)
    pt.close()


def test_page_locked_handoff_buffer(tmp_path):
    """Progressive display hand-off: RenderFrame() into a page-locked buffer from
    AllocOutImage() (ptk_host_alloc) delivers the same RGB8 frame as into ordinary memory."""
    from pbrpathtracer_amd import scenes as S
    from pbrpathtracer_amd.pathtracer import PathTracer
    pts, scene, _ = S.build_config("C1", str(tmp_path), width=96, height=64)
    frames = []
    for pinned in (False, True):
        pt = PathTracer(0)
        pt.LoadSceneFile(pts); pt.SetSeed(5)
        W, H = pt.GetResolution()
        out = pt.AllocOutImage() if pinned else np.zeros((H, W, 3), np.uint8)
        assert out.shape == (H, W, 3) and out.dtype == np.uint8
        pt.SetOutImage(out)
        for _ in range(3):
            pt.RenderFrame()
        assert pt.LastError() == "" and pt.GetSamples() == 3
        frames.append(np.array(out))
        pt.SetOutImage(None)
        del out, pt
    assert frames[0].any() and np.array_equal(frames[0], frames[1])


def test_bound_handoff_buffer_tracks_every_frame(tmp_path, oracle_mod):
    """The interactive loop (main.cpp:3563-3618: one RenderFrame(), one glTexSubImage2D(texData)): SetOutImage binds the
    caller's buffer: a ptk_host_alloc one is bound (ptk_bind_out_image) and the accumulate kernel's 8-bit resolve lands in it
    directly, an ordinary `new GLubyte[]`-like buffer is left alone and receives a copy of the frame.  Whatever happens between
    frames - camera moves that change which pixels are black for good, ResetImage, an Exit(), a new resolution - the
    buffer holds exactly the device's resolved frame (and the oracle's) after every RenderFrame()."""
    from pbrpathtracer_amd import scenes as S
    from pbrpathtracer_amd.pathtracer import PathTracer, camera_from_scene
    pts, scene, _ = S.build_config("C1", str(tmp_path), width=112, height=80, depth=4)
    for pinned in (False, True):
        pt = PathTracer(0); pt.LoadSceneFile(pts); pt.SetSeed(9); pt.SetCameraAperture(0.0)
        W, H = pt.GetResolution()
        out = pt.AllocOutImage() if pinned else np.full((H, W, 3), 77, np.uint8)       # (garbage in: every frame overwrites it whole)
        pt.SetOutImage(out)
        ctx = None
        for frame in range(12):
            if frame == 4:
                pt.SetCamera((0.6, 0.2, -3.5), (-0.15, -0.05, 1.0), (0, 1, 0))       # other pixels miss the box now
            if frame == 7:
                pt.ResetImage()
            if frame == 9:
                pt.Exit()                                                              # (nothing in flight: must not disturb the next frame)
            pt.RenderFrame()
            assert pt.LastError() == ""
            ctx = ctx or pt.context()
            dev = np.zeros((H, W, 3), np.uint8)
            ctx.L.ptk_resolve_rgb8(ctx.h, dev.ctypes.data)                               # into another buffer: a real copy of the device's frame
            assert np.array_equal(np.asarray(out), dev), (pinned, frame)
        assert np.asarray(out).any()
        # the last frames against the oracle: 5 samples since the reset at frame 7
        cam = camera_from_scene(scene); cam["aperture"] = 0.0
        cam["pos"] = np.array([0.6, 0.2, -3.5], np.float32)
        d = np.array([-0.15, -0.05, 1.0], np.float32); cam["dir"] = d / np.float32(np.sqrt((d * d).sum(dtype=np.float32)))
        o = oracle_mod.Oracle(pt.StagedScene())
        ocam = oracle_mod.make_camera(cam["pos"], cam["dir"], cam["up"], cam["focal"], cam["fovy"], cam["focal_dist"], cam["aperture"])
        _, ref8 = o.render(ocam, W, H, pt.GetTraceDepth(), 0, 5, 9)
        assert pt.GetSamples() == 5 and np.array_equal(np.asarray(out), ref8), pinned
        # a new resolution: the caller reallocates its buffer and hands it over again (main.cpp:3425-3446)
        pt.SetResolution((64, 48))
        out2 = np.full((48, 64, 3), 5, np.uint8)
        if not pinned:
            # ordinary memory is NEVER registered with the runtime (round 3 page-locked it in place; the viewer frees texData before
            # it hands over the next buffer and at exit without a word, main.cpp:3433-3445, :3622 - ADVICE r03): the frame is copied
            import ctypes
            hip = ctypes.CDLL("libamdhip64.so"); dev_alias = ctypes.c_void_p(0)
            assert hip.hipHostGetDevicePointer(ctypes.byref(dev_alias), ctypes.c_void_p(out.ctypes.data), 0) != 0
            hip.hipGetLastError()
        pt.SetOutImage(out2); pt.ResetImage(); pt.RenderFrame()
        dev2 = np.zeros((48, 64, 3), np.uint8); ctx = pt.context(); ctx.L.ptk_resolve_rgb8(ctx.h, dev2.ctypes.data)
        assert np.array_equal(out2, dev2) and out2.any()
        pt.SetOutImage(None)
        pt.RenderFrame()                                                               # unbound again: nothing may touch out2
        assert np.array_equal(out2, dev2)
        # ... nor when the unbinding coincides with a frame change and a reset (found by tools/soak_api.py: the binding was
        # then kept, and ResetImage cleared a buffer the caller had already freed)
        pt.SetOutImage(out2); pt.RenderFrame()
        out2[:] = 201
        pt.SetOutImage(None); pt.SetTraceDepth(3); pt.ResetImage(); pt.RenderFrame(); pt.RenderFrame()
        assert (out2 == 201).all()
        pt.close()
        del out, out2


def test_textures_set_after_the_build_take_effect_even_where_they_make_nans(tmp_path, oracle_mod):
    """Set...TextureForElement after BuildBVH and a first render (pathtracer.cpp:147-241: the triangles point at the materials, so the
    next RenderFrame sees the new Image): a present file and a MISSING one (an Image without data samples as 0), on the diffuse,
    normal and roughness slots.  The Cornell shell has no texture coordinates, so its tangent frames are NaN and a normal map
    turns every path through that wall into NaN - in the reference (DirectIllumimation's `dot <= 0` test lets a NaN pass,
    :519), in the oracle, and in the kernels: the same pixels are NaN, every other pixel is equal bit for bit (found by
    tools/soak_api.py: the kernel's negated form of that test had sent NaNs the other way)."""
    from pbrpathtracer_amd import scenes as S
    from pbrpathtracer_amd.pathtracer import PathTracer, camera_from_scene
    pts, scene, _ = S.build_config("C1", str(tmp_path), width=96, height=64)
    good = str(tmp_path / "t.ppm"); S.write_ppm(good, S.tex_checker(16, 4))
    cam = camera_from_scene(scene); cam["aperture"] = 0.0
    nan_cases = 0
    for tf in (good, str(tmp_path / "missing.ppm")):
        for slot in (0, 1, 3):
            for flat in (1, 0):
                pt = PathTracer(0); pt.LoadSceneFile(pts); pt.SetCameraAperture(0.0); pt.SetSeed(3)
                pt.RenderFrames(1)
                pt.context().set_option("flat", flat)
                pt._set_tex(slot, 0, 4, tf)
                pt.ResetImage(); pt.RenderFrames(3)
                assert pt.LastError() == ""
                got = pt.ReadAccumulation(); st = pt.StagedScene()
                assert st["materials"]["tex"][4][slot] == 0 and len(st["textures"]) == 1
                o = oracle_mod.Oracle(st)
                ocam = oracle_mod.make_camera(cam["pos"], cam["dir"], cam["up"], cam["focal"], cam["fovy"], cam["focal_dist"], cam["aperture"])
                ref, _ = o.render(ocam, 96, 64, pt.GetTraceDepth(), 0, 3, 3)
                o.close()
                assert np.array_equal(np.isnan(ref), np.isnan(got)) and np.array_equal(ref, got, equal_nan=True), (tf, slot, flat)
                nan_cases += int(np.isnan(ref).any())
                pt.close()
    assert nan_cases == 4                                               # the normal slot, both files, both kernels


def test_two_tracers_render_concurrently_from_two_threads(tmp_path, oracle_mod):
    """Two PathTracer instances on the same GPU driven by two host threads at once (each context has its own streams, queues and
    buffers; nothing is shared but the device): scene loads, BVH builds, renders in small batches and hand-offs interleave
    freely, and each accumulator equals the oracle's for its own scene."""
    import threading
    from pbrpathtracer_amd import scenes as S
    from pbrpathtracer_amd.pathtracer import PathTracer, camera_from_scene
    jobs = []
    for k, (cfg, kw) in enumerate([("C1", dict(width=88, height=64, depth=4)), ("C4", dict(width=72, height=56, depth=5, grid=16))]):
        d = tmp_path / f"s{k}"; d.mkdir()
        jobs.append(S.build_config(cfg, str(d), **kw))
    results = [None, None]; errors = []

    def work(k):
        try:
            pts, scene, _ = jobs[k]
            for rep in range(3):                                        # (loads and builds overlap the other thread's renders)
                pt = PathTracer(0); pt.LoadSceneFile(pts); pt.SetSeed(100 + k); pt.SetCameraAperture(0.0)
                out = pt.AllocOutImage(); pt.SetOutImage(out); pt.ResetImage()
                for _ in range(12):
                    pt.RenderFrames(1 + (k + rep) % 3)
                assert pt.LastError() == ""
                results[k] = (pt.ReadAccumulation(), pt.GetSamples(), pt.StagedScene(), np.array(out), pt.GetResolution(), pt.GetTraceDepth())
                pt.SetOutImage(None)
                if rep < 2: pt.close()
                else: results[k] += (pt,)
        except Exception as e:                                          # surfaced in the main thread
            errors.append((k, repr(e)))
    threads = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in threads: t.start()
    for t in threads: t.join(timeout=300)
    assert not errors and all(not t.is_alive() for t in threads), errors
    for k in range(2):
        got, n, staged, out8, (W, H), Dp, pt = results[k]
        cam = camera_from_scene(jobs[k][1]); cam["aperture"] = 0.0
        o = oracle_mod.Oracle(staged)
        ocam = oracle_mod.make_camera(cam["pos"], cam["dir"], cam["up"], cam["focal"], cam["fovy"], cam["focal_dist"], cam["aperture"])
        ref, ref8 = o.render(ocam, W, H, Dp, 0, n, 100 + k)
        o.close()
        assert np.array_equal(got, ref) and np.array_equal(out8, ref8), k
        pt.close()


def test_device_resident_handoff_buffer(tmp_path):
    """ptk_bind_out_device (N3 without the PCIe hop: the frameTex <- texData upload of main.cpp:3026-3029 for a display path
    that lives on the GPU): the accumulate kernel's 8-bit image lands in a caller-owned DEVICE buffer - here a torch tensor -
    and equals the device's resolved frame after every render, through camera moves, resets and a new resolution; host
    memory, another binding and a resolution change are handled as documented."""
    import torch
    from pbrpathtracer_amd import scenes as S
    from pbrpathtracer_amd.pathtracer import PathTracer
    pts, scene, _ = S.build_config("C1", str(tmp_path), width=100, height=76, depth=4)       # (a row of 300 bytes: dword rows)
    pt = PathTracer(0); pt.LoadSceneFile(pts); pt.SetSeed(4); pt.SetCameraAperture(0.0)
    W, H = pt.GetResolution()
    ctx = None
    buf = torch.full((H, W, 3), 99, dtype=torch.uint8, device="cuda:0")
    torch.cuda.synchronize()
    pt.SetOutDeviceImage(buf.data_ptr())                                # (PathTracer::SetOutDeviceImage -> ptk_bind_out_device)
    for frame in range(9):
        if frame == 3:
            pt.SetCamera((0.6, 0.2, -3.5), (-0.15, -0.05, 1.0), (0, 1, 0))
        if frame == 6:
            pt.ResetImage()
        pt.RenderFrame()                                                # (returns with the frame in the buffer)
        assert pt.LastError() == ""
        got = buf.cpu().numpy()
        ctx = ctx or pt.context()
        dev = ctx.resolve_rgb8()
        assert np.array_equal(got, dev), frame
        if frame == 6:
            assert pt.GetSamples() == 1
    assert dev.any()
    # an odd row length (the byte path of the accumulate kernel) and an offset, unaligned target
    pt.SetOutDeviceImage(None)                                          # (the caller's buffer has the old size: main.cpp:3425-3446)
    pt.SetResolution((67, 41)); pt.ResetImage(); pt.RenderFrame()
    assert np.array_equal(buf.cpu().numpy(), dev)                       # unbound: untouched
    raw = torch.zeros(67 * 41 * 3 + 8, dtype=torch.uint8, device="cuda:0"); torch.cuda.synchronize()
    ctx = pt.context(); ctx.bind_out_device(raw.data_ptr() + 1)          # (the C-ABI call itself)
    pt.RenderFrame(); pt.RenderFrame(); ctx.synchronize()
    got = raw.cpu().numpy()
    assert np.array_equal(got[1:1 + 67 * 41 * 3].reshape(41, 67, 3), ctx.resolve_rgb8()) and got[0] == 0 and not got[1 + 67 * 41 * 3:].any()
    # a host buffer takes the binding over; unbinding leaves the device buffer alone
    out = np.zeros((41, 67, 3), np.uint8)
    pt.SetOutImage(out); pt.RenderFrame()
    assert np.array_equal(out, ctx.resolve_rgb8()) and np.array_equal(raw.cpu().numpy(), got)
    pt.SetOutImage(None); pt.RenderFrame()
    # not device memory / nothing bound yet: errors, and the context goes on working
    from pbrpathtracer_amd import ptk
    host = np.zeros(67 * 41 * 3, np.uint8)
    with pytest.raises(ptk.PtkError, match="not a device allocation"):
        ctx.bind_out_device(host.ctypes.data)
    ctx.bind_out_device(None)
    pt.RenderFrame()
    assert pt.LastError() == ""
    pt.close()


def test_gl_buffer_binding_needs_a_current_context(tmp_path):
    """ptk_bind_gl_buffer / PathTracer::SetOutGLBuffer on a headless box: no OpenGL context is current, so the binding is
    refused with an error (never a crash inside the runtime) and rendering goes on into the library's own image.  The
    mapped path itself (hipGraphicsMapResources -> the same device hand-off as ptk_bind_out_device -> unmap) cannot run
    here: NOT exercised on hardware."""
    from pbrpathtracer_amd import ptk
    from pbrpathtracer_amd import scenes as S
    from pbrpathtracer_amd.pathtracer import PathTracer
    pts, scene, _ = S.build_config("C1", str(tmp_path), width=64, height=48, depth=3)
    pt = PathTracer(0); pt.LoadSceneFile(pts); pt.SetSeed(1)
    pt.RenderFrame()
    ctx = pt.context()
    ref = ctx.resolve_rgb8()
    with pytest.raises(ptk.PtkError, match="no OpenGL context is current"):
        ctx.bind_gl_buffer(7)
    ctx.bind_gl_buffer(0)                                               # unbinding nothing is fine
    pt.SetOutGLBuffer(7); pt.RenderFrame()
    assert "no OpenGL context" in pt.LastError()
    pt.SetOutGLBuffer(0); pt.ResetImage(); pt.RenderFrame()
    assert np.array_equal(ctx.resolve_rgb8(), ref)
    pt.close()


@pytest.mark.parametrize("cfg,world", [("C1", 1), ("C2", 97), ("C3", 149), ("C4", 499), ("C5", 1999)])
def test_full_size_spot_check_against_oracle(tmp_path, oracle_mod, cfg, world):
    """The BASELINE configs at their FULL size and sample count (persistent waves, live-quadrant list, two passes for
    C4): the oracle renders the tiles of one rank of a `world`-way split (a few dozen of the frame's tiles, spread
    over it; C1, the reference's CPU-runnable case, whole) and the GPU's accumulator must equal it there bit for bit."""
    from pbrpathtracer_amd import scenes as S
    from pbrpathtracer_amd import distributed as D
    from pbrpathtracer_amd.pathtracer import PathTracer
    pts, scene, spp = S.build_config(cfg, str(tmp_path))
    from pbrpathtracer_amd.pathtracer import camera_from_scene
    pt = PathTracer(0); pt.LoadSceneFile(pts); pt.SetSeed(21)
    cam = camera_from_scene(scene)
    if scene.pinhole:
        pt.SetCameraAperture(0.0)               # as bench.py does (the .pts carries F = 1e9)
        cam["aperture"] = 0.0
    W, H = pt.GetResolution(); Dp = pt.GetTraceDepth()
    pt.RenderFrames(spp)
    assert pt.LastError() == "" and pt.GetSamples() == spp
    got = pt.ReadAccumulation()
    o = oracle_mod.Oracle(pt.StagedScene())
    ocam = oracle_mod.make_camera(cam["pos"], cam["dir"], cam["up"], cam["focal"], cam["fovy"], cam["focal_dist"], cam["aperture"])
    rank = world // 3
    ref, _ = o.render(ocam, W, H, Dp, 0, spp, 21, rank=rank, world=world, want_rgb8=False)
    mask = D.tile_owner_mask(W, H, rank, world)[::-1]           # accumulator rows are bottom-up
    assert mask.sum() >= 16 * 16 * 4 and ref[mask].any()
    assert np.array_equal(got[mask], ref[mask])
    pt.close()


def test_headline_config_whole_frame_at_full_sample_count(tmp_path, oracle_mod):
    """bench.py's headline workload - C2, 1280x720, 8 bounces, 256 spp: 236 M samples - EVERY pixel against the oracle, float
    accumulator and RGB8, bit for bit (the oracle takes ~10 s of the GPU box's host cores for it)."""
    from pbrpathtracer_amd import scenes as S
    from pbrpathtracer_amd.pathtracer import PathTracer, camera_from_scene
    pts, scene, spp = S.build_config("C2", str(tmp_path))
    pt = PathTracer(0); pt.LoadSceneFile(pts); pt.SetSeed(33)
    cam = camera_from_scene(scene)
    pt.SetCameraAperture(0.0); cam["aperture"] = 0.0           # as bench.py does (the .pts carries F = 1e9)
    W, H = pt.GetResolution(); Dp = pt.GetTraceDepth()
    out = np.zeros((H, W, 3), np.uint8); pt.SetOutImage(out)
    pt.RenderFrames(spp)
    assert pt.LastError() == "" and pt.GetSamples() == spp == 256 and (W, H, Dp) == (1280, 720, 8)
    got = pt.ReadAccumulation()
    o = oracle_mod.Oracle(pt.StagedScene())
    ocam = oracle_mod.make_camera(cam["pos"], cam["dir"], cam["up"], cam["focal"], cam["fovy"], cam["focal_dist"], cam["aperture"])
    ref, ref8 = o.render(ocam, W, H, Dp, 0, spp, 33)
    assert np.array_equal(got, ref)
    assert np.array_equal(out, ref8)
    assert (ref != 0).any(axis=2).mean() > 0.15                 # (the box covers about a fifth of this camera's frame)
    pt.close()


@pytest.mark.parametrize("cfg,spp", [("C3", 12), ("C4", 12), ("C5", 6)])
def test_bvh_configs_whole_frame(tmp_path, oracle_mod, cfg, spp):
    """EVERY pixel of the BVH-walk configs at their full resolution (the sample count reduced so that the oracle needs seconds,
    not minutes, of the box's host cores): thin lens + textures + opacity (C3), 70 k triangles (C4), 1 M triangles at depth 12
    on the device-built tree (C5) - float accumulator and RGB8 against the oracle, bit for bit.  (The spot checks above run
    the full sample counts on a spread of tiles; tools/fullframe_check.py is the one-off at 64-128 spp.)"""
    from pbrpathtracer_amd import scenes as S
    from pbrpathtracer_amd.pathtracer import PathTracer, camera_from_scene
    pts, scene, _ = S.build_config(cfg, str(tmp_path))
    pt = PathTracer(0); pt.LoadSceneFile(pts); pt.SetSeed(41)
    cam = camera_from_scene(scene)
    if scene.pinhole:
        pt.SetCameraAperture(0.0); cam["aperture"] = 0.0
    W, H = pt.GetResolution(); Dp = pt.GetTraceDepth()
    out = np.zeros((H, W, 3), np.uint8); pt.SetOutImage(out)
    pt.RenderFrames(spp)
    assert pt.LastError() == "" and pt.GetSamples() == spp
    got = pt.ReadAccumulation()
    o = oracle_mod.Oracle(pt.StagedScene())
    ocam = oracle_mod.make_camera(cam["pos"], cam["dir"], cam["up"], cam["focal"], cam["fovy"], cam["focal_dist"], cam["aperture"])
    ref, ref8 = o.render(ocam, W, H, Dp, 0, spp, 41)
    assert np.array_equal(got, ref)
    assert np.array_equal(out, ref8)
    assert (ref != 0).any(axis=2).mean() > 0.1
    pt.SetOutImage(None)
    pt.close()


def test_c5_tile_split_over_8_ranks_on_one_gpu(tmp_path, oracle_mod):
    """BASELINE configs[4]'s shape - the 1 M-triangle frame tile-split over 8 ranks and gathered - with this one GPU playing
    every rank in turn: each rank's share is rendered (ptk_set_tile), packed by the exchange's pack kernel, and the eight
    packed buffers are scattered by its unpack kernel; the result equals the single-GPU frame bit for bit, and the
    oracle confirms a spread of its tiles.  (Samples per pixel reduced to 32: the split, not the sample count, is the subject.)"""
    from pbrpathtracer_amd import scenes as S
    from pbrpathtracer_amd import distributed as D
    from pbrpathtracer_amd.pathtracer import PathTracer, camera_from_scene
    pts, scene, _ = S.build_config("C5", str(tmp_path))
    spp = 32
    pt = PathTracer(0); pt.LoadSceneFile(pts); pt.SetSeed(8); pt.SetCameraAperture(0.0)
    W, H = pt.GetResolution(); Dp = pt.GetTraceDepth()
    pt.RenderFrames(spp)
    whole = pt.ReadAccumulation()
    ctx = pt.context()
    packed = []
    for r in range(8):
        ctx.set_tile(r, 8); ctx.reset(); ctx.render(0, spp, 8)
        part = ctx.read_accum()
        mask = D.tile_owner_mask(W, H, r, 8)[::-1]
        assert not part[~mask].any() and np.array_equal(part[mask], whole[mask])
        packed.append(ctx.probe_pack(r, 8))
    ctx.set_tile(0, 1)
    assert np.array_equal(ctx.probe_unpack(8, np.concatenate(packed)), whole)
    cam = camera_from_scene(scene); cam["aperture"] = 0.0
    o = oracle_mod.Oracle(pt.StagedScene())
    ocam = oracle_mod.make_camera(cam["pos"], cam["dir"], cam["up"], cam["focal"], cam["fovy"], cam["focal_dist"], cam["aperture"])
    ref, _ = o.render(ocam, W, H, Dp, 0, spp, 8, rank=666, world=1999, want_rgb8=False)
    m = D.tile_owner_mask(W, H, 666, 1999)[::-1]
    assert ref[m].any() and np.array_equal(whole[m], ref[m])
    pt.close()


def test_native_cpp_render_loop_matches_python_mirror(tmp_path):
    """tests/cpp/dropin_render_loop.cpp - the reference's render loop written in C++ against include/pathtracer.h and
    libptk.so - renders the Cornell box into its caller-owned RGB8 buffer; the same calls through the Python mirror give
    the same bytes."""
    import subprocess
    from test_host_cpu import _build_dropin_example
    from pbrpathtracer_amd import scenes as S
    from pbrpathtracer_amd.pathtracer import PathTracer
    groups, mats = S.cornell_groups()
    obj = str(tmp_path / "cornell.obj")
    S.write_obj(obj, groups)
    W, H, D, frames = 112, 80, 5, 6
    exe = _build_dropin_example(tmp_path)
    raw = str(tmp_path / "out.rgb")
    r = subprocess.run([exe, obj, str(W), str(H), str(D), str(frames), raw], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "samples %d triangles 12" % frames in r.stdout, r.stdout + r.stderr
    got = np.fromfile(raw, np.uint8).reshape(H, W, 3)
    pt = PathTracer(0)
    pt.ClearScene(); pt.LoadObject(obj)
    for e, m in enumerate(mats):
        pt.SetMaterial(0, e, m)
    pt.BuildBVH()
    pt.SetResolution((W, H)); pt.SetTraceDepth(D)
    pt.SetCamera((0, 0, -3.5), (0, 0, 1), (0, 1, 0)); pt.SetProjection(0.05, 70.0)
    pt.SetCameraFocalDist(3.5); pt.SetCameraAperture(0.0)
    out = np.zeros((H, W, 3), np.uint8)
    pt.SetOutImage(out); pt.ResetImage()
    for _ in range(frames):
        pt.RenderFrame()
    assert pt.GetSamples() == frames and out.any()
    assert np.array_equal(got, out)
    pt.close()


def test_state_changes_between_overlapped_renders(tmp_path, oracle_mod):
    """Renders are asynchronous and consecutive batches overlap on two internal streams; whatever the host changes in
    between - camera, resolution, trace depth, seed, scene - the next batch must see it, and two tracers on one GPU must
    not disturb each other.  Every stage is compared with the oracle, bit for bit."""
    from pbrpathtracer_amd import scenes as S
    from pbrpathtracer_amd.pathtracer import PathTracer, camera_from_scene
    pts, scene, _ = S.build_config("C4", str(tmp_path / "a"), width=96, height=64, grid=16)
    pts2, scene2, _ = S.build_config("C1", str(tmp_path / "b"), width=80, height=48)
    pt = PathTracer(0); pt.LoadSceneFile(pts)
    other = PathTracer(0); other.LoadSceneFile(pts2)

    def check(p, sc, W, H, D, cam, first, spp, seed):
        o = oracle_mod.Oracle(p.StagedScene())
        ocam = oracle_mod.make_camera(cam["pos"], cam["dir"], cam["up"], cam["focal"], cam["fovy"], cam["focal_dist"], cam["aperture"])
        ref, _ = o.render(ocam, W, H, D, first, spp, seed, want_rgb8=False)
        return ref

    cam = camera_from_scene(scene); cam["aperture"] = 0.0
    pt.SetCameraAperture(0.0); pt.SetSeed(3)
    W, H = pt.GetResolution(); D = pt.GetTraceDepth()
    # back-to-back batches, no synchronisation in between, the other tracer rendering in the middle
    pt.RenderFrames(3); other.RenderFrames(2); pt.RenderFrames(2); pt.RenderFrames(4)
    assert np.array_equal(pt.ReadAccumulation(), check(pt, scene, W, H, D, cam, 0, 9, 3))
    # new camera: the cached camera hits and the live-quadrant list must be rebuilt before the next trace kernel
    cam["pos"] = np.array([0.3, 0.2, -3.0], np.float32)
    pt.SetCamera(cam["pos"], cam["dir"], cam["up"]); pt.ResetImage()
    pt.RenderFrames(2); pt.RenderFrames(3)
    assert np.array_equal(pt.ReadAccumulation(), check(pt, scene, W, H, D, cam, 0, 5, 3))
    # new resolution, depth and seed
    pt.SetResolution((70, 50)); pt.SetTraceDepth(3); pt.SetSeed(8); pt.ResetImage()
    pt.RenderFrames(4)
    assert np.array_equal(pt.ReadAccumulation(), check(pt, scene, 70, 50, 3, cam, 0, 4, 8))
    # thin lens: the primary-hit cache and its live masks no longer apply
    cam["aperture"] = 0.02
    pt.SetCameraAperture(0.02); pt.ResetImage(); pt.RenderFrames(3)
    assert np.array_equal(pt.ReadAccumulation(), check(pt, scene, 70, 50, 3, cam, 0, 3, 8))
    # another scene in the same tracer
    pt.ClearScene(); pt.LoadSceneFile(pts2); pt.SetCameraAperture(0.0); pt.SetSeed(8)
    cam2 = camera_from_scene(scene2); cam2["aperture"] = 0.0
    W2, H2 = pt.GetResolution(); D2 = pt.GetTraceDepth()
    pt.RenderFrames(5)
    assert np.array_equal(pt.ReadAccumulation(), check(pt, scene2, W2, H2, D2, cam2, 0, 5, 8))
    # the other tracer was left alone meanwhile
    other.SetCameraAperture(0.0)          # (its two frames so far used the .pts aperture; start over with the pinhole)
    other.ResetImage(); other.RenderFrames(3)
    Wo, Ho = other.GetResolution()
    assert np.array_equal(other.ReadAccumulation(), check(other, scene2, Wo, Ho, other.GetTraceDepth(), cam2, 0, 3, 0))
    pt.close(); other.close()
