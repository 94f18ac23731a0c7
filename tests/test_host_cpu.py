"""CPU tests of the host layer and the C-ABI library (no GPU compute): the shared object loads and
exports every symbol the headers declare, the C++ scene staging reproduces the reference's staging
(golden vectors from the real reference), the BVH builder is structurally valid, .pts round-trips."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_golden, scene_from_golden


def test_library_exports_every_declared_symbol():
    from pbrpathtracer_amd import ptk, pathtracer
    L = ptk.load()
    for hdr, names in (("ptk.h", ptk.SYMBOLS), ("ptk_host.h", pathtracer.HOST_SYMBOLS)):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        declared = set(re.findall(r"\b(p(?:tk|th)_[a-z0-9_]+)\s*\(", text))
        assert declared == set(names), (hdr, declared ^ set(names))
        for n in names:
            assert hasattr(L, n), n


def test_no_cpu_fallback_without_gpu():
    """Without a GPU the product refuses loudly; it never renders on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pbrpathtracer_amd import ptk
    with pytest.raises(ptk.PtkError):
        ptk.Context(0)


def test_bvh_builder_structure(tmp_path):
    exe = str(tmp_path / "test_bvh")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "pbrpathtracer_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "test_bvh.cpp"),
                           os.path.join(ROOT, "pbrpathtracer_amd", "csrc", "bvh_build.cpp"), "-o", exe, "-pthread"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_trs_and_euler_match_glm_0931():
    from pbrpathtracer_amd import pathtracer as P
    z = load_golden("tier_k.npz")
    for i in range(len(z["trs_loc"])):
        M = P.trs_matrix(z["trs_loc"][i], z["trs_rot"][i], z["trs_scl"][i]).reshape(16)
        assert np.allclose(M, z["trs_out"][i], rtol=0, atol=2e-6), i
        d, u = P.euler_camera(z["trs_rot"][i])
        assert np.allclose(np.concatenate([d, u]), z["euler_out"][i], rtol=0, atol=2e-6), i


def test_triangle_init_matches_reference():
    from pbrpathtracer_amd import pathtracer as P
    z = load_golden("tier_k.npz")
    for i in range(len(z["ti_in"])):
        assert np.array_equal(P.triangle_init(z["ti_in"][i]), z["ti_out"][i], equal_nan=True), i


def test_image_loader_and_sampler(tmp_path):
    from pbrpathtracer_amd import pathtracer as P, scenes as S
    z = load_golden("tier_k.npz")
    rgba = z["tx_rgba"]
    p = str(tmp_path / "t.ppm")
    S.write_ppm(p, rgba[..., :3])
    got = P.image_load(p)
    assert np.array_equal(got, rgba)                     # forced to 4 channels, alpha 255 (stbi_load(..., 4))
    for i, (u, v) in enumerate(z["tx_uv"]):
        uu = np.fmod(np.float32(u), np.float32(1)); uu = uu + np.float32(1) if uu < 0 else uu
        vv = np.fmod(np.float32(v), np.float32(1)); vv = vv + np.float32(1) if vv < 0 else vv
        if np.float32(7) * uu >= 7 or np.float32(5) * vv >= 5:
            continue                                      # reference over-read (image.cpp:71-77)
        assert np.array_equal(P.image_tex2d(float(u), float(v)), z["tx_out"][i]), i
    # > 1024: longest side becomes 1024 (image.cpp:47-60); size pinned, filtered values unpinned
    big = np.zeros((int(z["tx_big_in"][1]), int(z["tx_big_in"][0]), 3), np.uint8)
    p2 = str(tmp_path / "big.ppm"); S.write_ppm(p2, big)
    g = P.image_load(p2)
    assert (g.shape[1], g.shape[0]) == tuple(z["tx_big_out"])
    assert P.image_load(str(tmp_path / "missing.ppm")) is None


def test_png_decoder(tmp_path):
    import struct, zlib
    from pbrpathtracer_amd import pathtracer as P
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (9, 13, 4), dtype=np.uint8)

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)
    for ctype, ch in ((6, 4), (2, 3), (0, 1)):
        raw = b""
        src = img[..., :ch]
        prev = np.zeros((13 * ch,), np.int32)
        for y in range(9):
            row = src[y].reshape(-1).astype(np.int32)
            ft = y % 3                                   # exercise None / Sub / Up filters
            if ft == 0: enc = row
            elif ft == 1:
                left = np.concatenate([np.zeros(ch, np.int32), row[:-ch]]); enc = (row - left) & 255
            else: enc = (row - prev) & 255
            raw += bytes([ft]) + enc.astype(np.uint8).tobytes()
            prev = row
        png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 13, 9, 8, ctype, 0, 0, 0)) \
            + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")
        p = str(tmp_path / f"t{ctype}.png")
        open(p, "wb").write(png)
        got = P.image_load(p)
        exp = np.full((9, 13, 4), 255, np.uint8)
        if ch == 1: exp[..., :3] = src
        else: exp[..., :ch] = src
        assert np.array_equal(got, exp), ctype


def test_load_object_staging_matches_reference(tmp_path):
    """PathTracer::LoadObject + SetMaterial + BuildBVH staging (pathtracer.cpp:41-145, :243-274) vs the
    triangle array the real reference staged from the same OBJ + model matrix."""
    from pbrpathtracer_amd.pathtracer import PathTracer
    z = load_golden("tier_k_scene.npz")
    obj = str(tmp_path / "scene.obj")
    open(obj, "wb").write(z["obj_file"].tobytes())
    pt = PathTracer()
    pt.LoadObject(obj, z["model"].reshape(4, 4))
    assert pt.GetLoadedObjects() == [int(z["n_elements"])]
    for j, m in enumerate(z["materials_in"]):
        pt.SetMaterial(0, j, m)
    pt.SetMaterial(0, 99, z["materials_in"][0])         # bad ids are ignored (pathtracer.cpp:245-248)
    pt.SetMaterial(7, 0, z["materials_in"][0])
    s = pt.StagedScene()
    ref = scene_from_golden(z)
    assert pt.GetTriangleCount() == len(ref["verts"])
    # the reference sorts mTriangles while building its tree (mesh.cpp:171-176): compare as sets
    def key(a):
        return np.lexsort(np.round(a["verts"].astype(np.float64), 6).T[::-1])
    ia, ib = key(s), key(ref)
    for k in ("verts", "normals", "uvs", "tbn"):
        assert np.allclose(s[k][ia], ref[k][ib], rtol=0, atol=1e-6, equal_nan=True), k
    assert np.array_equal(s["smoothing"][ia], ref["smoothing"][ib])
    assert np.array_equal(s["material"][ia], ref["material"][ib])
    for f in ("type", "diffuse", "specular", "emissive", "emissive_intensity", "roughness", "reflectiveness", "translucency", "ior"):
        assert np.array_equal(s["materials"][f], ref["materials"][f]), f
    assert sorted(np.round(s["verts"][s["lights"]].sum(1), 5)) == sorted(np.round(ref["verts"][ref["lights"]].sum(1), 5))
    pt.close()


def test_obj_files_of_every_flavour_stage_like_the_reference(tmp_path):
    """PathTracer::LoadObject (pathtracer.cpp:41-145) = tinyobj::LoadObj 2.0.0 with triangulation, on OBJ files the way exporters
    write them: quads (cut along the shorter diagonal), convex / concave / star-shaped / collinear / bent polygons (tinyobj's ear
    clipping in its own projection plane, including the faces it gives up on), relative indices, v-only / v//vn / v/vt corners,
    `o` and `g` statements in every order (bare `g`, empty groups, several names, faces before the first group, `usemtl` between),
    smoothing groups (`s 1`, `s off`, `s` without a value), CRLF / tabs / comments / vertex colours / `l` and `p` statements /
    exotic number forms / no trailing newline / no faces at all.  The staged triangles - positions, normals, uvs, tangent
    frames, smoothing flags, element ids, IN FILE ORDER - the element count and the object and element names equal what the real
    reference staged (golden: oracle/gen_golden.py tier_k_obj_variants)."""
    from pbrpathtracer_amd.pathtracer import PathTracer
    z = load_golden("tier_k_obj_variants.npz")
    names = [str(n) for n in z["names"]]
    assert len(names) >= 18
    for name in names:
        p = str(tmp_path / (name + ".obj"))
        open(p, "wb").write(z["obj_" + name].tobytes())
        pt = PathTracer()
        pt.LoadObject(p, z["model"].reshape(4, 4))
        t = z["tris_" + name]; nel = int(z["elements_" + name])
        assert pt.GetLoadedObjects() == [nel], name
        assert pt.GetTriangleCount() == len(t), (name, pt.GetTriangleCount(), len(t))
        on, en = pt.GetNames(0)
        assert [on] + en == [str(x) for x in z["labels_" + name]][:1 + nel], name
        if len(t):
            s_ = pt.StagedScene()
            for k, (a, b) in dict(verts=(0, 9), normals=(9, 18), uvs=(18, 24), tbn=(24, 33)).items():
                assert np.allclose(s_[k], t[:, a:b], rtol=0, atol=1e-6, equal_nan=True), (name, k)
            assert np.array_equal(s_["smoothing"], (t[:, 33] != 0).astype(np.uint8)), name
            assert np.array_equal(s_["material"], t[:, 35].astype(np.int32)), name
        pt.close()


def test_object_names_follow_the_reference_for_odd_paths(tmp_path, monkeypatch):
    """LoadObject's object name (pathtracer.cpp:49-56): between the last '/' and the last '.', with the reference's quirks - no
    extension loses the last character, a dot that lies before the file name ("./mesh", "dir.v2/mesh") gives the whole file name.
    (Expected values taken from the reference itself with oracle/ref_harness.cpp's ref_name.)"""
    from pbrpathtracer_amd.pathtracer import PathTracer
    (tmp_path / "dir.with.dot").mkdir()
    monkeypatch.chdir(tmp_path)
    cases = {"plain.obj": "plain", "a.b.c.obj": "a.b.c", "noext": "noex", "./noext": "noext", "dir.with.dot/noext": "noext",
             "dir.with.dot/x.obj": "x", ".hidden": "", "dir.with.dot/.obj": "", "trail.": "trail", "sp ace.obj": "sp ace", "back\\slash.obj": "back\\slash"}
    for rel, want in cases.items():
        open(rel, "w").write("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 3\n")
        pt = PathTracer(); pt.LoadObject(rel)
        assert pt.GetLoadedObjects() == [1] and pt.GetNames(0)[0] == want, (rel, pt.GetNames(0))
        pt.close()


def test_pts_reader_and_scene_push(tmp_path):
    from pbrpathtracer_amd import scenes as S
    from pbrpathtracer_amd.pathtracer import PathTracer, lib
    pts, sc, spp = S.build_config("C1", str(tmp_path), width=64, height=48)
    pt = PathTracer()
    pt.LoadSceneFile(pts)
    assert pt.GetTriangleCount() == 12 and pt.GetResolution() == (64, 48) and pt.GetTraceDepth() == 4
    s = pt.StagedScene()
    assert len(s["lights"]) == 2 and len(s["materials"]) == 6
    assert np.allclose(sorted(s["materials"]["diffuse"][:, 0]), sorted([0.75, 0.75, 0.75, 0.75, 0.25, 0.75]))
    # write -> read -> write is a fixed point (the reference's own writer is stale)
    a, b = str(tmp_path / "a.pts"), str(tmp_path / "b.pts")
    assert lib().pth_pts_roundtrip(pts.encode(), a.encode()) == 0
    assert lib().pth_pts_roundtrip(a.encode(), b.encode()) == 0
    assert open(a).read() == open(b).read()
    with pytest.raises(Exception):
        pt.LoadSceneFile(str(tmp_path / "nope.pts"))
    bad = tmp_path / "bad.pts"; bad.write_text("Path Tracer Scene File\nVersion=1.9.0\n")
    with pytest.raises(Exception):
        pt.LoadSceneFile(str(bad))
    pt.close()


def test_textured_scene_staging(tmp_path):
    from pbrpathtracer_amd import scenes as S
    from pbrpathtracer_amd.pathtracer import PathTracer
    pts, sc, spp = S.build_config("C3", str(tmp_path), width=64, height=36, nu=8, nv=6, tex_size=16)
    pt = PathTracer()
    pt.LoadSceneFile(pts)
    s = pt.StagedScene()
    assert len(s["textures"]) >= 5 and s["texels"].size == sum(int(t["width"]) * int(t["height"]) * 4 for t in s["textures"])
    assert (s["materials"]["tex"] >= 0).sum() >= 8 and s["smoothing"].sum() > 0
    assert (s["materials"]["type"] == 1).sum() == 2
    pt.close()


def test_obj_reader_subset(tmp_path):
    """OBJ features the loader promises (tinyobj subset, pathtracer.cpp:41-145): quads and polygons
    (triangulated as tinyobj does: test_obj_files_of_every_flavour_stage_like_the_reference pins the triangles to the
    reference's), negative indices, v//vn and v/vt forms, `g` groups as elements, smoothing groups,
    faces before any group, CRLF line ends."""
    from pbrpathtracer_amd.pathtracer import PathTracer
    obj = tmp_path / "m.obj"
    obj.write_bytes(b"\r\n".join([
        b"v 0 0 0", b"v 1 0 0", b"v 1 1 0", b"v 0 1 0", b"v 0 0 1",
        b"vt 0 0", b"vt 1 0", b"vt 1 1", b"vt 0 1",
        b"vn 0 0 1",
        b"f 1 2 3",                          # before any group: element with empty name
        b"g quad", b"s 1",
        b"f 1/1/1 2/2/1 3/3/1 4/4/1",        # quad -> 2 triangles, smoothing group 1
        b"o penta", b"s off",
        b"f -5//1 -4//1 -3//1 -2//1 -1//1",  # negative indices, pentagon -> 3 triangles
        b"g empty_group",                    # no faces: does not become an element
        b"g last",
        b"f 1/1 2/2 5/3", b""]))
    pt = PathTracer()
    pt.LoadObject(str(obj))
    assert pt.GetLoadedObjects() == [4]
    assert pt.GetTriangleCount() == 1 + 2 + 3 + 1
    s = pt.StagedScene()
    assert list(s["material"]) == [0, 1, 1, 2, 2, 2, 3]
    assert list(s["smoothing"]) == [0, 1, 1, 0, 0, 0, 0]
    # x is negated on load (pathtracer.cpp:74), v flipped (pathtracer.cpp:88)
    assert np.allclose(s["verts"][0], [0, 0, 0, -1, 0, 0, -1, 1, 0])
    # the unit square's diagonals are equally long: tinyobj then cuts (v1 v2 v4) (v2 v3 v4)
    assert np.allclose(s["uvs"][1], [0, 1, 1, 1, 0, 0]) and np.allclose(s["uvs"][2], [1, 1, 1, 0, 0, 0])
    assert np.allclose(s["normals"][1][:3], [0, 0, 1])
    pt.LoadObject(str(tmp_path / "does_not_exist.obj"))      # parse failure is silently ignored (:47)
    assert pt.GetLoadedObjects() == [4]
    pt.ClearScene()
    assert pt.GetTriangleCount() == 0 and pt.GetLoadedObjects() == []
    pt.close()


def test_unorm8_double_product_is_exact():
    """The kernel computes byte/255.0f as (float)((double)b * (1.0/255.0)) (ptk_kernels.hip `unorm8`)."""
    b = np.arange(256, dtype=np.uint32)
    assert np.array_equal(b.astype(np.float32) / np.float32(255.0), (b.astype(np.float64) * (1.0 / 255.0)).astype(np.float32))


def test_png_export_flips_to_top_down(tmp_path):
    """ExportAt writes texData with stbi_flip_vertically_on_write(true) (main.cpp:765-767)."""
    from pbrpathtracer_amd import pathtracer as P
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (7, 11, 3), dtype=np.uint8)        # bottom-up buffer
    p = str(tmp_path / "o.png")
    assert P.export_png(p, img)
    back = P.image_load(p)                                         # our decoder reads it back top-down
    assert np.array_equal(back[..., :3], img[::-1]) and (back[..., 3] == 255).all()


def test_texture_decoders_match_stb_image(tmp_path):
    """Image::Load (image.cpp:38-61) = stbi_load(file, ..., 4): JPEG (baseline and progressive, 4:4:4 / 4:2:2 /
    4:2:0, grey, CMYK / YCCK / 4 components without Adobe marker, restart markers, odd sizes), PNG (RGB, RGBA, grey, grey+alpha, palette, 16-bit, 1-bit, Adam7 interlace,
    colour-key transparency), BMP (24/32-bit,
    8/4/1-bit palettes, 565 bit fields, 555, top-down, OS/2 header, all-zero alpha) and TGA (RGB/RGBA/grey/
    grey+alpha/palette, raw and RLE, 16-bit colour and palette, top-down) and GIF (first frame: interlaced, transparency index,
    a frame smaller than the screen with stb's background fill) and Radiance HDR (run-length coded and flat, reduced to 8 bits
    with gamma 2.2) files decode to the very RGBA8
    texels the reference's stb_image 2.27 produces (golden: oracle/gen_golden.py)."""
    from pbrpathtracer_amd import pathtracer as P
    z = load_golden("tier_k_images.npz")
    assert len(z["names"]) >= 64
    for name in z["names"]:
        name = str(name)
        p = str(tmp_path / (name + (".jpg" if name.startswith("jpg") else ".png")))
        open(p, "wb").write(z["file_" + name].tobytes())
        got = P.image_load(p)
        assert got is not None, name
        assert got.shape == z["rgba_" + name].shape, name
        assert np.array_equal(got, z["rgba_" + name]), (name, int(np.abs(got.astype(int) - z["rgba_" + name].astype(int)).max()))


def test_jpegs_with_uncommon_sampling_factors_decode_like_stb_image(tmp_path):
    """Baseline JPEGs whose sampling factors no common encoder writes - 4:1:1, 4:4:0, 4:1:0, 3x1, 1x4, 4x4, different factors for
    Cb and Cr, a sub-sampled LUMA plane, a 2x2 grey image, restart markers - through the generic MCU layout and the generic
    up-sampler: the reference's RGBA8 (golden: oracle/gen_golden.py tier_k_images_jpeg_sampling, files from its minimal encoder)."""
    from pbrpathtracer_amd import pathtracer as P
    z = load_golden("tier_k_images_jpeg_sampling.npz")
    names = [str(n) for n in z["names"]]
    assert len(names) >= 11
    for name in names:
        p = str(tmp_path / (name + ".jpg"))
        open(p, "wb").write(z["file_" + name].tobytes())
        got = P.image_load(p)
        assert int(z["ok_" + name]) == 1 and got is not None, name
        want = z["rgba_" + name]
        assert got.shape == want.shape and np.array_equal(got, want), name
        assert len(np.unique(want.reshape(-1, 4), axis=0)) > 8, name          # (a real picture, not a flat field)


def test_psd_and_pic_textures_decode_like_stb_image(tmp_path):
    """The last two formats of stbi_load (Image::Load, image.cpp:38-61): Photoshop PSD - 8 / 16 bit, raw and PackBits with no-op
    codes, 0-5 channels, the white-matte removal for partial alpha, a truncated file (zeros), blocks to skip - and Softimage PIC -
    uncompressed, pure and mixed run-length packets, channels split over chained packets, 16-bit run lengths - decode to the
    reference's RGBA8; and the files stb_image refuses (CMYK / grey / PSB / zip / 32 bit PSDs, bad run lengths, 16-bit or unknown
    PIC packets, overruns, truncation) yield no texture here either (golden: oracle/gen_golden.py tier_k_images_psd_pic)."""
    from pbrpathtracer_amd import pathtracer as P
    z = load_golden("tier_k_images_psd_pic.npz")
    names = [str(n) for n in z["names"]]
    assert len(names) >= 35 and sum(n.startswith("fail_") for n in names) >= 12
    for name in names:
        p = str(tmp_path / (name + (".psd" if "psd" in name else ".pic")))
        open(p, "wb").write(z["file_" + name].tobytes())
        got = P.image_load(p)
        if name.startswith("fail_"):
            assert got is None, name
            continue
        assert got is not None, name
        want = z["rgba_" + name]
        assert got.shape == want.shape, name
        assert np.array_equal(got, want), (name, int(np.abs(got.astype(int) - want.astype(int)).max()))
    # the generator's own inputs, so that the fixture is known to hold real images and not noise stb happened to accept
    a = z["rgba_psd_rgba8_raw_matte"]; b = z["rgba_pic_rgba_raw"]
    assert np.array_equal(a[..., 3], b[..., 3]) and np.array_equal(z["rgba_psd_rgb8_raw"][..., :3], b[..., :3])
    assert np.array_equal(z["rgba_psd_rgb8_rle"], z["rgba_psd_rgb8_raw"]) and np.array_equal(z["rgba_pic_rgb_mixed"], z["rgba_pic_rgb_raw"])
    assert (z["rgba_psd_rgba8_raw_matte"][..., :3] != z["rgba_psd_rgb8_raw"][..., :3]).any()        # the matte removal did something


def test_oversized_pic_header_is_refused_not_allocated(tmp_path):
    """a 124-byte Softimage PIC whose header claims 65535 x 65535 asked decode_pic for 17 GB (std::bad_alloc through the
    C ABI).  stb_image refuses such a file (stbi__mad3sizes_valid, stb_image.h:6440): no texture, no allocation, no exception - checked in
    a child process whose address space is capped at 4 GB, so that an attempted allocation would end it."""
    import subprocess
    import sys
    pic = bytearray(b"\x53\x80\xf6\x34") + bytearray(84) + bytearray(b"PICT") + bytes([0xff, 0xff, 0xff, 0xff]) + bytearray(8)
    pic += bytes([0, 8, 0, 0xe0]) + bytearray(16)                  # one uncompressed RGB packet, then too little data
    path = str(tmp_path / "huge.pic")
    open(path, "wb").write(bytes(pic))
    code = ("import resource, sys; resource.setrlimit(resource.RLIMIT_AS, (4 << 30, 4 << 30)); sys.path.insert(0, %r); "
            "from pbrpathtracer_amd import pathtracer as P; r = P.image_load(%r); print('refused' if r is None else r.shape)") % (ROOT, path)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "refused", (out.returncode, out.stdout, out.stderr[-400:])


def _write_png(path, a):
    """Minimal PNG writer (8-bit grey / RGB / RGBA, filter 0) so the test needs no imaging library."""
    import struct
    import zlib
    a = np.ascontiguousarray(a, np.uint8)
    h, w = a.shape[:2]
    ctype = {1: 0, 3: 2, 4: 6}[1 if a.ndim == 2 else a.shape[2]]
    raw = b"".join(b"\x00" + a[y].tobytes() for y in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ctype, 0, 0, 0))
                           + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def test_files_cut_short_are_treated_as_stb_image_treats_them(tmp_path):
    """Cut-short textures, as the reference's stb_image 2.27 handles them (found with tools/fuzz_images_vs_reference.py): a JPEG
    without its EOI marker and a PNG without its IEND chunk are refused (no texture), a PNG whose IEND chunk lost only its
    checksum decodes, a binary PNM decodes with the samples behind the cut reading 0 (there: whatever the heap held)."""
    from pbrpathtracer_amd import pathtracer as P
    z = load_golden("tier_k_images.npz")
    names = [str(n) for n in z["names"]]
    jpg = next(n for n in names if n.startswith("jpg_444")); png = next(n for n in names if n.startswith("png"))
    for name, cuts in ((jpg, (0.5, 0.9, -2)), (png, (0.5, 0.9, -13))):
        data = z["file_" + name].tobytes()
        for c in cuts:
            p = str(tmp_path / f"{name}_{c}.bin")
            open(p, "wb").write(data[:int(len(data) * c) if c > 0 else c])
            assert P.image_load(p) is None, (name, c)
    data = z["file_" + png].tobytes()
    p = str(tmp_path / "no_crc.png"); open(p, "wb").write(data[:-3])
    assert np.array_equal(P.image_load(p), z["rgba_" + png])
    a = (np.arange(7 * 5 * 3) % 251).astype(np.uint8).reshape(5, 7, 3)
    ppm = b"P6\n7 5\n255\n" + a.tobytes()
    p = str(tmp_path / "cut.ppm"); open(p, "wb").write(ppm[:-30])
    got = P.image_load(p)
    want = np.concatenate([a.reshape(-1), np.zeros(0, np.uint8)]).copy(); want[-30:] = 0
    assert got is not None and np.array_equal(got[..., :3].reshape(-1), want) and (got[..., 3] == 255).all()


def test_truncated_radiance_hdr_does_not_hang(tmp_path):
    """A run-length coded .hdr cut off inside a scanline: past the end every byte reads as 0, i.e. a zero-length run
    that never advances - the reference's stb_image spins forever there; Image::Load here gives up (texture samples as 0)."""
    from pbrpathtracer_amd import pathtracer as P
    head = b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 2 +X 8\n"
    row = bytes([2, 2, 0, 8]) + b"".join(bytes([128 + 8, 100 + k]) for k in range(4))     # four runs of 8
    good = str(tmp_path / "ok.hdr"); open(good, "wb").write(head + row + row)
    assert P.image_load(good) is not None
    for cut in (len(head) + 4, len(head) + 7, len(head) + len(row) + 5):
        bad = str(tmp_path / f"cut{cut}.hdr"); open(bad, "wb").write((head + row + row)[:cut])
        assert P.image_load(bad) is None, cut


def test_large_textures_reduce_like_stb_image_resize(tmp_path):
    """Image::Load on a file with a side > 1024 (image.cpp:47-60): the reference reduces it with
    stbir_resize_uint8 (stb_image_resize v0.97, default Mitchell downsampling, clamped edges).  The
    restatement in csrc/image.cpp reproduces the reference's reduced texels bit for bit: exact 1/2 scale,
    non-integer scales in x, in y, hard stripes (ringing + saturation), RGBA, grey, a 2-row result."""
    from pbrpathtracer_amd import pathtracer as P
    from resize_cases import RESIZE_CASES, resize_case_input
    z = load_golden("tier_k_resize.npz")
    assert [str(n) for n in z["names"]] == RESIZE_CASES
    for name in RESIZE_CASES:
        p = str(tmp_path / (name + ".png"))
        _write_png(p, resize_case_input(name))
        got = P.image_load(p)
        exp = z["rgba_" + name]
        assert got is not None and got.shape == exp.shape, (name, None if got is None else got.shape, exp.shape)
        assert max(got.shape[0], got.shape[1]) == 1024
        assert np.array_equal(got, exp), (name, int((got != exp).sum()))


def _build_dropin_example(tmp_path):
    exe = str(tmp_path / "dropin_render_loop")
    libdir = os.path.join(ROOT, "pbrpathtracer_amd")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "dropin_render_loop.cpp"),
                           "-L" + libdir, "-lptk", "-Wl,-rpath," + libdir, "-o", exe])
    return exe


def test_dropin_header_compiles_and_links_natively(tmp_path):
    """include/pathtracer.h + libptk.so serve a plain C++ translation unit written like the reference's render loop
    (tests/cpp/dropin_render_loop.cpp): it compiles with g++ alone (no HIP, no GL, no glm install) and links."""
    exe = _build_dropin_example(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 2 and "usage" in out.stderr


def test_committed_counter_files_belong_to_the_kernels_as_built():
    """profiles/traffic_C*.json (what bench.py's roofline.traffic is read from) and profiles/r04/pmc_sq_*.json carry the sha256
    of the kernel sources they were collected for; bench.py ignores them once the sources change.  This keeps the committed
    set honest: whoever edits csrc/ptk_kernels.hip or csrc/ptk_device.h re-runs tools/profile_round.sh (or sees this fail)."""
    import glob
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    sha = bench.kernel_source_sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "traffic_C*.json")) + glob.glob(os.path.join(ROOT, "profiles", "r04", "pmc_sq_trace_kernel_C*.json")) +
                   glob.glob(os.path.join(ROOT, "profiles", "r04", "overlap_trace_C2.json")))
    assert len(files) >= 8
    for f in files:
        assert json.load(open(f)).get("kernel_source_sha256") == sha, f + " was collected for other kernel sources"
