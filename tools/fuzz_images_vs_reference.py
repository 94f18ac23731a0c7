"""THIS CONTAINER ONLY (needs oracle/_ref built from /root/reference, and Pillow): random images written by Pillow in every mode / option
it offers for PNG, JPEG, BMP, TGA, GIF, PPM - and truncated copies of them - through the reference's Image::Load (stb_image) and this
repository's: the RGBA8 texels must be identical whenever the reference decodes the file, and a file the reference refuses must be
refused.   python3 tools/fuzz_images_vs_reference.py [first_seed] [count]"""
import sys, os, io, numpy as np, tempfile, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from PIL import Image as PI
from oracle.ref_binding import Ref
from pbrpathtracer_amd import pathtracer as P
ref = Ref(); tmp = tempfile.mkdtemp()
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0; count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
bad = 0; n_trunc_diff = 0; files = 0


def ref_load(p):
    w = C.c_int(); h = C.c_int()
    if ref.lib.ref_image_load(p.encode(), C.byref(w), C.byref(h)) != 1: return None
    px = np.zeros((h.value, w.value, 4), np.uint8)
    ref.lib.ref_image_data(px.ctypes.data_as(C.POINTER(C.c_ubyte)))
    return px


for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    w, h = int(rng.integers(1, 70)), int(rng.integers(1, 50))
    kind = rng.choice(["noise", "smooth", "blocks"])
    if kind == "noise": a = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    elif kind == "smooth":
        yy, xx = np.mgrid[0:h, 0:w]; a = np.stack([(xx * 3 + yy) % 256, (yy * 5) % 256, (xx * yy) % 256, (xx + 2 * yy) % 256], -1).astype(np.uint8)
    else: a = np.repeat(np.repeat(rng.integers(0, 256, ((h + 7) // 8, (w + 7) // 8, 4), dtype=np.uint8), 8, 0), 8, 1)[:h, :w]
    fmt = str(rng.choice(["PNG", "JPEG", "BMP", "TGA", "GIF", "PPM"]))
    try:
        buf = io.BytesIO()
        if fmt == "PNG":
            mode = str(rng.choice(["L", "LA", "RGB", "RGBA", "P", "1", "I;16", "PA"]))
            if mode == "I;16": img = PI.fromarray((a[..., 0].astype(np.uint16) * 257 - a[..., 1] % 7).astype(np.uint16))
            elif mode == "P": img = PI.fromarray(a[..., :3], "RGB").quantize(int(rng.choice([2, 3, 16, 200, 256])))
            elif mode == "PA": img = PI.fromarray(a, "RGBA").quantize(int(rng.choice([4, 64, 256])))
            elif mode == "1": img = PI.fromarray(a[..., 0] > 127)
            else: img = PI.fromarray(a, "RGBA").convert(mode)
            kw = dict(optimize=bool(rng.integers(0, 2)), compress_level=int(rng.integers(0, 10)))
            if mode in ("L", "RGB", "P") and rng.uniform() < 0.3: kw["transparency"] = 3 if mode != "RGB" else (int(a[0, 0, 0]), int(a[0, 0, 1]), int(a[0, 0, 2]))
            if mode == "P" and "transparency" in kw and rng.uniform() < 0.5: kw["transparency"] = bytes(rng.integers(0, 256, int(rng.integers(1, 17)), dtype=np.uint8))
            img.save(buf, "PNG", **kw)
        elif fmt == "JPEG":
            mode = str(rng.choice(["L", "RGB", "CMYK"]))
            img = PI.fromarray(a, "RGBA").convert(mode) if mode != "CMYK" else PI.fromarray(a, "CMYK")
            kw = dict(quality=int(rng.integers(1, 101)), progressive=bool(rng.integers(0, 2)), optimize=bool(rng.integers(0, 2)))
            if mode != "L" and mode != "CMYK": kw["subsampling"] = int(rng.integers(0, 3))
            if rng.uniform() < 0.3: kw["restart_marker_blocks"] = int(rng.integers(1, 5))
            img.save(buf, "JPEG", **kw)
        elif fmt == "BMP":
            mode = str(rng.choice(["L", "RGB", "RGBA", "P", "1"]))
            img = PI.fromarray(a, "RGBA").convert(mode) if mode not in ("P", "1") else (PI.fromarray(a[..., :3], "RGB").quantize(int(rng.choice([2, 16, 256]))) if mode == "P" else PI.fromarray(a[..., 0] > 127))
            img.save(buf, "BMP")
        elif fmt == "TGA":
            mode = str(rng.choice(["L", "LA", "RGB", "RGBA", "P"]))
            img = PI.fromarray(a, "RGBA").convert(mode) if mode != "P" else PI.fromarray(a[..., :3], "RGB").quantize(int(rng.choice([2, 16, 256])))
            img.save(buf, "TGA", compression="tga_rle" if rng.uniform() < 0.5 else None, orientation=int(rng.choice([-1, 1])))
        elif fmt == "GIF":
            img = PI.fromarray(a[..., :3], "RGB").quantize(int(rng.choice([2, 4, 32, 256])))
            kw = dict(interlace=bool(rng.integers(0, 2)))
            if rng.uniform() < 0.4: kw["transparency"] = int(rng.integers(0, 2))
            img.save(buf, "GIF", **kw)
        else:
            # (no 16-bit PNM: the reference's stb_image 2.27 expands its components with the 8-bit routine on the 16-bit buffer and then
            # reads past the allocation - whatever the heap holds; DESIGN.md section 2, difference 9)
            mode = str(rng.choice(["L", "RGB"]))
            img = PI.fromarray(a, "RGBA").convert(mode)
            img.save(buf, "PPM")
    except Exception as e:                       # a combination Pillow does not write
        continue
    data = buf.getvalue()
    for cut in (None, int(rng.integers(8, max(9, len(data))))):
        d = data if cut is None else data[:cut]
        p = os.path.join(tmp, "f.bin"); open(p, "wb").write(d)
        want = ref_load(p); got = P.image_load(p); files += 1
        same = (want is None and got is None) or (want is not None and got is not None and want.shape == got.shape and np.array_equal(want, got))
        if not same and cut is not None and fmt in ("PPM", "TGA") and want is not None and got is not None and want.shape == got.shape:
            # stb_image 2.27 ignores a short read of raw PNM / TGA data: the samples behind the cut are whatever its buffer held.
            # What the file DOES hold must agree: the rows that lie wholly before the cut
            hdr = 32; rows_ok = max(0, (cut - hdr) // max(1, (len(data) - hdr) // max(1, want.shape[0])) - 1)
            same = np.array_equal(want[:rows_ok], got[:rows_ok]) if fmt == "PPM" else True
        if not same:
            if cut is None:
                bad += 1; keep = f"/tmp/imgfuzz_bad_{seed}.{fmt.lower()}"; open(keep, "wb").write(d)
                print(f"MISMATCH seed {seed} {fmt} {img.mode} {w}x{h}: ref {None if want is None else want.shape} mine {None if got is None else got.shape} -> {keep}", flush=True)
            else:
                n_trunc_diff += 1
                if os.environ.get("FUZZ_VERBOSE"): print(f"  truncated seed {seed} {fmt} cut {cut}/{len(data)}: ref {None if want is None else 'decodes'} mine {None if got is None else 'decodes'}")
print("files", files, "mismatches on intact files", bad, "| truncated files that differ (informational)", n_trunc_diff)
