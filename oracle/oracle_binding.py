"""TEST INFRASTRUCTURE ONLY — ctypes binding to oracle/libptoracle.so (the plain-C CPU restatement,
oracle/pt_oracle.c).  Imported only by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.

Scenes are passed as a dict of numpy arrays in the boundary's flat layout (include/ptk.h):
  verts[N,9] normals[N,9] uvs[N,6] tbn[N,9] float32; smoothing[N] uint8; material[N] int32;
  materials[M] (MATERIAL_DTYPE); textures[T] (TEXTURE_DTYPE); texels uint8[...]; lights[L] int32
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libptoracle.so")

MATERIAL_DTYPE = np.dtype([
    ("type", np.int32), ("diffuse", np.float32, 3), ("specular", np.float32, 3),
    ("emissive", np.float32, 3), ("emissive_intensity", np.float32), ("roughness", np.float32),
    ("reflectiveness", np.float32), ("translucency", np.float32), ("ior", np.float32),
    ("tex", np.int32, 6)], align=False)
assert MATERIAL_DTYPE.itemsize == 84
TEXTURE_DTYPE = np.dtype([("width", np.int32), ("height", np.int32), ("offset", np.int64)])
assert TEXTURE_DTYPE.itemsize == 16

_f = C.POINTER(C.c_float)


class SceneDescC(C.Structure):
    _fields_ = [
        ("num_triangles", C.c_int32), ("verts", C.c_void_p), ("normals", C.c_void_p),
        ("uvs", C.c_void_p), ("tbn", C.c_void_p), ("smoothing", C.c_void_p), ("material", C.c_void_p),
        ("num_materials", C.c_int32), ("materials", C.c_void_p),
        ("num_textures", C.c_int32), ("textures", C.c_void_p), ("texels", C.c_void_p),
        ("texel_bytes", C.c_int64),
        ("num_lights", C.c_int32), ("lights", C.c_void_p),
    ]


class CameraC(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("dir", C.c_float * 3), ("up", C.c_float * 3),
                ("focal", C.c_float), ("fovy", C.c_float), ("focal_dist", C.c_float),
                ("aperture", C.c_float)]


def glm_normalize(v) -> np.ndarray:
    """glm::normalize in float32, operation by operation: x * (1 / sqrt(x.x + y.y + z.z)) - what PathTracer::SetCamera does to
    dir and up (pathtracer.cpp:336-337) and ptk_set_camera repeats."""
    x = np.asarray(v, np.float32)
    sqr = np.float32(np.float32(np.float32(x[0] * x[0]) + np.float32(x[1] * x[1])) + np.float32(x[2] * x[2]))
    inv = np.float32(np.float32(1.0) / np.float32(np.sqrt(sqr)))
    return np.array([np.float32(x[0] * inv), np.float32(x[1] * inv), np.float32(x[2] * inv)], np.float32)


def make_camera(pos, dir, up, focal, fovy, focal_dist, aperture, normalise: bool = False) -> CameraC:
    """The camera as it stands AFTER the reference's setters.  `dir` and `up` are taken as given - fixtures store what
    SetCamera left - unless normalise=True: then they are put through SetCamera's own normalisation (pathtracer.cpp:336-337),
    for callers that hand the same RAW vectors to ptk_set_camera, which normalises them the same way (normalising twice is not
    always a no-op in float32: tools/soak_lens_cull.py found cameras whose every primary direction differed in the last bit)."""
    c = CameraC()
    if normalise:
        dir, up = glm_normalize(dir), glm_normalize(up)
    c.pos[:] = [float(x) for x in pos]
    c.dir[:] = [float(x) for x in dir]
    c.up[:] = [float(x) for x in up]
    c.focal, c.fovy, c.focal_dist, c.aperture = float(focal), float(fovy), float(focal_dist), float(aperture)
    return c


def normalise_arrays(a: dict) -> dict:
    """Contiguous arrays of the exact dtypes the C side expects (keeps references alive)."""
    n = len(a["verts"])
    out = {
        "verts": np.ascontiguousarray(a["verts"], dtype=np.float32).reshape(n, 9),
        "normals": np.ascontiguousarray(a["normals"], dtype=np.float32).reshape(n, 9),
        "uvs": np.ascontiguousarray(a["uvs"], dtype=np.float32).reshape(n, 6),
        "tbn": np.ascontiguousarray(a["tbn"], dtype=np.float32).reshape(n, 9),
        "smoothing": np.ascontiguousarray(a["smoothing"], dtype=np.uint8).reshape(n),
        "material": np.ascontiguousarray(a["material"], dtype=np.int32).reshape(n),
        "materials": np.ascontiguousarray(a["materials"], dtype=MATERIAL_DTYPE),
        "textures": np.ascontiguousarray(a.get("textures", np.zeros(0, TEXTURE_DTYPE)), dtype=TEXTURE_DTYPE),
        "texels": np.ascontiguousarray(a.get("texels", np.zeros(0, np.uint8)), dtype=np.uint8),
        "lights": np.ascontiguousarray(a["lights"], dtype=np.int32),
    }
    return out


def fill_desc(desc, a: dict):
    """Fill a SceneDescC-shaped ctypes struct from normalised arrays."""
    desc.num_triangles = len(a["verts"])
    for k in ("verts", "normals", "uvs", "tbn", "smoothing", "material", "materials", "textures", "texels", "lights"):
        setattr(desc, k, a[k].ctypes.data if a[k].size else None)
    desc.num_materials = len(a["materials"])
    desc.num_textures = len(a["textures"])
    desc.texel_bytes = a["texels"].size
    desc.num_lights = len(a["lights"])
    return desc


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "pt_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return LIB_PATH


class Oracle:
    def __init__(self, arrays: dict):
        build()
        self.lib = L = C.CDLL(LIB_PATH)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.POINTER(SceneDescC)]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_render.argtypes = [C.c_void_p, C.POINTER(CameraC), C.c_int, C.c_int, C.c_int, C.c_uint32,
                                 C.c_uint32, C.c_uint64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_primary_dirs.argtypes = [C.POINTER(CameraC), C.c_int, C.c_int, C.c_void_p]
        L.orc_trace_tape.argtypes = [C.c_void_p, _f, _f, C.c_int, _f, C.c_int, C.c_int, _f]
        L.orc_trace_counter.argtypes = [C.c_void_p, _f, _f, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, _f]
        L.orc_intersect_triangle.argtypes = [_f] * 6
        L.orc_hit.argtypes = [C.c_void_p, _f, _f, _f, C.POINTER(C.c_int32)]
        L.orc_hit_brute.argtypes = [C.c_void_p, _f, _f, _f, C.POINTER(C.c_int32)]
        L.orc_tex2d.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, _f]
        L.orc_triangle_init.argtypes = [_f, _f]
        L.orc_sincos.argtypes = [C.c_float, _f, _f]
        L.orc_rand_u01.restype = C.c_float
        L.orc_rand_u01.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]
        L.orc_sample_circle.argtypes = [C.c_float, C.c_float, _f]
        L.orc_bvh_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.orc_direct_illumination_tape.argtypes = [C.c_void_p, _f, _f, _f, _f, _f]
        self.arrays = normalise_arrays(arrays)
        self.desc = fill_desc(SceneDescC(), self.arrays)
        self.h = L.orc_create(C.byref(self.desc))

    def close(self):
        if self.h:
            self.lib.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(_f)

    def render(self, cam: CameraC, width, height, depth, first_sample, spp, seed, total=None,
               rank=0, world=1, threads=0, want_rgb8=True):
        if total is None:
            total = np.zeros((height, width, 3), dtype=np.float32)
        assert total.dtype == np.float32 and total.flags.c_contiguous
        rgb = np.zeros((height, width, 3), dtype=np.uint8) if want_rgb8 else None
        self.lib.orc_render(self.h, C.byref(cam), width, height, depth, first_sample, spp, seed, rank, world,
                            total.ctypes.data, rgb.ctypes.data if rgb is not None else None, threads)
        return total, rgb

    def render_tape(self, cam: CameraC, width, height, depth, tape):
        """One RenderFrame with the reference's draws on tape (its single-thread pixel order); returns (total, draws consumed)."""
        total = np.zeros((height, width, 3), dtype=np.float32)
        tape = np.ascontiguousarray(tape, np.float32)
        self.lib.orc_render_tape.restype = C.c_int
        self.lib.orc_render_tape.argtypes = [C.c_void_p, C.POINTER(CameraC), C.c_int, C.c_int, C.c_int, _f, C.c_int, _f]
        n = self.lib.orc_render_tape(self.h, C.byref(cam), width, height, depth, self._p(tape), len(tape), self._p(total))
        return total, n

    def primary_dirs(self, cam: CameraC, width, height):
        out = np.zeros((height, width, 3), dtype=np.float32)
        self.lib.orc_primary_dirs(C.byref(cam), width, height, out.ctypes.data)
        return out

    def trace_tape(self, ro, rd, depth, tape, mode=2):
        ro = np.asarray(ro, np.float32); rd = np.asarray(rd, np.float32)
        tape = np.ascontiguousarray(tape, np.float32)
        out = np.zeros(3, np.float32)
        n = self.lib.orc_trace_tape(self.h, self._p(ro), self._p(rd), depth, self._p(tape), len(tape), mode, self._p(out))
        return out, n

    def trace_counter(self, ro, rd, depth, seed, pixel, sample, mode=0):
        ro = np.asarray(ro, np.float32); rd = np.asarray(rd, np.float32)
        out = np.zeros(3, np.float32)
        self.lib.orc_trace_counter(self.h, self._p(ro), self._p(rd), depth, seed, pixel, sample, mode, self._p(out))
        return out

    def hit(self, ro, rd, brute=False):
        ro = np.asarray(ro, np.float32); rd = np.asarray(rd, np.float32)
        tuv = np.zeros(3, np.float32); tri = C.c_int32(-1)
        fn = self.lib.orc_hit_brute if brute else self.lib.orc_hit
        h = fn(self.h, self._p(ro), self._p(rd), self._p(tuv), C.byref(tri))
        return h, tri.value, tuv

    def tex2d(self, tex, u, v):
        out = np.zeros(4, np.float32)
        self.lib.orc_tex2d(self.h, tex, u, v, self._p(out))
        return out

    def direct_illumination_tape(self, p, n, diffuse, tape3) -> np.ndarray:
        out = np.zeros(3, np.float32)
        a = [np.ascontiguousarray(x, np.float32) for x in (p, n, diffuse, tape3)]
        self.lib.orc_direct_illumination_tape(self.h, *[self._p(x) for x in a], self._p(out))
        return out

    def bvh_info(self):
        n = C.c_int32(); d = C.c_int32()
        self.lib.orc_bvh_info(self.h, C.byref(n), C.byref(d))
        return n.value, d.value


def lib():
    build()
    L = C.CDLL(LIB_PATH)
    L.orc_intersect_triangle.argtypes = [_f] * 6
    L.orc_triangle_init.argtypes = [_f, _f]
    L.orc_sincos.argtypes = [C.c_float, _f, _f]
    L.orc_rand_u01.restype = C.c_float
    L.orc_rand_u01.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int]
    L.orc_sample_circle.argtypes = [C.c_float, C.c_float, _f]
    L.orc_primary_dirs.argtypes = [C.POINTER(CameraC), C.c_int, C.c_int, C.c_void_p]
    L.orc_aabb_intersect.argtypes = [_f] * 4
    L.orc_aabb_build.argtypes = [_f, C.c_int, _f]
    return L
