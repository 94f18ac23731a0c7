"""TEST INFRASTRUCTURE ONLY — ctypes binding to oracle/_ref/libptref.so (the real reference built by
oracle/Makefile.ref).  Only oracle/gen_golden.py and a few opt-in tests use it; it exists only in
the container that has /root/reference and is never imported by the product."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libptref.so")

_f = C.POINTER(C.c_float)


def _fp(a):
    return a.ctypes.data_as(_f)


def available() -> bool:
    return os.path.exists(LIB_PATH)


class Ref:
    def __init__(self):
        self.lib = C.CDLL(LIB_PATH)
        L = self.lib
        L.ref_render_frames.restype = C.c_double
        L.ref_render_frames.argtypes = [C.c_int, C.c_int]
        L.ref_set_projection.argtypes = [C.c_float, C.c_float]
        L.ref_set_focal_dist.argtypes = [C.c_float]
        L.ref_set_aperture.argtypes = [C.c_float]
        L.ref_tex2d.argtypes = [C.c_float, C.c_float, _f]
        L.ref_seed.argtypes = [C.c_uint]

    # --- scene -------------------------------------------------------------------------------
    def load_scene(self, scene, model_of=None, exact_pinhole=False):
        """scene: pbrpathtracer_amd.scenes.SceneDesc.  Follows Previewer::SendObjectsToPathTracer
        (previewer.cpp:770-817) + SetPathTracerCamera (:924-930) through the reference's API."""
        from pbrpathtracer_amd import scenes as S
        L = self.lib
        L.ref_clear()
        for i, o in enumerate(scene.objects):
            M = np.zeros(16, dtype=np.float32)
            L.ref_trs_matrix(_fp(np.array(o.location, dtype=np.float32)),
                             _fp(np.array(o.rotation, dtype=np.float32)),
                             _fp(np.array(o.scale, dtype=np.float32)), _fp(M))
            L.ref_load_obj(o.obj_path.encode(), _fp(M))
            for j, e in enumerate(o.elements):
                m = e.material.as_floats()
                L.ref_set_material(i, j, _fp(m))
                for s, slot in enumerate(S.TEX_SLOTS):
                    if slot in e.material.textures:
                        L.ref_set_texture(i, j, s, e.material.textures[slot].encode())
        L.ref_build()
        du = np.zeros(6, dtype=np.float32)
        L.ref_euler_camera(_fp(np.array(scene.cam_rot, dtype=np.float32)), _fp(du))
        L.ref_set_camera(_fp(np.array(scene.cam_pos, dtype=np.float32)), _fp(du[:3].copy()), _fp(du[3:].copy()))
        L.ref_set_projection(S.PTS_FOCAL, S.PTS_FOVY)
        L.ref_set_focal_dist(scene.focal_dist)
        L.ref_set_aperture(np.float32(S.PTS_FOCAL) / np.float32(scene.camera_f))
        if exact_pinhole and getattr(scene, "pinhole", False):
            L.ref_set_aperture(0.0)                       # SetCameraAperture(0), SURVEY.md §8(d2)
        L.ref_set_depth(scene.trace_depth)
        L.ref_set_resolution(scene.width, scene.height)

    def triangles(self) -> np.ndarray:
        n = self.lib.ref_num_triangles()
        out = np.zeros((n, 38), dtype=np.float32)
        self.lib.ref_get_triangles(_fp(out))
        return out

    def trace(self, ro, rd) -> np.ndarray:
        out = np.zeros(3, dtype=np.float32)
        self.lib.ref_trace(_fp(np.asarray(ro, dtype=np.float32)), _fp(np.asarray(rd, dtype=np.float32)), _fp(out))
        return out

    def hit(self, ro, rd):
        out = np.zeros(3, dtype=np.float32)
        tri = C.c_int(-1)
        h = self.lib.ref_hit(_fp(np.asarray(ro, dtype=np.float32)), _fp(np.asarray(rd, dtype=np.float32)),
                             _fp(out), C.byref(tri))
        return h, tri.value, out

    def peek_tape(self, n: int) -> np.ndarray:
        t = np.zeros(n, dtype=np.float32)
        self.lib.ref_peek_tape(_fp(t), n)
        return t

    def render(self, frames: int, threads: int = 0):
        sec = self.lib.ref_render_frames(frames, threads)
        return sec

    def total(self, w: int, h: int) -> np.ndarray:
        out = np.zeros((h, w, 3), dtype=np.float32)
        self.lib.ref_read_total(_fp(out))
        return out

    def rgb8(self, w: int, h: int) -> np.ndarray:
        out = np.zeros((h, w, 3), dtype=np.uint8)
        self.lib.ref_read_rgb8(out.ctypes.data_as(C.POINTER(C.c_ubyte)))
        return out
