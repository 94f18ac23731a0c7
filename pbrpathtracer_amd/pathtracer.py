"""Python mirror of the reference's `PathTracer` class (reference PathTracing/src/pathtracer.h:100-130),
bound to the C++ host layer in libptk.so through the C wrapper of include/ptk_host.h.

Same method names, argument meaning, call order and (silent) error behaviour as the reference, so the
tests read like calls into the reference class:

    pt = PathTracer()
    pt.LoadObject("cornell.obj", model)          # 4x4, column-major like glm
    pt.SetMaterial(0, 0, material_floats)
    pt.BuildBVH(); pt.SetResolution((w, h)); pt.SetTraceDepth(d)
    pt.SetOutImage(rgb8); pt.ResetImage()
    pt.RenderFrame()                              # one sample per pixel, on the GPU

Nothing here computes pixels on the CPU: rendering needs libptk.so and an MI355X.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional, Sequence

import numpy as np

from . import ptk as _ptk

_f = C.POINTER(C.c_float)

HOST_SYMBOLS = [
    "pth_create", "pth_destroy", "pth_load_object", "pth_set_material", "pth_set_texture", "pth_build_bvh",
    "pth_reset_image", "pth_clear_scene", "pth_get_samples", "pth_get_triangle_count", "pth_get_trace_depth",
    "pth_set_trace_depth", "pth_set_out_image", "pth_set_out_gl_buffer", "pth_set_out_device_image", "pth_set_resolution", "pth_get_resolution", "pth_num_objects",
    "pth_num_elements", "pth_name", "pth_set_camera", "pth_set_projection", "pth_set_focal_dist", "pth_set_aperture",
    "pth_render_frame", "pth_exit", "pth_set_seed", "pth_set_tile", "pth_render_frames", "pth_read_accum",
    "pth_last_error", "pth_context", "pth_staged_scene", "pth_load_scene_file", "pth_pts_roundtrip",
    "pth_trs_matrix", "pth_euler_camera", "pth_triangle_init", "pth_image_load", "pth_image_data", "pth_image_tex2d",
    "pth_export_png",
]

_bound = False
_bind_lock = threading.Lock()


def lib() -> C.CDLL:
    L = _ptk.load()
    if _bound:
        return L
    with _bind_lock:                 # (the prototypes are complete before any thread's first call: a pointer returned through
        return _bind_locked(L)       # ctypes' default int would be cut to 32 bits)


def _bind_locked(L) -> C.CDLL:
    global _bound
    if _bound:
        return L
    vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
    L.pth_create.restype = vp; L.pth_create.argtypes = [i32]
    L.pth_destroy.restype = None; L.pth_destroy.argtypes = [vp]
    L.pth_load_object.restype = None; L.pth_load_object.argtypes = [vp, C.c_char_p, _f]
    L.pth_set_material.restype = None; L.pth_set_material.argtypes = [vp, i32, i32, _f]
    L.pth_set_texture.restype = None; L.pth_set_texture.argtypes = [vp, i32, i32, i32, C.c_char_p]
    for n in ("pth_build_bvh", "pth_reset_image", "pth_clear_scene", "pth_render_frame", "pth_exit"):
        getattr(L, n).restype = None; getattr(L, n).argtypes = [vp]
    for n in ("pth_get_samples", "pth_get_triangle_count", "pth_get_trace_depth", "pth_num_objects"):
        getattr(L, n).restype = i32; getattr(L, n).argtypes = [vp]
    L.pth_num_elements.restype = i32; L.pth_num_elements.argtypes = [vp, i32]
    L.pth_name.restype = i32; L.pth_name.argtypes = [vp, i32, i32, C.c_char_p, i32]
    L.pth_set_trace_depth.restype = None; L.pth_set_trace_depth.argtypes = [vp, i32]
    L.pth_set_out_image.restype = None; L.pth_set_out_image.argtypes = [vp, vp]
    L.pth_set_out_gl_buffer.restype = None; L.pth_set_out_gl_buffer.argtypes = [vp, C.c_uint]
    L.pth_set_out_device_image.restype = None; L.pth_set_out_device_image.argtypes = [vp, vp]
    L.pth_set_resolution.restype = None; L.pth_set_resolution.argtypes = [vp, i32, i32]
    L.pth_get_resolution.restype = None; L.pth_get_resolution.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.pth_set_camera.restype = None; L.pth_set_camera.argtypes = [vp, _f, _f, _f]
    L.pth_set_projection.restype = None; L.pth_set_projection.argtypes = [vp, f32, f32]
    L.pth_set_focal_dist.restype = None; L.pth_set_focal_dist.argtypes = [vp, f32]
    L.pth_set_aperture.restype = None; L.pth_set_aperture.argtypes = [vp, f32]
    L.pth_set_seed.restype = None; L.pth_set_seed.argtypes = [vp, C.c_uint64]
    L.pth_set_tile.restype = None; L.pth_set_tile.argtypes = [vp, i32, i32]
    L.pth_render_frames.restype = None; L.pth_render_frames.argtypes = [vp, i32]
    L.pth_read_accum.restype = i32; L.pth_read_accum.argtypes = [vp, vp]
    L.pth_last_error.restype = C.c_char_p; L.pth_last_error.argtypes = [vp]
    L.pth_context.restype = vp; L.pth_context.argtypes = [vp]
    L.pth_staged_scene.restype = C.POINTER(_ptk.SceneDesc); L.pth_staged_scene.argtypes = [vp]
    L.pth_load_scene_file.restype = i32; L.pth_load_scene_file.argtypes = [vp, C.c_char_p]
    L.pth_pts_roundtrip.restype = i32; L.pth_pts_roundtrip.argtypes = [C.c_char_p, C.c_char_p]
    L.pth_trs_matrix.restype = None; L.pth_trs_matrix.argtypes = [_f, _f, _f, _f]
    L.pth_euler_camera.restype = None; L.pth_euler_camera.argtypes = [_f, _f]
    L.pth_triangle_init.restype = None; L.pth_triangle_init.argtypes = [_f, _f]
    L.pth_image_load.restype = i32; L.pth_image_load.argtypes = [C.c_char_p, C.POINTER(i32), C.POINTER(i32)]
    L.pth_image_data.restype = None; L.pth_image_data.argtypes = [vp]
    L.pth_image_tex2d.restype = None; L.pth_image_tex2d.argtypes = [f32, f32, _f]
    L.pth_export_png.restype = i32; L.pth_export_png.argtypes = [C.c_char_p, vp, i32, i32]
    _bound = True
    return L


def _fp(a):
    return a.ctypes.data_as(_f)


def _f3(v):
    return np.ascontiguousarray(v, dtype=np.float32).reshape(3)


class _PinnedBuffer:
    """Owner of one ptk_host_alloc block, exposed to numpy through the array interface."""

    def __init__(self, shape):
        from . import ptk as _ptk
        self._L = _ptk.load()
        n = int(np.prod(shape))
        self._p = self._L.ptk_host_alloc(n)
        if not self._p:
            raise MemoryError("ptk_host_alloc failed (no HIP device?)")
        self.__array_interface__ = {"shape": tuple(shape), "typestr": "|u1", "data": (self._p, False), "version": 3}

    def __del__(self):
        if getattr(self, "_p", None):
            self._L.ptk_host_free(self._p)
            self._p = None


class PathTracer:
    """The reference's PathTracer API (method names kept verbatim) + the marked extensions."""

    def __init__(self, device: int = 0):
        self.L = lib()
        self.h = self.L.pth_create(device)
        if not self.h:
            raise _ptk.PtkError("pth_create failed")
        self._out = None

    def close(self):
        if getattr(self, "h", None):
            self.L.pth_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- reference API -------------------------------------------------------------------------
    def LoadObject(self, file: str, model=None):
        M = np.eye(4, dtype=np.float32) if model is None else np.ascontiguousarray(model, dtype=np.float32)
        self.L.pth_load_object(self.h, file.encode(), _fp(M.reshape(16)))

    def _set_tex(self, slot, objId, elementId, file):
        self.L.pth_set_texture(self.h, objId, elementId, slot, file.encode())

    def SetDiffuseTextureForElement(self, objId, elementId, file): self._set_tex(0, objId, elementId, file)
    def SetNormalTextureForElement(self, objId, elementId, file): self._set_tex(1, objId, elementId, file)
    def SetEmissTextureForElement(self, objId, elementId, file): self._set_tex(2, objId, elementId, file)
    def SetRoughnessTextureForElement(self, objId, elementId, file): self._set_tex(3, objId, elementId, file)
    def SetMetallicTextureForElement(self, objId, elementId, file): self._set_tex(4, objId, elementId, file)
    def SetOpacityTextureForElement(self, objId, elementId, file): self._set_tex(5, objId, elementId, file)

    def SetMaterial(self, objId: int, elementId: int, material):
        """material: 15 floats (type, diffuse rgb, specular rgb, emissive rgb, emissiveIntensity, roughness,
        reflectiveness, translucency, ior) or scenes.MaterialDesc."""
        m = material.as_floats() if hasattr(material, "as_floats") else np.ascontiguousarray(material, dtype=np.float32)
        self.L.pth_set_material(self.h, objId, elementId, _fp(m))

    def BuildBVH(self): self.L.pth_build_bvh(self.h)
    def ResetImage(self): self.L.pth_reset_image(self.h)
    def ClearScene(self): self.L.pth_clear_scene(self.h)
    def GetSamples(self) -> int: return self.L.pth_get_samples(self.h)
    def GetTriangleCount(self) -> int: return self.L.pth_get_triangle_count(self.h)
    def GetTraceDepth(self) -> int: return self.L.pth_get_trace_depth(self.h)
    def SetTraceDepth(self, depth: int): self.L.pth_set_trace_depth(self.h, depth)

    def SetOutImage(self, out: Optional[np.ndarray]):
        """out: caller-owned uint8 array of W*H*3 (rows bottom-up), written by every RenderFrame()."""
        if out is not None:
            assert out.dtype == np.uint8 and out.flags.c_contiguous
        self.L.pth_set_out_image(self.h, out.ctypes.data if out is not None else None)
        self._out = out                 # (after the call: the previous buffer - this may be its last reference - is unbound before it is freed)

    def SetOutGLBuffer(self, gl_buffer: int):
        """Extension: the 8-bit image goes into an OpenGL buffer object of the current context (0 switches back)."""
        self.L.pth_set_out_gl_buffer(self.h, int(gl_buffer))
        if gl_buffer:
            self._out = None

    def SetOutDeviceImage(self, device_ptr):
        """Extension: the 8-bit image goes into W*H*3 bytes of this GPU's memory (an address, e.g. a torch uint8 tensor's
        data_ptr(), kept alive by the caller); None switches back."""
        self.L.pth_set_out_device_image(self.h, C.c_void_p(device_ptr) if device_ptr else None)
        if device_ptr:
            self._out = None

    def AllocOutImage(self) -> np.ndarray:
        """A page-locked W*H*3 uint8 hand-off buffer (ptk_host_alloc) for SetOutImage: RenderFrame()'s copy into
        it is a single DMA transfer.  Freed when the returned array (and its views) are garbage-collected."""
        w, h = self.GetResolution()
        return np.asarray(_PinnedBuffer((h, w, 3)))

    def SetResolution(self, res: Sequence[int]): self.L.pth_set_resolution(self.h, int(res[0]), int(res[1]))

    def GetResolution(self):
        w = C.c_int(); h = C.c_int()
        self.L.pth_get_resolution(self.h, C.byref(w), C.byref(h))
        return w.value, h.value

    def GetNames(self, obj: int):
        """(object name, [element names]) of loaded object `obj` (PathTracerLoader::Object, pathtracer.cpp:49-62)."""
        def one(e):
            buf = C.create_string_buffer(1024)
            n = self.L.pth_name(self.h, obj, e, buf, 1024)
            return None if n < 0 else buf.value.decode("latin-1")
        return one(-1), [one(e) for e in range(self.L.pth_num_elements(self.h, obj))]

    def GetLoadedObjects(self):
        return [self.L.pth_num_elements(self.h, i) for i in range(self.L.pth_num_objects(self.h))]

    def SetCamera(self, pos, dir, up): self.L.pth_set_camera(self.h, _fp(_f3(pos)), _fp(_f3(dir)), _fp(_f3(up)))
    def SetProjection(self, f: float, fovy: float): self.L.pth_set_projection(self.h, f, fovy)
    def SetCameraFocalDist(self, dist: float): self.L.pth_set_focal_dist(self.h, dist)
    def SetCameraAperture(self, aperture: float): self.L.pth_set_aperture(self.h, aperture)
    def RenderFrame(self): self.L.pth_render_frame(self.h)
    def Exit(self): self.L.pth_exit(self.h)

    # ---- extensions ----------------------------------------------------------------------------
    def SetSeed(self, seed: int): self.L.pth_set_seed(self.h, seed)
    def SetTile(self, rank: int, world: int): self.L.pth_set_tile(self.h, rank, world)
    def RenderFrames(self, count: int): self.L.pth_render_frames(self.h, count)

    def ReadAccumulation(self) -> np.ndarray:
        w, h = self.GetResolution()
        out = np.empty((h, w, 3), np.float32)
        if not self.L.pth_read_accum(self.h, out.ctypes.data):
            raise _ptk.PtkError("ReadAccumulation failed: " + self.LastError())
        return out

    def LastError(self) -> str: return self.L.pth_last_error(self.h).decode()

    def LoadSceneFile(self, pts_path: str):
        """.pts -> LoadObject/SetMaterial/Set*Texture/BuildBVH/SetCamera/... (main.cpp:261-438,
        previewer.cpp:770-817, :924-930)."""
        if self.L.pth_load_scene_file(self.h, pts_path.encode()) != 0:
            raise _ptk.PtkError(self.LastError())

    def context(self) -> "_ptk.Context":
        """Borrowed view of this tracer's ptk context (for stats / timing / device pointers)."""
        c = _ptk.Context.__new__(_ptk.Context)
        c.L = self.L
        c.h = C.c_void_p(self.L.pth_context(self.h))
        if not c.h:
            raise _ptk.PtkError("no HIP device: " + self.LastError())
        c.width, c.height = self.GetResolution()
        c._keep = self
        c.close = lambda: None
        return c

    def StagedScene(self) -> dict:
        """The staged scene as numpy arrays in the boundary layout (host only, no GPU needed)."""
        d = self.L.pth_staged_scene(self.h).contents
        n = d.num_triangles

        def arr(ptr, count, dtype):
            if not ptr or count == 0:
                return np.zeros(0, dtype)
            nbytes = count * np.dtype(dtype).itemsize
            buf = (C.c_char * nbytes).from_address(ptr)
            return np.frombuffer(buf, dtype=dtype, count=count).copy()

        return {
            "verts": arr(d.verts, n * 9, np.float32).reshape(n, 9),
            "normals": arr(d.normals, n * 9, np.float32).reshape(n, 9),
            "uvs": arr(d.uvs, n * 6, np.float32).reshape(n, 6),
            "tbn": arr(d.tbn, n * 9, np.float32).reshape(n, 9),
            "smoothing": arr(d.smoothing, n, np.uint8),
            "material": arr(d.material, n, np.int32),
            "materials": arr(d.materials, d.num_materials, _ptk.MATERIAL_DTYPE),
            "textures": arr(d.textures, d.num_textures, _ptk.TEXTURE_DTYPE),
            "texels": arr(d.texels, d.texel_bytes, np.uint8),
            "lights": arr(d.lights, d.num_lights, np.int32),
        }


# ---- host-only helpers (no GPU) ---------------------------------------------------------------------

def trs_matrix(loc, rot_deg, scl) -> np.ndarray:
    out = np.zeros(16, np.float32)
    lib().pth_trs_matrix(_fp(_f3(loc)), _fp(_f3(rot_deg)), _fp(_f3(scl)), _fp(out))
    return out.reshape(4, 4)       # [column][row], glm layout


def euler_camera(rot_deg):
    out = np.zeros(6, np.float32)
    lib().pth_euler_camera(_fp(_f3(rot_deg)), _fp(out))
    return out[:3].copy(), out[3:].copy()


def triangle_init(in15) -> np.ndarray:
    i = np.ascontiguousarray(in15, np.float32); out = np.zeros(9, np.float32)
    lib().pth_triangle_init(_fp(i), _fp(out))
    return out


def image_load(path: str):
    w = C.c_int(); h = C.c_int()
    ok = lib().pth_image_load(path.encode(), C.byref(w), C.byref(h))
    if not ok:
        return None
    data = np.zeros((h.value, w.value, 4), np.uint8)
    lib().pth_image_data(data.ctypes.data)
    return data


def image_tex2d(u: float, v: float) -> np.ndarray:
    out = np.zeros(4, np.float32)
    lib().pth_image_tex2d(u, v, _fp(out))
    return out


def export_png(path: str, rgb8_bottom_up: np.ndarray) -> bool:
    """ExportAt (main.cpp:760-771): the bottom-up RGB8 buffer as a top-down PNG."""
    a = np.ascontiguousarray(rgb8_bottom_up, np.uint8)
    h, w, c = a.shape
    assert c == 3
    return bool(lib().pth_export_png(path.encode(), a.ctypes.data, w, h))


def camera_from_scene(scene):
    """(pos, dir, up, focal, fovy, focal_dist, aperture) of a scenes.SceneDesc, the way
    Previewer::SetPathTracerCamera derives them (previewer.cpp:924-930)."""
    from . import scenes as S
    d, u = euler_camera(scene.cam_rot)
    return dict(pos=np.array(scene.cam_pos, np.float32), dir=d, up=u, focal=S.PTS_FOCAL, fovy=S.PTS_FOVY,
                focal_dist=float(scene.focal_dist), aperture=float(np.float32(S.PTS_FOCAL) / np.float32(scene.camera_f)))
