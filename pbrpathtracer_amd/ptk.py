"""ctypes binding of the ptk C-ABI (include/ptk.h) — the MI355X render path.

There is NO CPU fallback here: if libptk.so (HIP kernels for gfx950 + host side) is missing or no
GPU is visible, loading / creating a context raises.  Scenes are dicts of numpy arrays in the
boundary's flat layout (see include/ptk.h `ptk_scene_desc`).
"""
from __future__ import annotations

import ctypes as C
import threading
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# The product loads the in-tree library, nothing else.  Only tools/*.sh (A/B of two builds inside one GPU-box call) point the
# binding at another build, and they have to say so twice: PTK_DEV_TOOLS=1 and PTK_LIB_PATH=<a libptk_*.so beside libptk.so>.
LIB_PATH = os.path.join(_HERE, "libptk.so")
if os.environ.get("PTK_DEV_TOOLS") == "1" and os.environ.get("PTK_LIB_PATH", "").startswith(os.path.join(_HERE, "libptk_")):
    LIB_PATH = os.environ["PTK_LIB_PATH"]

PTK_OK = 0
PTK_TILE = 16
PTK_MAX_BVH_DEPTH = 32

MATERIAL_DTYPE = np.dtype([
    ("type", np.int32), ("diffuse", np.float32, 3), ("specular", np.float32, 3),
    ("emissive", np.float32, 3), ("emissive_intensity", np.float32), ("roughness", np.float32),
    ("reflectiveness", np.float32), ("translucency", np.float32), ("ior", np.float32),
    ("tex", np.int32, 6)], align=False)
TEXTURE_DTYPE = np.dtype([("width", np.int32), ("height", np.int32), ("offset", np.int64)])
assert MATERIAL_DTYPE.itemsize == 84 and TEXTURE_DTYPE.itemsize == 16


class SceneDesc(C.Structure):
    _fields_ = [
        ("num_triangles", C.c_int32), ("verts", C.c_void_p), ("normals", C.c_void_p),
        ("uvs", C.c_void_p), ("tbn", C.c_void_p), ("smoothing", C.c_void_p), ("material", C.c_void_p),
        ("num_materials", C.c_int32), ("materials", C.c_void_p),
        ("num_textures", C.c_int32), ("textures", C.c_void_p), ("texels", C.c_void_p),
        ("texel_bytes", C.c_int64),
        ("num_lights", C.c_int32), ("lights", C.c_void_p),
    ]


class Stats(C.Structure):
    _fields_ = [(k, C.c_uint64) for k in
                ("samples", "rays", "shadow_rays", "node_visits", "tri_tests", "hits_shaded", "tex_fetches",
                 "walk_wave_iters", "walk_lane_iters", "shade_wave_execs", "shade_lanes", "gen_wave_execs", "gen_lanes",
                 "tri_wave_execs", "tri_lanes", "max_walk_nodes", "paths_started")]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


class PtkError(RuntimeError):
    pass


_lib = None

# every symbol include/ptk.h declares
SYMBOLS = [
    "ptk_create", "ptk_destroy", "ptk_upload_scene", "ptk_update_materials", "ptk_set_camera", "ptk_set_frame", "ptk_set_tile",
    "ptk_reset", "ptk_render", "ptk_resolve_rgb8", "ptk_read_accum", "ptk_write_accum", "ptk_samples",
    "ptk_request_exit", "ptk_synchronize", "ptk_last_error", "ptk_accum_device_ptr", "ptk_rgb8_device_ptr",
    "ptk_bind_accum", "ptk_set_stream", "ptk_gather_accum", "ptk_set_option", "ptk_last_render_ms",
    "ptk_last_kernel_ms", "ptk_collect_stats",
    "ptk_bvh_info", "ptk_bvh_layout", "ptk_upload_timing", "ptk_download_bvh", "ptk_probe_hits", "ptk_probe_primary_dirs", "ptk_probe_math", "ptk_probe_direct", "ptk_host_alloc", "ptk_host_free",
    "ptk_packed_floats", "ptk_packed_layout", "ptk_comm_unique_id", "ptk_comm_init", "ptk_comm_destroy",
    "ptk_gather_wait", "ptk_read_gathered", "ptk_gathered_device_ptr", "ptk_probe_pack", "ptk_probe_unpack",
    "ptk_bind_out_image", "ptk_bind_out_device", "ptk_bind_gl_buffer", "ptk_comm_info", "ptk_kernel_log", "ptk_kernel_log_read",
    "ptk_debug_stall_exchange",
]


_load_lock = threading.Lock()


def load() -> C.CDLL:
    """Load libptk.so; raises if it has not been built (python -c 'import __graft_entry__ as g; g.build()')."""
    global _lib
    if _lib is not None:
        return _lib
    with _load_lock:                 # (two threads' first calls: ONE library object gets the prototypes, everybody uses that one)
        return _load_locked()


def _load_locked() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PtkError(f"{LIB_PATH} not found: the HIP extension is not built (run __graft_entry__.build()); "
                       "there is no CPU fallback")
    L = C.CDLL(LIB_PATH)
    vp, i32, u32, u64, f32 = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.c_float
    fp = C.POINTER(C.c_float)
    L.ptk_create.argtypes = [C.POINTER(vp), i32]
    L.ptk_destroy.argtypes = [vp]; L.ptk_destroy.restype = None
    L.ptk_upload_scene.argtypes = [vp, C.POINTER(SceneDesc)]
    L.ptk_update_materials.argtypes = [vp, i32, vp]
    L.ptk_set_camera.argtypes = [vp, fp, fp, fp, f32, f32, f32, f32]
    L.ptk_set_frame.argtypes = [vp, i32, i32, i32]
    L.ptk_set_tile.argtypes = [vp, i32, i32]
    L.ptk_reset.argtypes = [vp]
    L.ptk_render.argtypes = [vp, u32, u32, u64]
    L.ptk_resolve_rgb8.argtypes = [vp, vp]
    L.ptk_read_accum.argtypes = [vp, vp]
    L.ptk_host_alloc.restype = vp; L.ptk_host_alloc.argtypes = [C.c_size_t]
    L.ptk_host_free.restype = None; L.ptk_host_free.argtypes = [vp]
    L.ptk_write_accum.argtypes = [vp, vp, i32]
    L.ptk_samples.argtypes = [vp]
    L.ptk_request_exit.argtypes = [vp]
    L.ptk_synchronize.argtypes = [vp]
    L.ptk_last_error.argtypes = [vp]; L.ptk_last_error.restype = C.c_char_p
    L.ptk_accum_device_ptr.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.ptk_rgb8_device_ptr.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.ptk_bind_accum.argtypes = [vp, vp]
    L.ptk_set_stream.argtypes = [vp, vp]
    L.ptk_gather_accum.argtypes = [vp, vp, i32]
    L.ptk_packed_floats.argtypes = [i32, i32, i32, i32]; L.ptk_packed_floats.restype = C.c_int64
    L.ptk_packed_layout.argtypes = [i32, i32, i32, i32, vp]
    L.ptk_comm_unique_id.argtypes = [vp]
    L.ptk_comm_init.argtypes = [vp, vp, i32, i32]
    L.ptk_comm_destroy.argtypes = [vp]
    L.ptk_comm_info.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.ptk_bind_out_image.argtypes = [vp, vp]
    L.ptk_bind_out_device.argtypes = [vp, vp]
    L.ptk_bind_gl_buffer.argtypes = [vp, C.c_uint]
    L.ptk_gather_wait.argtypes = [vp]
    L.ptk_read_gathered.argtypes = [vp, vp]
    L.ptk_gathered_device_ptr.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]
    L.ptk_probe_pack.argtypes = [vp, i32, i32, vp]
    L.ptk_probe_unpack.argtypes = [vp, i32, vp, vp]
    L.ptk_last_render_ms.argtypes = [vp, fp, C.POINTER(i32)]
    L.ptk_last_kernel_ms.argtypes = [vp, fp, fp]
    L.ptk_set_option.argtypes = [vp, C.c_char_p, C.c_double]
    L.ptk_collect_stats.argtypes = [vp, u32, u32, u64, C.POINTER(Stats)]
    try:
        L.ptk_kernel_log.argtypes = [vp, i32]
        L.ptk_kernel_log_read.argtypes = [vp, C.POINTER(C.c_float), i32, C.POINTER(C.c_int)]
        L.ptk_debug_stall_exchange.argtypes = [vp, i32]
    except AttributeError:
        if LIB_PATH.endswith("libptk.so"):      # (an older build loaded through PTK_DEV_TOOLS for an A/B may lack the newest entry points)
            raise
    L.ptk_bvh_info.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.ptk_bvh_layout.argtypes = [vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.ptk_download_bvh.argtypes = [vp, vp, vp]
    L.ptk_upload_timing.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    L.ptk_probe_hits.argtypes = [vp, i32, vp, vp, vp, vp]
    L.ptk_probe_math.argtypes = [vp, i32, i32, vp, vp]
    L.ptk_probe_direct.argtypes = [vp, i32, vp, vp, vp, vp, vp]
    L.ptk_probe_primary_dirs.argtypes = [vp, vp]
    _lib = L
    return L


def normalise_arrays(a: dict) -> dict:
    n = len(a["verts"])
    return {
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


def scene_desc(a: dict) -> SceneDesc:
    d = SceneDesc()
    d.num_triangles = len(a["verts"])
    for k in ("verts", "normals", "uvs", "tbn", "smoothing", "material", "materials", "textures", "texels", "lights"):
        setattr(d, k, a[k].ctypes.data if a[k].size else None)
    d.num_materials = len(a["materials"])
    d.num_textures = len(a["textures"])
    d.texel_bytes = a["texels"].size
    d.num_lights = len(a["lights"])
    return d


class Context:
    """One GPU, one HIP stream (ptk_ctx)."""

    def __init__(self, device: int = 0):
        self.L = load()
        h = C.c_void_p()
        rc = self.L.ptk_create(C.byref(h), device)
        if rc != PTK_OK:
            raise PtkError(f"ptk_create(device={device}) failed with {rc}: no usable MI355X / HIP device")
        self.h = h
        self.width = self.height = 0
        self._keep = None

    def _chk(self, rc: int, what: str):
        if rc != PTK_OK:
            raise PtkError(f"{what} failed ({rc}): {self.L.ptk_last_error(self.h).decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.L.ptk_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- scene / camera / frame ------------------------------------------------------------
    def upload_scene(self, arrays: dict):
        a = normalise_arrays(arrays)
        d = scene_desc(a)
        self._chk(self.L.ptk_upload_scene(self.h, C.byref(d)), "ptk_upload_scene")

    def update_materials(self, materials: np.ndarray):
        m = np.ascontiguousarray(materials, dtype=MATERIAL_DTYPE)
        self._chk(self.L.ptk_update_materials(self.h, len(m), m.ctypes.data), "ptk_update_materials")

    def set_camera(self, pos, dir, up, focal, fovy, focal_dist, aperture):
        f3 = C.c_float * 3
        self._chk(self.L.ptk_set_camera(self.h, f3(*map(float, pos)), f3(*map(float, dir)), f3(*map(float, up)),
                                        float(focal), float(fovy), float(focal_dist), float(aperture)), "ptk_set_camera")

    def set_frame(self, width: int, height: int, max_depth: int):
        self._chk(self.L.ptk_set_frame(self.h, width, height, max_depth), "ptk_set_frame")
        self.width, self.height = width, height

    def set_tile(self, rank: int, world: int):
        self._chk(self.L.ptk_set_tile(self.h, rank, world), "ptk_set_tile")

    def reset(self):
        self._chk(self.L.ptk_reset(self.h), "ptk_reset")

    # ---- render ----------------------------------------------------------------------------
    def render(self, first_sample: int, spp: int, seed: int):
        self._chk(self.L.ptk_render(self.h, first_sample, spp, seed), "ptk_render")

    def synchronize(self):
        self._chk(self.L.ptk_synchronize(self.h), "ptk_synchronize")

    def read_accum(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 3), dtype=np.float32)
        self._chk(self.L.ptk_read_accum(self.h, out.ctypes.data), "ptk_read_accum")
        return out

    def write_accum(self, total: np.ndarray, samples: int):
        t = np.ascontiguousarray(total, dtype=np.float32)
        assert t.shape == (self.height, self.width, 3)
        self._chk(self.L.ptk_write_accum(self.h, t.ctypes.data, samples), "ptk_write_accum")

    def resolve_rgb8(self, out: Optional[np.ndarray] = None) -> np.ndarray:
        if out is None:
            out = np.empty((self.height, self.width, 3), dtype=np.uint8)
        self._chk(self.L.ptk_resolve_rgb8(self.h, out.ctypes.data), "ptk_resolve_rgb8")
        return out

    def samples(self) -> int:
        return self.L.ptk_samples(self.h)

    def request_exit(self):
        self._chk(self.L.ptk_request_exit(self.h), "ptk_request_exit")

    def last_render_ms(self):
        ms = C.c_float(0); n = C.c_int(0)
        self._chk(self.L.ptk_last_render_ms(self.h, C.byref(ms), C.byref(n)), "ptk_last_render_ms")
        return ms.value, n.value

    def last_kernel_ms(self):
        t = C.c_float(0); a = C.c_float(0)
        self._chk(self.L.ptk_last_kernel_ms(self.h, C.byref(t), C.byref(a)), "ptk_last_kernel_ms")
        return t.value, a.value

    def kernel_log(self, capacity: int):
        """Start (capacity > 0) or stop (0) the log of every trace launch's duration."""
        self._chk(self.L.ptk_kernel_log(self.h, int(capacity)), "ptk_kernel_log")
        self._klog_cap = int(capacity)

    def kernel_log_read(self):
        """Durations (ms) of the trace launches since the log was started / last read, in launch order."""
        cap = max(1, getattr(self, "_klog_cap", 0))
        buf = (C.c_float * cap)(); n = C.c_int(0)
        self._chk(self.L.ptk_kernel_log_read(self.h, buf, cap, C.byref(n)), "ptk_kernel_log_read")
        return [float(buf[i]) for i in range(n.value)]

    def set_option(self, name: str, value: float):
        self._chk(self.L.ptk_set_option(self.h, name.encode(), float(value)), "ptk_set_option")

    def collect_stats(self, first_sample: int, spp: int, seed: int) -> dict:
        s = Stats()
        self._chk(self.L.ptk_collect_stats(self.h, first_sample, spp, seed, C.byref(s)), "ptk_collect_stats")
        return s.as_dict()

    def bvh_info(self):
        n = C.c_int32(); d = C.c_int32(); t = C.c_int32()
        self._chk(self.L.ptk_bvh_info(self.h, C.byref(n), C.byref(d), C.byref(t)), "ptk_bvh_info")
        return n.value, d.value, t.value

    def bvh_layout(self):
        w = C.c_int32(); b = C.c_int32(); s = C.c_int32()
        self._chk(self.L.ptk_bvh_layout(self.h, C.byref(w), C.byref(b), C.byref(s)), "ptk_bvh_layout")
        return w.value, b.value, s.value

    def download_bvh(self):
        n_nodes, _, n_tris = self.bvh_info()
        nodes = np.zeros((n_nodes, 16), np.float32); order = np.zeros(n_tris, np.int32)
        self._chk(self.L.ptk_download_bvh(self.h, nodes.ctypes.data, order.ctypes.data), "ptk_download_bvh")
        return nodes, order

    def upload_timing(self):
        t = (C.c_double * 4)(); dev = C.c_int(0)
        self._chk(self.L.ptk_upload_timing(self.h, t, C.byref(dev)), "ptk_upload_timing")
        d = dict(zip(("bvh_ms", "pack_ms", "copy_ms", "total_ms"), [round(x, 2) for x in t]))
        d["built_on_device"] = bool(dev.value)
        return d

    def node_width(self) -> int:
        return self.bvh_layout()[0]

    def accum_device_ptr(self):
        p = C.c_void_p(); b = C.c_size_t()
        self._chk(self.L.ptk_accum_device_ptr(self.h, C.byref(p), C.byref(b)), "ptk_accum_device_ptr")
        return p.value, b.value

    def bind_accum(self, dev_ptr: int):
        self._chk(self.L.ptk_bind_accum(self.h, dev_ptr), "ptk_bind_accum")

    def set_stream(self, stream_handle: int):
        self._chk(self.L.ptk_set_stream(self.h, stream_handle), "ptk_set_stream")

    # ---- multi-GPU exchange step -------------------------------------------------------------
    def comm_init(self, unique_id: bytes, rank: int, world: int):
        assert len(unique_id) == 128
        self._chk(self.L.ptk_comm_init(self.h, C.c_char_p(unique_id), rank, world), "ptk_comm_init")

    def comm_destroy(self):
        self._chk(self.L.ptk_comm_destroy(self.h), "ptk_comm_destroy")

    def comm_info(self) -> dict:
        """rank / world / device as the library's own RCCL communicator reports them, and the context's HIP ordinal."""
        r, w, d, cd = C.c_int32(-1), C.c_int32(0), C.c_int32(-1), C.c_int32(-1)
        self._chk(self.L.ptk_comm_info(self.h, C.byref(r), C.byref(w), C.byref(d), C.byref(cd)), "ptk_comm_info")
        return {"rank": r.value, "world": w.value, "comm_device": d.value, "ctx_device": cd.value}

    def bind_out_image(self, out):
        """ptk_bind_out_image: `out` = a C-contiguous uint8 [H, W, 3] array (kept alive by the caller) or None."""
        if out is None:
            self._chk(self.L.ptk_bind_out_image(self.h, None), "ptk_bind_out_image")
            return
        assert out.dtype == np.uint8 and out.flags["C_CONTIGUOUS"] and out.size == self.width * self.height * 3
        self._chk(self.L.ptk_bind_out_image(self.h, out.ctypes.data_as(C.c_void_p)), "ptk_bind_out_image")

    def bind_out_device(self, device_ptr):
        """ptk_bind_out_device: the address of W*H*3 bytes of this GPU's memory (e.g. a torch uint8 tensor's data_ptr()) or None."""
        self._chk(self.L.ptk_bind_out_device(self.h, C.c_void_p(device_ptr) if device_ptr else None), "ptk_bind_out_device")

    def bind_gl_buffer(self, gl_buffer: int):
        """ptk_bind_gl_buffer: an OpenGL buffer object of the calling thread's current context (0 unbinds)."""
        self._chk(self.L.ptk_bind_gl_buffer(self.h, int(gl_buffer)), "ptk_bind_gl_buffer")

    def gather_accum(self, root: int = 0, comm=None):
        self._chk(self.L.ptk_gather_accum(self.h, comm, root), "ptk_gather_accum")

    def debug_stall_exchange(self, milliseconds: int):
        self._chk(self.L.ptk_debug_stall_exchange(self.h, int(milliseconds)), "ptk_debug_stall_exchange")

    def gather_wait(self):
        self._chk(self.L.ptk_gather_wait(self.h), "ptk_gather_wait")

    def read_gathered(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 3), dtype=np.float32)
        self._chk(self.L.ptk_read_gathered(self.h, out.ctypes.data), "ptk_read_gathered")
        return out

    def probe_pack(self, rank: int, world: int) -> np.ndarray:
        out = np.zeros(packed_floats(self.width, self.height, rank, world), np.float32)
        self._chk(self.L.ptk_probe_pack(self.h, rank, world, out.ctypes.data), "ptk_probe_pack")
        return out

    def probe_unpack(self, world: int, packed_all: np.ndarray) -> np.ndarray:
        p = np.ascontiguousarray(packed_all, np.float32)
        out = np.empty((self.height, self.width, 3), np.float32)
        self._chk(self.L.ptk_probe_unpack(self.h, world, p.ctypes.data, out.ctypes.data), "ptk_probe_unpack")
        return out

    # ---- probes ----------------------------------------------------------------------------
    def probe_hits(self, ro: np.ndarray, rd: np.ndarray):
        ro = np.ascontiguousarray(ro, np.float32); rd = np.ascontiguousarray(rd, np.float32)
        n = len(ro)
        tri = np.empty(n, np.int32); tuv = np.empty((n, 3), np.float32)
        self._chk(self.L.ptk_probe_hits(self.h, n, ro.ctypes.data, rd.ctypes.data, tri.ctypes.data, tuv.ctypes.data), "ptk_probe_hits")
        return tri, tuv

    def probe_direct(self, p, n, diffuse, tape) -> np.ndarray:
        """DirectIllumimation at surface points p with normals n, its three draws per point on `tape` ([N, 3] each)."""
        a = [np.ascontiguousarray(x, np.float32).reshape(-1, 3) for x in (p, n, diffuse, tape)]
        out = np.empty_like(a[0])
        self._chk(self.L.ptk_probe_direct(self.h, len(a[0]), a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data, out.ctypes.data),
                  "ptk_probe_direct")
        return out

    def probe_math(self, op: int, x: np.ndarray) -> np.ndarray:
        """The kernels' exact-arithmetic helpers on an array (op 0 rcp, 1 rcp with special cases, 2 sqrt, 3 1/sqrt)."""
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty_like(x)
        self._chk(self.L.ptk_probe_math(self.h, op, x.size, x.ctypes.data, out.ctypes.data), "ptk_probe_math")
        return out

    def primary_dirs(self) -> np.ndarray:
        out = np.empty((self.height, self.width, 3), np.float32)
        self._chk(self.L.ptk_probe_primary_dirs(self.h, out.ctypes.data), "ptk_probe_primary_dirs")
        return out


# ---- packed exchange layout (host-only functions of the library: no GPU needed) ---------------------------------
def packed_floats(width: int, height: int, rank: int, world: int) -> int:
    n = load().ptk_packed_floats(width, height, rank, world)
    if n < 0:
        raise PtkError("ptk_packed_floats: bad arguments")
    return int(n)


def packed_layout(width: int, height: int, rank: int, world: int) -> np.ndarray:
    """For every float of rank `rank`'s packed buffer: its index in the accumulator (W*H*3, rows bottom-up), -1 = padding."""
    idx = np.empty(packed_floats(width, height, rank, world), np.int64)
    if load().ptk_packed_layout(width, height, rank, world, idx.ctypes.data) != PTK_OK:
        raise PtkError("ptk_packed_layout: bad arguments")
    return idx


def comm_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    if load().ptk_comm_unique_id(buf) != PTK_OK:
        raise PtkError("ptk_comm_unique_id (ncclGetUniqueId) failed")
    return buf.raw
