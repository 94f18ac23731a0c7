"""Procedural benchmark / test scenes, written through the reference's own on-disk formats.

The reference ships no assets, so every scene named in BASELINE.json.configs is
synthesised here with fixed seeds and written as Wavefront ``.obj`` meshes + a ``.pts`` scene file
following the *reader's* grammar (reference ``PathTracing/src/main.cpp:261-438``).
Textures are written as binary PPM (``P6``), which the reference's stb_image 2.27 decodes, so the
same files feed the real reference (oracle/_ref) and this project.

This module is tooling (bench + tests); the render path itself is the C++/HIP library.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

OPAQUE = 0
TRANSLUCENT = 1
TEX_SLOTS = ("diffuse", "normal", "emissive", "roughness", "metallic", "opacity")

# fixed camera constants of the .pts path (reference previewer.cpp:13-14)
PTS_FOCAL = 0.05
PTS_FOVY = 70.0


@dataclass
class MaterialDesc:
    """Mirror of the reference's ``Material`` (mesh.h:21-59) with its defaults."""
    type: int = OPAQUE
    diffuse: Tuple[float, float, float] = (1.0, 1.0, 1.0)
    specular: Tuple[float, float, float] = (1.0, 1.0, 1.0)
    emissive: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    emissive_intensity: float = 1.0
    roughness: float = 1.0
    reflectiveness: float = 0.0
    translucency: float = 1.0
    ior: float = 1.5
    textures: Dict[str, str] = field(default_factory=dict)  # slot name -> file path

    def as_floats(self) -> np.ndarray:
        return np.array([float(self.type), *self.diffuse, *self.specular, *self.emissive,
                         self.emissive_intensity, self.roughness, self.reflectiveness,
                         self.translucency, self.ior], dtype=np.float32)


@dataclass
class ElementDesc:
    name: str
    material: MaterialDesc


@dataclass
class ObjectDesc:
    obj_path: str
    name: str
    elements: List[ElementDesc]
    location: Tuple[float, float, float] = (0.0, 0.0, 0.0)
    rotation: Tuple[float, float, float] = (0.0, 0.0, 0.0)  # degrees, X then Y then Z
    scale: Tuple[float, float, float] = (1.0, 1.0, 1.0)


@dataclass
class SceneDesc:
    """Everything a ``.pts`` file holds (main.cpp:261-438)."""
    trace_depth: int = 4
    width: int = 512
    height: int = 512
    auto_res: int = 0
    cam_pos: Tuple[float, float, float] = (0.0, 0.0, -3.5)
    cam_rot: Tuple[float, float, float] = (0.0, 0.0, 0.0)  # degrees
    focal_dist: float = 3.5
    camera_f: float = 1.0e9  # f-number; aperture radius = 0.05 / F  (previewer.cpp:929)
    # pinhole configs: the .pts carries a huge F (aperture 5e-11); drivers that want an exact pinhole call
    # SetCameraAperture(0) after loading, as SURVEY.md §8(d2) prescribes for C1/C2
    pinhole: bool = True
    objects: List[ObjectDesc] = field(default_factory=list)


# --------------------------------------------------------------------------------------------
# mesh writers


@dataclass
class MeshGroup:
    """One OBJ ``o`` group = one tinyobj shape = one reference *element*."""
    name: str
    positions: np.ndarray          # [V,3] float
    faces: np.ndarray              # [F,3] int (0-based into positions)
    uvs: Optional[np.ndarray] = None       # [V,2]
    normals: Optional[np.ndarray] = None   # [V,3]
    smooth: bool = False           # emit `s 1` (-> Triangle::smoothing, pathtracer.cpp:131-135)


def write_obj(path: str, groups: Sequence[MeshGroup]) -> None:
    """Write groups as a Wavefront OBJ with global (1-based) indices.

    If any group carries normals/uvs every group must (the reference keys on
    ``attrib.normals.size() != 0`` for the whole file, pathtracer.cpp:78/85)."""
    any_uv = any(g.uvs is not None for g in groups)
    any_n = any(g.normals is not None for g in groups)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "w") as f:
        f.write("# synthetic mesh (pbrpathtracer_amd.scenes)\n")
        voff = 1
        for g in groups:
            nv = len(g.positions)
            uvs = g.uvs if g.uvs is not None else (np.zeros((nv, 2)) if any_uv else None)
            nrm = g.normals if g.normals is not None else (np.tile([0.0, 1.0, 0.0], (nv, 1)) if any_n else None)
            f.write(f"o {g.name}\n")
            np.savetxt(f, np.asarray(g.positions, dtype=np.float64), fmt="v %.9g %.9g %.9g")
            if uvs is not None:
                np.savetxt(f, np.asarray(uvs, dtype=np.float64), fmt="vt %.9g %.9g")
            if nrm is not None:
                np.savetxt(f, np.asarray(nrm, dtype=np.float64), fmt="vn %.9g %.9g %.9g")
            f.write("s 1\n" if g.smooth else "s off\n")
            idx = np.asarray(g.faces, dtype=np.int64) + voff
            if uvs is not None and nrm is not None:
                cols = np.repeat(idx, 3, axis=1)
                np.savetxt(f, cols, fmt="f %d/%d/%d %d/%d/%d %d/%d/%d")
            elif uvs is not None:
                cols = np.repeat(idx, 2, axis=1)
                np.savetxt(f, cols, fmt="f %d/%d %d/%d %d/%d")
            elif nrm is not None:
                cols = np.repeat(idx, 2, axis=1)
                np.savetxt(f, cols, fmt="f %d//%d %d//%d %d//%d")
            else:
                np.savetxt(f, idx, fmt="f %d %d %d")
            voff += nv


def write_ppm(path: str, rgb: np.ndarray) -> None:
    """Binary PPM (P6), 8-bit RGB, rows top-down."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, c = rgb.shape
    assert c == 3
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (w, h))
        f.write(rgb.tobytes())


def _fmt(x: float) -> str:
    return repr(float(np.float32(x))) if abs(x) < 1e7 else "%.9g" % x


def write_pts(path: str, scene: SceneDesc) -> None:
    """Write a scene in the grammar the reference's *reader* parses (main.cpp:261-438).

    (The reference's own writer ``SaveAt`` is stale and does not round-trip — SURVEY.md §5.1.)"""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    L: List[str] = ["Path Tracer Scene File", "Version=2.1.0"]
    L.append(str(int(scene.trace_depth)))
    L.append(f"{int(scene.width)} {int(scene.height)}")
    L.append(str(int(scene.auto_res)))
    L.append(" ".join(_fmt(v) for v in scene.cam_pos))
    L.append(" ".join(_fmt(v) for v in scene.cam_rot))
    L.append(_fmt(scene.focal_dist))
    L.append(_fmt(scene.camera_f))
    L.append(str(len(scene.objects)))
    for o in scene.objects:
        L.append(o.obj_path)
        L.append(o.name)
        L.append(" ".join(_fmt(v) for v in o.location))
        L.append(" ".join(_fmt(v) for v in o.rotation))
        L.append(" ".join(_fmt(v) for v in o.scale))
        L.append(str(len(o.elements)))
        for e in o.elements:
            m = e.material
            L.append(e.name)
            L.append(" ".join(_fmt(v) for v in m.diffuse))
            L.append(" ".join(_fmt(v) for v in m.specular))
            L.append(" ".join(_fmt(v) for v in m.emissive))
            L.append(_fmt(m.emissive_intensity))
            L.append(f"{int(m.type)} {_fmt(m.roughness)} {_fmt(m.reflectiveness)} "
                     f"{_fmt(m.translucency)} {_fmt(m.ior)}")
            for slot in TEX_SLOTS:
                L.append(m.textures.get(slot, ""))
    with open(path, "w") as f:
        f.write("\n".join(L) + "\n")


# --------------------------------------------------------------------------------------------
# geometry generators (all in OBJ space; the loader negates x — pathtracer.cpp:74)


def quad(name: str, p0, p1, p2, p3, uv: bool = False) -> MeshGroup:
    pos = np.array([p0, p1, p2, p3], dtype=np.float64)
    faces = np.array([[0, 1, 2], [0, 2, 3]])
    uvs = np.array([[0, 0], [1, 0], [1, 1], [0, 1]], dtype=np.float64) if uv else None
    return MeshGroup(name, pos, faces, uvs)


def cornell_groups(uv: bool = False) -> Tuple[List[MeshGroup], List[MaterialDesc]]:
    """The 12-triangle Cornell box of SURVEY.md §8(d2): 10 wall triangles on [-1,1]^3, open
    towards -z, plus a 0.6x0.6 two-triangle ceiling light at y = 0.99."""
    g = [
        quad("floor", (-1, -1, -1), (1, -1, -1), (1, -1, 1), (-1, -1, 1), uv),
        quad("ceiling", (-1, 1, -1), (-1, 1, 1), (1, 1, 1), (1, 1, -1), uv),
        quad("back", (-1, -1, 1), (1, -1, 1), (1, 1, 1), (-1, 1, 1), uv),
        # OBJ x is negated on load: OBJ x=+1 becomes world x=-1
        quad("left", (1, -1, -1), (1, 1, -1), (1, 1, 1), (1, -1, 1), uv),
        quad("right", (-1, -1, -1), (-1, -1, 1), (-1, 1, 1), (-1, 1, -1), uv),
        quad("light", (-0.3, 0.99, -0.3), (-0.3, 0.99, 0.3), (0.3, 0.99, 0.3), (0.3, 0.99, -0.3), uv),
    ]
    grey = (0.75, 0.75, 0.75)
    m = [
        MaterialDesc(diffuse=grey), MaterialDesc(diffuse=grey), MaterialDesc(diffuse=grey),
        MaterialDesc(diffuse=(0.75, 0.25, 0.25)), MaterialDesc(diffuse=(0.25, 0.75, 0.25)),
        MaterialDesc(diffuse=grey, emissive=(1.0, 1.0, 1.0), emissive_intensity=1.0),
    ]
    return g, m


def uv_sphere(name: str, center, radius: float, nu: int = 64, nv: int = 32, smooth: bool = True,
              displace: float = 0.0, seed: int = 0) -> MeshGroup:
    """UV sphere with per-vertex normals and uvs; optional seeded smooth radial displacement."""
    u = np.linspace(0.0, 1.0, nu + 1)
    v = np.linspace(0.0, 1.0, nv + 1)
    uu, vv = np.meshgrid(u, v)
    theta = uu * 2.0 * np.pi
    phi = vv * np.pi
    d = np.stack([np.sin(phi) * np.cos(theta), np.cos(phi), np.sin(phi) * np.sin(theta)], axis=-1)
    r = np.full(uu.shape, radius)
    if displace > 0.0:
        rng = np.random.default_rng(seed)
        for _ in range(6):
            k = rng.integers(2, 9, size=3)
            ph = rng.uniform(0, 2 * np.pi, size=3)
            r = r + displace * radius / 6.0 * (np.sin(k[0] * theta + ph[0]) * np.sin(k[1] * phi + ph[1])
                                               * np.cos(k[2] * phi + ph[2]))
    pos = np.asarray(center, dtype=np.float64) + d * r[..., None]
    idx = np.arange((nu + 1) * (nv + 1)).reshape(nv + 1, nu + 1)
    a = idx[:-1, :-1].ravel(); b = idx[:-1, 1:].ravel(); c = idx[1:, 1:].ravel(); e = idx[1:, :-1].ravel()
    faces = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, e], 1)], axis=0)
    # drop degenerate pole triangles
    P = pos.reshape(-1, 3)
    area = np.linalg.norm(np.cross(P[faces[:, 1]] - P[faces[:, 0]], P[faces[:, 2]] - P[faces[:, 0]]), axis=1)
    faces = faces[area > 1e-12]
    return MeshGroup(name, P, faces, np.stack([uu, vv], -1).reshape(-1, 2), d.reshape(-1, 3), smooth)


def heightfield(name: str, nx: int, nz: int, x0: float, x1: float, z0: float, z1: float, y0: float,
                amp: float, seed: int) -> MeshGroup:
    """nx x nz cells (2 triangles each) displaced height field."""
    rng = np.random.default_rng(seed)
    x = np.linspace(x0, x1, nx + 1)
    z = np.linspace(z0, z1, nz + 1)
    xx, zz = np.meshgrid(x, z)
    y = np.full(xx.shape, y0)
    for _ in range(8):
        kx, kz = rng.uniform(1.0, 14.0, size=2)
        ph = rng.uniform(0, 2 * np.pi, size=2)
        y = y + amp / 8.0 * np.sin(kx * xx + ph[0]) * np.cos(kz * zz + ph[1])
    pos = np.stack([xx, y, zz], -1).reshape(-1, 3)
    idx = np.arange((nx + 1) * (nz + 1)).reshape(nz + 1, nx + 1)
    a = idx[:-1, :-1].ravel(); b = idx[:-1, 1:].ravel(); c = idx[1:, 1:].ravel(); e = idx[1:, :-1].ravel()
    faces = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, e], 1)], axis=0)
    uv = np.stack([(xx - x0) / (x1 - x0), (zz - z0) / (z1 - z0)], -1).reshape(-1, 2)
    return MeshGroup(name, pos, faces, uv, None, False)


# --------------------------------------------------------------------------------------------
# procedural textures (PCG-free: numpy Generator with fixed seeds)


def tex_checker(n: int = 256, cells: int = 8, a=(230, 230, 230), b=(40, 60, 200)) -> np.ndarray:
    y, x = np.mgrid[0:n, 0:n]
    m = ((x * cells // n) + (y * cells // n)) % 2
    return np.where(m[..., None] == 0, np.array(a, dtype=np.uint8), np.array(b, dtype=np.uint8)).astype(np.uint8)


def tex_normal_waves(n: int = 256, waves: int = 6, strength: float = 0.5) -> np.ndarray:
    y, x = np.mgrid[0:n, 0:n] / float(n)
    dx = strength * np.cos(2 * np.pi * waves * x)
    dy = strength * np.cos(2 * np.pi * waves * y)
    nrm = np.stack([-dx, -dy, np.ones_like(dx)], -1)
    nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
    return np.clip((nrm * 0.5 + 0.5) * 255.0 + 0.5, 0, 255).astype(np.uint8)


def tex_noise(n: int = 256, seed: int = 1234, lo: int = 0, hi: int = 255, cell: int = 8) -> np.ndarray:
    rng = np.random.default_rng(seed)
    g = rng.integers(lo, hi + 1, size=(n // cell, n // cell), dtype=np.int64)
    t = np.kron(g, np.ones((cell, cell), dtype=np.int64)).astype(np.uint8)
    return np.stack([t, t, t], -1)


def tex_dots(n: int = 256, cells: int = 16, radius: float = 0.35) -> np.ndarray:
    y, x = np.mgrid[0:n, 0:n] / float(n) * cells
    fx, fy = x - np.floor(x) - 0.5, y - np.floor(y) - 0.5
    m = (fx * fx + fy * fy) < radius * radius
    t = np.where(m, 0, 255).astype(np.uint8)
    return np.stack([t, t, t], -1)


# --------------------------------------------------------------------------------------------
# the BASELINE.json configurations


def _cornell_object(out_dir: str, uv: bool = False) -> ObjectDesc:
    groups, mats = cornell_groups(uv)
    obj = os.path.join(out_dir, "cornell.obj")
    write_obj(obj, groups)
    return ObjectDesc(obj, "cornell", [ElementDesc(g.name, m) for g, m in zip(groups, mats)])


def make_cornell(out_dir: str, width: int = 512, height: int = 512, depth: int = 4) -> SceneDesc:
    """C1 / C2: 12-triangle Cornell box, pinhole (cameraF huge)."""
    sc = SceneDesc(trace_depth=depth, width=width, height=height, focal_dist=3.5, camera_f=1.0e9)
    sc.objects.append(_cornell_object(out_dir))
    return sc


def make_spheres(out_dir: str, width: int = 1280, height: int = 720, depth: int = 8,
                 nu: int = 64, nv: int = 32, tex_size: int = 1024) -> SceneDesc:
    """C3: Cornell shell + 5x2 textured UV spheres (one translucent) + thin-lens DOF."""
    sc = SceneDesc(trace_depth=depth, width=width, height=height, focal_dist=3.0, camera_f=1.0, pinhole=False)
    sc.objects.append(_cornell_object(out_dir, uv=True))
    tex = {
        "albedo": (os.path.join(out_dir, "albedo.ppm"), tex_checker(tex_size)),
        "normal": (os.path.join(out_dir, "normal.ppm"), tex_normal_waves(tex_size)),
        "rough": (os.path.join(out_dir, "rough.ppm"), tex_noise(tex_size, 1234, 20, 235)),
        "metal": (os.path.join(out_dir, "metal.ppm"), tex_noise(tex_size, 4321, 0, 255, 16)),
        "opacity": (os.path.join(out_dir, "opacity.ppm"), tex_dots(tex_size)),
    }
    for p, img in tex.values():
        write_ppm(p, img)
    groups: List[MeshGroup] = []
    elems: List[ElementDesc] = []
    rough = [0.0, 0.25, 0.5, 0.75, 1.0]
    refl = [0.2, 0.9]
    k = 0
    for row in range(2):
        for col in range(5):
            cx = -0.72 + 0.36 * col
            cy = -0.78 + 0.5 * row
            cz = -0.25 + 0.55 * row
            # OBJ x is negated on load
            groups.append(uv_sphere(f"sphere{k}", (-cx, cy, cz), 0.17, nu, nv, True))
            m = MaterialDesc(diffuse=(0.8, 0.8, 0.8), roughness=rough[col], reflectiveness=refl[row])
            if k == 2:
                m = MaterialDesc(type=TRANSLUCENT, diffuse=(0.95, 0.95, 0.95), roughness=0.0,
                                 reflectiveness=0.0, translucency=1.0, ior=1.5)
            elif k == 7:
                m = MaterialDesc(type=TRANSLUCENT, diffuse=(0.9, 0.95, 1.0), roughness=0.3,
                                 reflectiveness=0.1, translucency=0.8, ior=1.33)
            else:
                if k % 2 == 0:
                    m.textures["diffuse"] = tex["albedo"][0]
                if k in (1, 6):
                    m.textures["normal"] = tex["normal"][0]
                if k in (3, 8):
                    m.textures["roughness"] = tex["rough"][0]
                if k in (4, 5):
                    m.textures["metallic"] = tex["metal"][0]
                if k == 9:
                    m.textures["opacity"] = tex["opacity"][0]
            elems.append(ElementDesc(f"sphere{k}", m))
            k += 1
    obj = os.path.join(out_dir, "spheres.obj")
    write_obj(obj, groups)
    sc.objects.append(ObjectDesc(obj, "spheres", elems))
    return sc


def make_blob(out_dir: str, width: int = 1920, height: int = 1080, depth: int = 8,
              grid: int = 187) -> SceneDesc:
    """C4 stand-in for the Stanford bunny (not in the container): a seeded displaced sphere with
    ~2*grid^2 (= 69.9 k at grid 187) smooth-shaded triangles inside the Cornell shell."""
    sc = SceneDesc(trace_depth=depth, width=width, height=height, focal_dist=3.5, camera_f=1.0e9)
    sc.objects.append(_cornell_object(out_dir))
    g = uv_sphere("blob", (0.0, -0.35, 0.1), 0.6, grid, grid, True, displace=0.6, seed=7)
    obj = os.path.join(out_dir, "blob.obj")
    write_obj(obj, [g])
    sc.objects.append(ObjectDesc(obj, "blob", [ElementDesc("blob", MaterialDesc(
        diffuse=(0.8, 0.7, 0.55), roughness=0.4, reflectiveness=0.15))]))
    return sc


def make_heightfield(out_dir: str, width: int = 1920, height: int = 1080, depth: int = 12,
                     nx: int = 1000, nz: int = 500) -> SceneDesc:
    """C5: ~1 M-triangle (nx*nz*2) displaced height field folded into the Cornell shell."""
    sc = SceneDesc(trace_depth=depth, width=width, height=height, focal_dist=3.5, camera_f=1.0e9)
    sc.objects.append(_cornell_object(out_dir))
    g = heightfield("terrain", nx, nz, -0.98, 0.98, -0.98, 0.98, -0.7, 0.5, 42)
    obj = os.path.join(out_dir, "terrain.obj")
    write_obj(obj, [g])
    sc.objects.append(ObjectDesc(obj, "terrain", [ElementDesc("terrain", MaterialDesc(
        diffuse=(0.6, 0.7, 0.5), roughness=0.8, reflectiveness=0.1))]))
    return sc


CONFIGS = {
    # name: (builder, kwargs, spp)  — BASELINE.json.configs[0..4]
    "C1": (make_cornell, dict(width=512, height=512, depth=4), 16),
    "C2": (make_cornell, dict(width=1280, height=720, depth=8), 256),
    "C3": (make_spheres, dict(width=1280, height=720, depth=8), 512),
    "C4": (make_blob, dict(width=1920, height=1080, depth=8), 256),
    "C5": (make_heightfield, dict(width=1920, height=1080, depth=12), 1024),
}


def build_config(name: str, out_dir: str, **overrides) -> Tuple[str, SceneDesc, int]:
    """Write config `name` under out_dir; returns (pts path, scene, spp)."""
    fn, kw, spp = CONFIGS[name]
    kw = dict(kw); kw.update(overrides)
    sc = fn(out_dir, **kw)
    pts = os.path.join(out_dir, f"{name}.pts")
    write_pts(pts, sc)
    return pts, sc, spp
