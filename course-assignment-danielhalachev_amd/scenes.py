"""Deterministic synthetic `.crtscene` generators for the five BASELINE configs.

The reference's original scene files are not in its repository, so every
workload is produced here, in the reference's own JSON schema
(reference: SourceCode/src/SceneParser.cpp:17-35,88-322), from fixed parameters and a
splitmix64 stream -- never from a library RNG whose sequence could change between versions.

A scene is a plain dict:

    {"settings": {"background_color": [r,g,b],
                  "image_settings": {"width": W, "height": H, "bucket_size": B}},
     "camera":   {"matrix": [9 floats, row major], "position": [3 floats]},
     "lights":   [{"intensity": int, "position": [3 floats]}, ...],
     "textures": [ {...}, ... ]            (optional; reference: SceneParser.cpp:150-209)
     "materials":[{"type": "diffuse"|"reflective"|"refractive"|"constant",
                   "albedo": [r,g,b] | "texture name", "smooth_shading": bool, "ior": float}],
     "objects":  [{"material_index": int, "vertices": float32[nv,3],
                   "uvs": float32[nv,3] (optional), "triangles": uint32[nt,3]}]}

All coordinates are float32 values; `to_json` prints them with 9 significant digits, which
survives the reference's double -> float conversion (`GetFloat`, SceneParser.cpp:84,294) exactly.
`to_blob` writes the same scene as a flat little-endian binary ("CRTS") that the oracle's C code
and the reference driver under oracle/ read without needing a JSON library.
"""
from __future__ import annotations

import io
import json
import math
import struct

import numpy as np

F32 = np.float32
MASK64 = (1 << 64) - 1

MATERIAL_TYPES = {"diffuse": 0, "reflective": 1, "constant": 2, "refractive": 3}  # Material.h:7
TEXTURE_KINDS = {"albedo": 0, "edges": 1, "checker": 2, "bitmap": 3}


# ----------------------------------------------------------------------------- RNG
class SplitMix64:
    """splitmix64; the only source of randomness in the generators."""

    def __init__(self, seed: int):
        self.state = seed & MASK64

    def next_u64(self) -> int:
        self.state = (self.state + 0x9E3779B97F4A7C15) & MASK64
        z = self.state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
        return z ^ (z >> 31)

    def uniform(self, lo: float = 0.0, hi: float = 1.0) -> float:
        return lo + (hi - lo) * ((self.next_u64() >> 11) * (1.0 / (1 << 53)))


def _hash_u64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        z = x.astype(np.uint64) + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


# ----------------------------------------------------------------------------- meshes
def _mesh(material_index, vertices, triangles, uvs=None):
    m = {
        "material_index": int(material_index),
        "vertices": np.ascontiguousarray(vertices, dtype=F32).reshape(-1, 3),
        "triangles": np.ascontiguousarray(triangles, dtype=np.uint32).reshape(-1, 3),
    }
    if uvs is not None:
        m["uvs"] = np.ascontiguousarray(uvs, dtype=F32).reshape(-1, 3)
    return m


def quad(material_index, p00, p10, p11, p01, nu=1, nv=1, uv_scale=(1.0, 1.0)):
    """Planar grid p00->p10 (u) / p00->p01 (v), counter-clockwise seen from the side its normal
    (u x v) points to.  UVs run 0..uv_scale over the quad (all >= 0)."""
    p00, p10, p11, p01 = (np.asarray(p, dtype=np.float64) for p in (p00, p10, p11, p01))
    us = np.linspace(0.0, 1.0, nu + 1)
    vs = np.linspace(0.0, 1.0, nv + 1)
    U, V = np.meshgrid(us, vs, indexing="xy")  # [nv+1, nu+1]
    P = ((1 - U) * (1 - V))[..., None] * p00 + (U * (1 - V))[..., None] * p10 \
        + (U * V)[..., None] * p11 + ((1 - U) * V)[..., None] * p01
    uv = np.stack([U * uv_scale[0], V * uv_scale[1], np.zeros_like(U)], axis=-1)
    idx = np.arange((nu + 1) * (nv + 1)).reshape(nv + 1, nu + 1)
    a = idx[:-1, :-1].ravel()
    b = idx[:-1, 1:].ravel()
    c = idx[1:, 1:].ravel()
    d = idx[1:, :-1].ravel()
    tris = np.concatenate([np.stack([a, b, c], 1), np.stack([a, c, d], 1)], 0)
    return _mesh(material_index, P.reshape(-1, 3), tris, uv.reshape(-1, 3))


def uv_sphere(material_index, center, radius, n_lon, n_lat):
    """Outward-facing UV sphere: pole fans + quad bands, no degenerate triangles.
    Triangle count = 2 * n_lon * (n_lat - 1)."""
    cx, cy, cz = center
    verts = [(cx, cy + radius, cz)]
    uvs = [(0.5, 1.0, 0.0)]
    for i in range(1, n_lat):
        th = math.pi * i / n_lat
        for j in range(n_lon):
            ph = 2.0 * math.pi * j / n_lon
            verts.append((cx + radius * math.sin(th) * math.cos(ph),
                          cy + radius * math.cos(th),
                          cz + radius * math.sin(th) * math.sin(ph)))
            uvs.append((j / n_lon, 1.0 - i / n_lat, 0.0))
    verts.append((cx, cy - radius, cz))
    uvs.append((0.5, 0.0, 0.0))
    south = len(verts) - 1

    def ring(i, j):
        return 1 + (i - 1) * n_lon + (j % n_lon)

    tris = []
    for j in range(n_lon):
        tris.append((0, ring(1, j + 1), ring(1, j)))
    for i in range(1, n_lat - 1):
        for j in range(n_lon):
            a, b = ring(i, j), ring(i, j + 1)
            c, d = ring(i + 1, j + 1), ring(i + 1, j)
            tris.append((a, b, c))
            tris.append((a, c, d))
    for j in range(n_lon):
        tris.append((south, ring(n_lat - 1, j), ring(n_lat - 1, j + 1)))
    return _mesh(material_index, np.array(verts), np.array(tris), np.array(uvs))


def torus_knot(material_index, center, scale, n_u, n_v, seed, p=2, q=3,
               tube=0.22, bump=0.035):
    """Seeded, displaced (p,q) torus knot: the 'dragon stand-in' of SURVEY.md §8d (C4).
    n_u segments along the curve, n_v around the tube -> 2*n_u*n_v triangles, outward facing."""
    rng = SplitMix64(seed)
    waves = [(rng.uniform(3.0, 17.0), rng.uniform(2.0, 9.0), rng.uniform(0.0, 2 * math.pi),
              rng.uniform(0.4, 1.0)) for _ in range(6)]
    t = np.linspace(0.0, 2.0 * math.pi, n_u, endpoint=False)
    s = np.linspace(0.0, 2.0 * math.pi, n_v, endpoint=False)

    def curve(tt):
        r = 1.0 + 0.45 * np.cos(q * tt)
        return np.stack([r * np.cos(p * tt), 0.45 * np.sin(q * tt) * 1.3, r * np.sin(p * tt)], -1)

    C = curve(t)
    eps = 1e-4
    T = curve(t + eps) - curve(t - eps)
    T /= np.linalg.norm(T, axis=-1, keepdims=True)
    up = np.array([0.0, 1.0, 0.0])
    B = np.cross(T, up)
    B /= np.linalg.norm(B, axis=-1, keepdims=True)
    N = np.cross(B, T)
    tt, ss = np.meshgrid(t, s, indexing="ij")  # [n_u, n_v]
    disp = np.zeros_like(tt)
    for fu, fv, ph, amp in waves:
        disp += amp * np.sin(np.round(fu) * tt + np.round(fv) * ss + ph)
    rad = tube * (1.0 + bump / tube * disp / len(waves) * 2.0)
    P = C[:, None, :] + rad[..., None] * (np.cos(ss)[..., None] * N[:, None, :]
                                           + np.sin(ss)[..., None] * B[:, None, :])
    P = P * scale + np.asarray(center, dtype=np.float64)
    uv = np.stack([tt / (2 * math.pi), ss / (2 * math.pi), np.zeros_like(tt)], -1)
    idx = np.arange(n_u * n_v).reshape(n_u, n_v)
    a = idx
    b = np.roll(idx, -1, axis=0)
    c = np.roll(np.roll(idx, -1, axis=0), -1, axis=1)
    d = np.roll(idx, -1, axis=1)
    tris = np.concatenate([np.stack([a.ravel(), b.ravel(), c.ravel()], 1),
                           np.stack([a.ravel(), c.ravel(), d.ravel()], 1)], 0)
    m = _mesh(material_index, P.reshape(-1, 3), tris, uv.reshape(-1, 3))
    # make the winding outward: the face normal of triangle 0 must point away from the curve
    v = m["vertices"].astype(np.float64)
    t0 = m["triangles"][0]
    n0 = np.cross(v[t0[1]] - v[t0[0]], v[t0[2]] - v[t0[0]])
    outward = v[t0[0]] - (C[0] * scale + np.asarray(center, dtype=np.float64))
    if np.dot(n0, outward) < 0:
        m["triangles"] = np.ascontiguousarray(m["triangles"][:, ::-1])
    return m


def noise_bitmap(width, height, seed):
    """Seeded RGB8 'noise + gradient' image for the bitmap texture (C5)."""
    y, x = np.mgrid[0:height, 0:width].astype(np.uint64)
    out = np.empty((height, width, 3), dtype=np.uint8)
    for ch in range(3):
        cell = (y // np.uint64(8)) * np.uint64(4099) + (x // np.uint64(8)) + np.uint64(seed * 7919 + ch * 104729)
        h = (_hash_u64(cell) >> np.uint64(40)).astype(np.float64) / float(1 << 24)
        grad = (x.astype(np.float64) / max(width - 1, 1)) if ch != 1 else (y.astype(np.float64) / max(height - 1, 1))
        val = 0.55 * h + 0.45 * (grad if ch < 2 else 1.0 - grad)
        out[..., ch] = np.clip(val * 255.0, 0, 255).astype(np.uint8)
    return out


# ----------------------------------------------------------------------------- scene pieces
IDENTITY = [1.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 1.0]


def _settings(width, height, bucket=48, bg=(0.0, 0.5, 0.0)):
    return {"background_color": list(bg),
            "image_settings": {"width": int(width), "height": int(height), "bucket_size": int(bucket)}}


def _room(materials, objects, x0=-3.0, x1=3.0, y0=-1.5, y1=2.5, z0=-8.0, z1=1.0, sub=1):
    """Five inward-facing walls (floor, ceiling, back, left, right), one mesh each."""
    base = len(materials)
    materials += [
        {"type": "diffuse", "albedo": [0.70, 0.70, 0.70], "smooth_shading": False},  # floor
        {"type": "diffuse", "albedo": [0.85, 0.85, 0.85], "smooth_shading": False},  # ceiling
        {"type": "diffuse", "albedo": [0.80, 0.78, 0.70], "smooth_shading": False},  # back
        {"type": "diffuse", "albedo": [0.80, 0.15, 0.12], "smooth_shading": False},  # left
        {"type": "diffuse", "albedo": [0.12, 0.25, 0.80], "smooth_shading": False},  # right
    ]
    objects += [
        quad(base + 0, (x0, y0, z1), (x1, y0, z1), (x1, y0, z0), (x0, y0, z0), sub, sub),  # floor, +y
        quad(base + 1, (x0, y1, z0), (x1, y1, z0), (x1, y1, z1), (x0, y1, z1), sub, sub),  # ceiling, -y
        quad(base + 2, (x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0), sub, sub),  # back, +z
        quad(base + 3, (x0, y0, z1), (x0, y0, z0), (x0, y1, z0), (x0, y1, z1), sub, sub),  # left, +x
        quad(base + 4, (x1, y0, z0), (x1, y0, z1), (x1, y1, z1), (x1, y1, z0), sub, sub),  # right, -x
    ]


def _strip_uvs(scene):
    for o in scene["objects"]:
        o.pop("uvs", None)
    return scene


# ----------------------------------------------------------------------------- the five configs
def hw07_scene(width=640, height=480, detail=1.0):
    """C1 'HW07-like': one flat-shaded diffuse mesh + one light; depth 1."""
    n_lon, n_lat = max(8, int(32 * detail)), max(4, int(16 * detail))
    materials = [{"type": "diffuse", "albedo": [0.85, 0.45, 0.20], "smooth_shading": False}]
    objects = [uv_sphere(0, (0.1, 0.05, -3.2), 1.1, n_lon, n_lat)]
    scene = {"settings": _settings(width, height),
             "camera": {"matrix": list(IDENTITY), "position": [0.0, 0.0, 0.0]},
             "lights": [{"intensity": 170, "position": [-2.5, 2.75, 0.5]}],
             "materials": materials, "objects": objects}
    return _strip_uvs(scene)


def hw08_scene(width=1280, height=720, detail=1.0):
    """C2 'HW08-like': floor + smooth diffuse sphere + three lights; primary + shadow rays."""
    n_lon, n_lat = max(8, int(64 * detail)), max(4, int(32 * detail))
    materials = [{"type": "diffuse", "albedo": [0.75, 0.75, 0.72], "smooth_shading": False},
                 {"type": "diffuse", "albedo": [0.25, 0.55, 0.90], "smooth_shading": True}]
    objects = [quad(0, (-6.0, -1.0, 1.0), (6.0, -1.0, 1.0), (6.0, -1.0, -11.0), (-6.0, -1.0, -11.0), 4, 4),
               uv_sphere(1, (0.0, 0.0, -4.0), 1.0, n_lon, n_lat)]
    lights = [{"intensity": 220, "position": [-3.0, 3.5, -1.0]},
              {"intensity": 160, "position": [3.25, 2.5, -2.0]},
              {"intensity": 90, "position": [0.5, 4.0, -7.5]}]
    scene = {"settings": _settings(width, height),
             "camera": {"matrix": list(IDENTITY), "position": [0.0, 0.25, 0.5]},
             "lights": lights, "materials": materials, "objects": objects}
    return _strip_uvs(scene)


def _room_lights():
    return [{"intensity": 42, "position": [-1.55, 1.6, -2.5]},
            {"intensity": 42, "position": [1.6, 1.7, -5.5]},
            {"intensity": 30, "position": [0.1, 1.5, -0.4]},
            {"intensity": 28, "position": [-0.6, 0.6, -6.9]}]


def hw11_scene(width=1920, height=1080, detail=1.0):
    """C3 'HW11-like': coloured box room + one reflective and one refractive (ior 1.5, smooth)
    tessellated sphere + four lights; reflect + refract to depth 8."""
    n_lon, n_lat = max(8, int(96 * detail)), max(4, int(48 * detail))
    materials, objects = [], []
    _room(materials, objects)
    materials += [{"type": "reflective", "albedo": [0.90, 0.90, 0.92], "smooth_shading": True},
                  {"type": "refractive", "albedo": [0.0, 0.0, 0.0], "smooth_shading": True, "ior": 1.5}]
    objects += [uv_sphere(5, (-1.25, -0.65, -5.0), 0.85, n_lon, n_lat),
                uv_sphere(6, (1.15, -0.55, -3.9), 0.8, n_lon, n_lat)]
    scene = {"settings": _settings(width, height),
             "camera": {"matrix": list(IDENTITY), "position": [0.0, 0.4, 0.6]},
             "lights": _room_lights(), "materials": materials, "objects": objects}
    return _strip_uvs(scene)


def hw14_scene(width=1920, height=1080, detail=1.0):
    """C4 'HW14-like' (headline): the C3 room with a ~200k-triangle seeded displaced torus knot on a
    raised plane, plus a reflective and a refractive sphere; depth 8."""
    n_u = max(24, int(round(1000 * detail)))
    n_v = max(8, int(round(100 * math.sqrt(detail)))) if detail < 1.0 else 100
    n_lon, n_lat = max(8, int(64 * detail)), max(4, int(32 * detail))
    materials, objects = [], []
    _room(materials, objects)
    materials += [{"type": "diffuse", "albedo": [0.55, 0.52, 0.48], "smooth_shading": False},   # plane
                  {"type": "diffuse", "albedo": [0.85, 0.65, 0.25], "smooth_shading": True},    # knot
                  {"type": "reflective", "albedo": [0.92, 0.92, 0.95], "smooth_shading": True},
                  {"type": "refractive", "albedo": [0.0, 0.0, 0.0], "smooth_shading": True, "ior": 1.5}]
    objects += [quad(5, (-2.2, -1.1, -2.6), (2.2, -1.1, -2.6), (2.2, -1.1, -6.6), (-2.2, -1.1, -6.6), 2, 2),
                torus_knot(6, (0.0, -0.05, -4.6), 0.82, n_u, n_v, seed=14),
                uv_sphere(7, (-2.05, -0.75, -5.9), 0.7, n_lon, n_lat),
                uv_sphere(8, (1.95, -0.85, -3.1), 0.6, n_lon, n_lat)]
    scene = {"settings": _settings(width, height),
             "camera": {"matrix": list(IDENTITY), "position": [0.0, 0.45, 0.7]},
             "lights": _room_lights(), "materials": materials, "objects": objects}
    return _strip_uvs(scene)


def hw12_scene(width=3840, height=2160, detail=1.0, bitmap_size=1024):
    """C5 'HW12-like': textured variant -- albedo / edges / checker / bitmap textures on quads and a
    large textured mesh (all UVs >= 0), plus a reflective and a refractive sphere; depth 8."""
    n_u = max(24, int(round(400 * detail)))
    n_v = max(8, int(round(64 * math.sqrt(detail)))) if detail < 1.0 else 64
    n_lon, n_lat = max(8, int(64 * detail)), max(4, int(32 * detail))
    bmp = noise_bitmap(bitmap_size, bitmap_size, seed=12)
    textures = [
        {"name": "plain grey", "type": "albedo", "albedo": [0.72, 0.72, 0.70]},
        {"name": "wire", "type": "edges", "inner_color": [0.9, 0.85, 0.2], "edge_color": [0.1, 0.1, 0.4],
         "edge_width": 0.04},
        {"name": "checker", "type": "checker", "color_A": [0.9, 0.9, 0.9], "color_B": [0.15, 0.15, 0.15],
         "square_size": 0.125},
        {"name": "noise", "type": "bitmap", "file_path": "/hw12_noise.ppm", "_pixels": bmp},
        {"name": "plain red", "type": "albedo", "albedo": [0.8, 0.2, 0.15]},
    ]
    materials = [
        {"type": "diffuse", "albedo": "checker", "smooth_shading": False},      # 0 floor
        {"type": "diffuse", "albedo": "plain grey", "smooth_shading": False},   # 1 ceiling
        {"type": "diffuse", "albedo": "noise", "smooth_shading": False},        # 2 back wall
        {"type": "diffuse", "albedo": "wire", "smooth_shading": False},         # 3 left wall
        {"type": "diffuse", "albedo": "plain red", "smooth_shading": False},    # 4 right wall
        {"type": "diffuse", "albedo": "noise", "smooth_shading": True},         # 5 knot
        {"type": "diffuse", "albedo": "wire", "smooth_shading": True},          # 6 sphere
        {"type": "reflective", "albedo": [0.9, 0.9, 0.9], "smooth_shading": True},
        {"type": "refractive", "albedo": [0.0, 0.0, 0.0], "smooth_shading": True, "ior": 1.5},
    ]
    x0, x1, y0, y1, z0, z1 = -3.0, 3.0, -1.5, 2.5, -8.0, 1.0
    objects = [
        quad(0, (x0, y0, z1), (x1, y0, z1), (x1, y0, z0), (x0, y0, z0), 2, 2, (1.5, 2.25)),
        quad(1, (x0, y1, z0), (x1, y1, z0), (x1, y1, z1), (x0, y1, z1), 1, 1),
        quad(2, (x0, y0, z0), (x1, y0, z0), (x1, y1, z0), (x0, y1, z0), 3, 2),
        quad(3, (x0, y0, z1), (x0, y0, z0), (x0, y1, z0), (x0, y1, z1), 6, 3),
        quad(4, (x1, y0, z0), (x1, y0, z1), (x1, y1, z1), (x1, y1, z0), 1, 1),
        torus_knot(5, (0.0, -0.1, -4.8), 0.8, n_u, n_v, seed=12),
        uv_sphere(6, (-2.0, -0.8, -3.4), 0.65, max(8, n_lon // 2), max(4, n_lat // 2)),
        uv_sphere(7, (-1.9, -0.7, -6.2), 0.75, n_lon, n_lat),
        uv_sphere(8, (1.9, -0.8, -3.3), 0.65, n_lon, n_lat),
    ]
    return {"settings": _settings(width, height),
            "camera": {"matrix": list(IDENTITY), "position": [0.0, 0.45, 0.7]},
            "lights": _room_lights(), "textures": textures, "materials": materials, "objects": objects}


def scatter_meshes(scene, n_objects, seed=7):
    """The scene with small spheres scattered through its room until it holds n_objects meshes: a top-level tree with hundreds of
    leaves (the reference's scenes have a handful of meshes; tools/many_meshes.py and the wide-plan tests use this)."""
    rng = np.random.default_rng(seed)
    objects = list(scene["objects"])
    n_mat = len(scene["materials"])
    while len(objects) < n_objects:
        c = rng.uniform([-2.6, -1.2, -7.5], [2.6, 2.0, -1.5]).astype(np.float32)
        m = uv_sphere(int(rng.integers(0, n_mat)), tuple(float(x) for x in c), float(rng.uniform(0.08, 0.25)), 16, 8)
        m.pop("uvs", None)
        objects.append(m)
    return dict(scene, objects=objects)


CONFIGS = {
    # name: (generator, width, height, max_depth)
    "hw07": (hw07_scene, 640, 480, 1),
    "hw08": (hw08_scene, 1280, 720, 1),
    "hw11": (hw11_scene, 1920, 1080, 8),
    "hw14": (hw14_scene, 1920, 1080, 8),
    "hw12": (hw12_scene, 3840, 2160, 8),
}
DESCRIPTIONS = {
    "hw07": "HW07 Scene-like: one flat-shaded diffuse sphere, one light",
    "hw08": "HW08 Light-like: floor + smooth diffuse sphere, three lights (primary + shadow rays)",
    "hw11": "HW11 Shading03-like: coloured room with a reflective and a refractive sphere, four lights",
    "hw14": "HW14 Optimizations02-like BVH scene: room + seeded torus knot + reflective and refractive spheres, four lights",
    "hw12": "HW12 Textures-like: albedo / edge / checker / bitmap textures on quads and a large textured mesh",
}


def make(name, width=None, height=None, detail=1.0, **kw):
    gen, w, h, _ = CONFIGS[name]
    return gen(width or w, height or h, detail, **kw)


# ----------------------------------------------------------------------------- serialisation
def _f(x) -> str:
    return "%.9g" % float(F32(x))


def _flist(a) -> str:
    return "[" + ",".join(_f(x) for x in np.asarray(a, dtype=F32).ravel()) + "]"


def to_json(scene) -> str:
    """Serialise to the reference's `.crtscene` JSON (SceneParser.cpp:17-35).  Bitmap pixel arrays
    (`_pixels`) are not part of the schema and are dropped: write them with `write_bitmaps`."""
    out = io.StringIO()
    st = scene["settings"]
    im = st["image_settings"]
    out.write('{"settings":{"background_color":%s,"image_settings":{"width":%d,"height":%d,"bucket_size":%d}},'
              % (_flist(st["background_color"]), im["width"], im["height"], im["bucket_size"]))
    cam = scene["camera"]
    out.write('"camera":{"matrix":%s,"position":%s},' % (_flist(cam["matrix"]), _flist(cam["position"])))
    out.write('"lights":[%s],' % ",".join(
        '{"intensity":%d,"position":%s}' % (l["intensity"], _flist(l["position"])) for l in scene["lights"]))
    if scene.get("textures") is not None:
        parts = []
        for t in scene["textures"]:
            d = {k: v for k, v in t.items() if not k.startswith("_")}
            items = []
            for k, v in d.items():
                if isinstance(v, str):
                    items.append("%s:%s" % (json.dumps(k), json.dumps(v)))
                elif isinstance(v, (list, tuple, np.ndarray)):
                    items.append("%s:%s" % (json.dumps(k), _flist(v)))
                else:
                    items.append("%s:%s" % (json.dumps(k), _f(v)))
            parts.append("{" + ",".join(items) + "}")
        out.write('"textures":[%s],' % ",".join(parts))
    mats = []
    for m in scene["materials"]:
        alb = m["albedo"]
        alb_s = json.dumps(alb) if isinstance(alb, str) else _flist(alb)
        s = '{"type":%s,"albedo":%s,"smooth_shading":%s' % (
            json.dumps(m["type"]), alb_s, "true" if m["smooth_shading"] else "false")
        if "ior" in m:
            s += ',"ior":%s' % _f(m["ior"])
        mats.append(s + "}")
    out.write('"materials":[%s],' % ",".join(mats))
    objs = []
    for o in scene["objects"]:
        s = '{"material_index":%d,"vertices":%s,' % (o["material_index"], _flist(o["vertices"]))
        if "uvs" in o:
            s += '"uvs":%s,' % _flist(o["uvs"])
        s += '"triangles":[%s]}' % ",".join(str(int(i)) for i in o["triangles"].ravel())
        objs.append(s)
    out.write('"objects":[%s]}' % ",".join(objs))
    return out.getvalue()


def write_bitmaps(scene, folder: str):
    """Write every bitmap texture as a binary P6 file at `folder + file_path` (the reference joins
    the two without a separator, SceneParser.cpp:201).  Returns the list of paths written."""
    paths = []
    for t in scene.get("textures") or []:
        if t["type"] == "bitmap":
            px = t["_pixels"]
            path = folder + t["file_path"]
            with open(path, "wb") as f:
                f.write(b"P6\n%d %d\n255\n" % (px.shape[1], px.shape[0]))
                f.write(np.ascontiguousarray(px, dtype=np.uint8).tobytes())
            paths.append(path)
    return paths


def resolved_materials(scene):
    """Materials with the texture name resolved to an index (-1 = constant albedo) exactly as the
    reference's lookup does (first texture whose name matches, SceneParser.cpp:245-251)."""
    tex = scene.get("textures") or []
    out = []
    for m in scene["materials"]:
        alb = m["albedo"]
        if isinstance(alb, str):
            ti = next(i for i, t in enumerate(tex) if t["name"] == alb)
            albedo = [0.0, 0.0, 0.0]
        else:
            ti = -1
            albedo = [float(F32(x)) for x in alb]
        out.append({"type": MATERIAL_TYPES[m["type"]], "albedo": albedo,
                    "smooth": 1 if m["smooth_shading"] else 0,
                    "ior": float(F32(m.get("ior", 0.0))), "texture": ti})
    return out


def to_blob(scene) -> bytes:
    """Flat little-endian 'CRTS' v1 binary of the scene (layout documented in oracle/scene_blob.h)."""
    b = io.BytesIO()
    st = scene["settings"]
    im = st["image_settings"]
    b.write(b"CRTS")
    b.write(struct.pack("<I", 1))
    b.write(struct.pack("<III", im["width"], im["height"], im["bucket_size"]))
    b.write(np.asarray(st["background_color"], dtype=F32).tobytes())
    b.write(np.asarray(scene["camera"]["position"], dtype=F32).tobytes())
    b.write(np.asarray(scene["camera"]["matrix"], dtype=F32).tobytes())
    tex = scene.get("textures") or []
    b.write(struct.pack("<I", len(tex)))
    for t in tex:
        kind = TEXTURE_KINDS[t["type"]]
        ca = cb = (0.0, 0.0, 0.0)
        scalar = 0.0
        w = h = 0
        if kind == 0:
            ca = t["albedo"]
        elif kind == 1:
            ca, cb, scalar = t["inner_color"], t["edge_color"], t["edge_width"]
        elif kind == 2:
            ca, cb, scalar = t["color_A"], t["color_B"], t["square_size"]
        else:
            h, w = t["_pixels"].shape[:2]
        b.write(struct.pack("<I", kind))
        b.write(np.asarray(ca, dtype=F32).tobytes())
        b.write(np.asarray(cb, dtype=F32).tobytes())
        b.write(struct.pack("<fII", float(F32(scalar)), w, h))
        if kind == 3:
            b.write(np.ascontiguousarray(t["_pixels"], dtype=np.uint8).tobytes())
    mats = resolved_materials(scene)
    b.write(struct.pack("<I", len(mats)))
    for m in mats:
        b.write(struct.pack("<I", m["type"]))
        b.write(np.asarray(m["albedo"], dtype=F32).tobytes())
        b.write(struct.pack("<Ifi", m["smooth"], m["ior"], m["texture"]))
    b.write(struct.pack("<I", len(scene["lights"])))
    for l in scene["lights"]:
        b.write(np.asarray(l["position"], dtype=F32).tobytes())
        b.write(struct.pack("<I", int(l["intensity"])))
    b.write(struct.pack("<I", len(scene["objects"])))
    for o in scene["objects"]:
        v = np.ascontiguousarray(o["vertices"], dtype=F32)
        t = np.ascontiguousarray(o["triangles"], dtype=np.uint32)
        has_uv = 1 if "uvs" in o else 0
        b.write(struct.pack("<IIII", o["material_index"], v.shape[0], t.shape[0], has_uv))
        b.write(v.tobytes())
        if has_uv:
            b.write(np.ascontiguousarray(o["uvs"], dtype=F32).tobytes())
        b.write(t.tobytes())
    return b.getvalue()


def triangle_count(scene) -> int:
    return int(sum(o["triangles"].shape[0] for o in scene["objects"]))
