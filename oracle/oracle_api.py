"""TEST INFRASTRUCTURE -- ctypes wrapper around oracle/liboracle.so (cpu_ref.c) and the prebuilt
real-reference binaries under oracle/_ref/.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; nothing in the product package does.
"""
from __future__ import annotations

import ctypes as C
import json
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
REF_PLAIN = os.path.join(HERE, "_ref", "ref_render")
REF_TEX = os.path.join(HERE, "_ref", "ref_render_tex")

COUNTER_NAMES = ["box_tests", "tri_tests", "leaf_index_reads", "shaded_hits", "light_evals", "texel_fetches",
                 "primary_rays", "secondary_rays", "shadow_rays"]


class Options(C.Structure):
    _fields_ = [("max_depth", C.c_uint32), ("shadow_bias", C.c_float), ("reflection_bias", C.c_float),
                ("refraction_bias", C.c_float), ("use_gi", C.c_uint32), ("gi_sample_size", C.c_uint32),
                ("rays_per_pixel", C.c_uint32), ("monte_carlo_bias", C.c_float), ("gi_seed", C.c_uint32)]


def make_options(max_depth=5, shadow_bias=1e-4, reflection_bias=1e-4, refraction_bias=1e-4, use_gi=0, gi_sample_size=2,
                 rays_per_pixel=1, monte_carlo_bias=1e-4, gi_seed=0):
    """RenderOptions (RayTracer.h:25-50); use_gi=1 selects the GI / multi-sample mode with the counter-based generator
    cpu_ref.c defines (gi_seed = the frame's seed)."""
    return Options(max_depth, shadow_bias, reflection_bias, refraction_bias, use_gi, gi_sample_size, rays_per_pixel,
                   monte_carlo_bias, gi_seed)


def build(force=False):
    """(Re)build liboracle.so with gcc; also the real-reference binaries when /root/reference exists."""
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(
            os.path.join(HERE, "cpu_ref.c")):
        subprocess.check_call(["make", "-C", HERE, "oracle"], stdout=subprocess.DEVNULL)
    if os.path.exists("/root/reference/SourceCode/src/RayTracer.cpp"):
        subprocess.check_call(["make", "-C", HERE, "-j4", "ref"], stdout=subprocess.DEVNULL)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.oracle_scene_load.restype = C.c_void_p
        L.oracle_scene_load.argtypes = [C.c_char_p, C.c_size_t]
        L.oracle_scene_free.argtypes = [C.c_void_p]
        L.oracle_scene_dims.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint32)] * 3
        L.oracle_set_camera.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_render.argtypes = [C.c_void_p, C.POINTER(Options), C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_bucket_grid.restype = C.c_uint32
        L.oracle_bucket_grid.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
        L.oracle_box_hit.argtypes = [C.c_void_p] * 4
        L.oracle_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_occluded.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]
        L.oracle_shoot.argtypes = [C.c_void_p, C.POINTER(Options), C.c_void_p, C.c_void_p, C.c_int, C.c_uint,
                                   C.c_void_p]
        L.oracle_camera_ray.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_void_p]
        L.oracle_mesh_count.restype = C.c_uint32
        L.oracle_mesh_count.argtypes = [C.c_void_p]
        L.oracle_tree_node_count.restype = C.c_uint32
        L.oracle_tree_node_count.argtypes = [C.c_void_p, C.c_int]
        L.oracle_tree_index_total.restype = C.c_uint64
        L.oracle_tree_index_total.argtypes = [C.c_void_p, C.c_int]
        L.oracle_tree_dump.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_mesh_sizes.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.oracle_mesh_normals.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.oracle_quantize.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.oracle_write_ppm.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.oracle_powf.restype = C.c_float
        L.oracle_powf.argtypes = [C.c_float, C.c_float]
        for name in ("oracle_sinf", "oracle_cosf"):
            getattr(L, name).restype = C.c_float
            getattr(L, name).argtypes = [C.c_float]
        L.oracle_gi_uniform.restype = C.c_float
        L.oracle_gi_uniform.argtypes = [C.c_uint32, C.c_uint32]
        L.oracle_gi_mix.restype = C.c_uint32
        L.oracle_gi_mix.argtypes = [C.c_uint32, C.c_uint32]
        L.oracle_sincos_array.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.oracle_gi_array.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.oracle_sincos_vs_libm.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class OracleScene:
    """A scene loaded into the CPU restatement (trees built exactly as the reference builds them)."""

    def __init__(self, blob: bytes):
        self._L = lib()
        self._h = self._L.oracle_scene_load(blob, len(blob))
        if not self._h:
            raise ValueError("oracle: bad scene blob")
        w, h, b = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._L.oracle_scene_dims(self._h, C.byref(w), C.byref(h), C.byref(b))
        self.width, self.height, self.bucket_count = w.value, h.value, b.value

    def close(self):
        if self._h:
            self._L.oracle_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_camera(self, position, matrix):
        p = np.ascontiguousarray(position, dtype=np.float32)
        m = np.ascontiguousarray(matrix, dtype=np.float32).reshape(9)
        self._L.oracle_set_camera(self._h, _p(p), _p(m))

    def render(self, max_depth=5, threads=0, buffer=None, options=None):
        """Returns (rgb float32 [H,W,3], counters dict)."""
        o = options or make_options(max_depth)
        rgb = buffer if buffer is not None else np.zeros((self.height, self.width, 3), dtype=np.float32)
        cnt = np.zeros(16, dtype=np.uint64)
        self._L.oracle_render(self._h, C.byref(o), _p(rgb), _p(cnt), threads)
        return rgb, {k: int(cnt[i]) for i, k in enumerate(COUNTER_NAMES)}

    def trace(self, origin, direction, ray_type=0):
        o = np.ascontiguousarray(origin, dtype=np.float32)
        d = np.ascontiguousarray(direction, dtype=np.float32)
        out = np.zeros(16, dtype=np.float32)
        hit = self._L.oracle_trace(self._h, _p(o), _p(d), ray_type, _p(out))
        return bool(hit), out[:11].copy()

    def occluded(self, origin, direction, distance):
        o = np.ascontiguousarray(origin, dtype=np.float32)
        d = np.ascontiguousarray(direction, dtype=np.float32)
        return bool(self._L.oracle_occluded(self._h, _p(o), _p(d), float(distance)))

    def shoot(self, origin, direction, ray_type=0, depth=0, max_depth=5):
        o = np.ascontiguousarray(origin, dtype=np.float32)
        d = np.ascontiguousarray(direction, dtype=np.float32)
        out = np.zeros(3, dtype=np.float32)
        opt = make_options(max_depth)
        self._L.oracle_shoot(self._h, C.byref(opt), _p(o), _p(d), ray_type, depth, _p(out))
        return out

    def camera_ray(self, row, col):
        out = np.zeros(6, dtype=np.float32)
        self._L.oracle_camera_ray(self._h, row, col, _p(out))
        return out[:3].copy(), out[3:].copy()

    @property
    def mesh_count(self):
        return self._L.oracle_mesh_count(self._h)

    def tree(self, mesh_index=-1):
        """(boxes [n,6], links [n,4] = child0, child1, parent, index count, indexes [total])"""
        n = self._L.oracle_tree_node_count(self._h, mesh_index)
        total = self._L.oracle_tree_index_total(self._h, mesh_index)
        boxes = np.zeros((n, 6), dtype=np.float32)
        links = np.zeros((n, 4), dtype=np.uint32)
        idx = np.zeros(max(total, 1), dtype=np.uint32)
        self._L.oracle_tree_dump(self._h, mesh_index, _p(boxes), _p(links), _p(idx))
        return boxes, links, idx[:total]

    def mesh_normals(self, mesh_index):
        nv, nt = C.c_uint32(), C.c_uint32()
        self._L.oracle_mesh_sizes(self._h, mesh_index, C.byref(nv), C.byref(nt))
        fn = np.zeros((nt.value, 3), dtype=np.float32)
        vn = np.zeros((nv.value, 3), dtype=np.float32)
        self._L.oracle_mesh_normals(self._h, mesh_index, _p(fn), _p(vn))
        return fn, vn


def box_hit(lo, hi, origin, direction) -> bool:
    a = [np.ascontiguousarray(x, dtype=np.float32) for x in (lo, hi, origin, direction)]
    return bool(lib().oracle_box_hit(*[_p(x) for x in a]))


def bucket_grid(width, height, bucket_count):
    rects = np.zeros((65536, 4), dtype=np.uint32)
    n = lib().oracle_bucket_grid(width, height, bucket_count, _p(rects), 65536)
    return rects[:n].copy()


def quantize(rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    out = np.zeros(rgb.shape, dtype=np.uint16)
    lib().oracle_quantize(_p(rgb), rgb.size, _p(out))
    return out


def write_ppm(path, rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    h, w = rgb.shape[:2]
    if lib().oracle_write_ppm(path.encode(), _p(rgb), w, h) != 0:
        raise OSError("oracle: cannot write " + path)


def powf(x, y):
    return float(lib().oracle_powf(x, y))


def sinf(x):
    return float(lib().oracle_sinf(x))


def cosf(x):
    return float(lib().oracle_cosf(x))


def sincos_array(y):
    """(sinf, cosf) of a float32 array by the oracle's restatement of glibc's routines."""
    y = np.ascontiguousarray(y, dtype=np.float32)
    s, c = np.empty_like(y), np.empty_like(y)
    lib().oracle_sincos_array(_p(y), y.size, _p(s), _p(c))
    return s, c


def gi_array(a, b):
    """(mix(a, b), u(a, b)) elementwise for uint32 arrays: the GI generator of cpu_ref.c."""
    a = np.ascontiguousarray(a, dtype=np.uint32)
    b = np.ascontiguousarray(b, dtype=np.uint32)
    m, u = np.empty_like(a), np.empty(a.shape, dtype=np.float32)
    lib().oracle_gi_array(_p(a), _p(b), a.size, _p(m), _p(u))
    return m, u


def sincos_vs_libm(first=0.0, last=6.2831860, stride=1):
    """Compares the restated sinf / cosf with this machine's libm over every stride-th float of [first, last] (both >= 0):
    returns (values tested, sinf mismatches, cosf mismatches)."""
    lo = int(np.float32(first).view(np.uint32))
    hi = int(np.float32(last).view(np.uint32))
    out = np.zeros(3, dtype=np.uint64)
    lib().oracle_sincos_vs_libm(lo, hi, stride, _p(out))
    return int(out[0]), int(out[1]), int(out[2])


def gi_uniform(key, d):
    return float(lib().oracle_gi_uniform(key, d))


def gi_mix(a, b):
    return int(lib().oracle_gi_mix(a, b))


# ----------------------------------------------------------------------------- the REAL reference
def reference_available(textured=False) -> bool:
    return os.path.exists(REF_TEX if textured else REF_PLAIN)


def reference_render(blob: bytes, max_depth=5, mode="bvhpool", textured=None, ppm_path=None, repeat=1,
                     cpus=None, gi=None, all_frames=False):
    """Run the real reference (oracle/_ref/ref_render[_tex]) on a CRTS blob.
    Returns (rgb float32 [H,W,3], info dict with render_s/build_s/threads).  gi=(GI_SAMPLE_SIZE, RAYS_PER_PIXEL) selects
    the reference's GI mode (every render differs); all_frames=True returns all `repeat` frames as [repeat,H,W,3]."""
    if textured is None:
        textured = blob[80:84] != b"\x00\x00\x00\x00"  # n_textures field (after the 80-byte header)
    exe = REF_TEX if textured else REF_PLAIN
    if not os.path.exists(exe):
        raise FileNotFoundError(exe)
    with tempfile.TemporaryDirectory() as td:
        sp = os.path.join(td, "scene.crts")
        op = os.path.join(td, "out.f32")
        with open(sp, "wb") as f:
            f.write(blob)
        cmd = [exe, sp, op, "--depth", str(max_depth), "--mode", mode, "--repeat", str(repeat)]
        if ppm_path:
            cmd += ["--ppm", ppm_path]
        if gi:
            cmd += ["--gi", str(int(gi[0])), str(int(gi[1]))]
        if all_frames:
            cmd += ["--all-frames"]
        if cpus:
            cmd = ["taskset", "-c", cpus] + cmd
        r = subprocess.run(cmd, capture_output=True, text=True, check=True)
        info = json.loads(r.stdout.strip().splitlines()[-1])
        rgb = np.fromfile(op, dtype=np.float32)
        rgb = rgb.reshape(-1, info["height"], info["width"], 3) if all_frames else rgb.reshape(info["height"], info["width"], 3)
    return rgb, info
