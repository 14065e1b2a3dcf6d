"""MI355X-native per-pixel hot path of the Chaos course ray tracer -- Python-side loader.

The product is the shared library `libcrt_hip.so` (HIP kernels for gfx950 + the C ABI of
include/crt_hip.h and include/crt_host.h + the C++ host mirror of the reference's
SceneParser / RayTracer).  This module only loads it with ctypes and wraps the handles; it holds
no rendering logic and has NO fallback: if the library is missing, or no GPU is present when a
tracer is created, it raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import scenes  # noqa: F401  (synthetic workload generators)

HERE = os.path.dirname(os.path.abspath(__file__))
# libcrt_hip.so is the product; libcrt_hip_test.so is the same objects plus the unit-test hooks (TEST_HOOK_SYMBOLS below), which the
# test suite loads (tests/conftest.py sets CRT_TEST_HOOKS; this package reads the variable, the library itself reads none)
LIB_PATH = os.path.join(HERE, "libcrt_hip_test.so" if os.environ.get("CRT_TEST_HOOKS") else "libcrt_hip.so")

CRT_OK, CRT_ERR_INVALID, CRT_ERR_NO_DEVICE, CRT_ERR_HIP, CRT_ERR_NOMEM, CRT_ERR_IO, CRT_ERR_PARSE = range(7)
OPT_NONE, OPT_REGIONS, OPT_BUCKETS_POOL, OPT_BUCKETS_QUEUE, OPT_AABB, OPT_BUCKETS_POOL_AABB, \
    OPT_BUCKETS_QUEUE_AABB, OPT_BVH, OPT_BVH_BUCKETS_POOL, OPT_BVH_BUCKETS_QUEUE = range(10)
LINK_END, LINK_LEAF, ENTRY_LAST = 0xFFFFFFFF, 0x80000000, 0x80000000


class CrtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("crt error %d: %s" % (code, message))
        self.code = code


class Node(C.Structure):
    _fields_ = [("lo", C.c_float * 3), ("miss", C.c_uint32), ("hi", C.c_float * 3), ("link", C.c_uint32)]


class Triangle(C.Structure):
    _fields_ = [("v0", C.c_float * 3), ("nx", C.c_float), ("v1", C.c_float * 3), ("ny", C.c_float),
                ("v2", C.c_float * 3), ("nz", C.c_float), ("plane", C.c_float), ("pad", C.c_uint32 * 3)]


class MeshRec(C.Structure):
    _fields_ = [("root", C.c_uint32), ("material", C.c_uint32), ("flags", C.c_uint32), ("pad", C.c_uint32)]


class MaterialRec(C.Structure):
    _fields_ = [("albedo", C.c_float * 3), ("ior", C.c_float), ("type", C.c_uint32), ("smooth", C.c_uint32),
                ("texture", C.c_int32), ("pad", C.c_uint32)]


class TextureRec(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("color_a", C.c_float * 3), ("color_b", C.c_float * 3), ("scalar", C.c_float),
                ("width", C.c_uint32), ("height", C.c_uint32), ("texel_offset", C.c_uint64)]


class LightRec(C.Structure):
    _fields_ = [("position", C.c_float * 3), ("intensity", C.c_uint32)]


class SceneDesc(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32), ("background", C.c_float * 3),
        ("nodes", C.POINTER(Node)), ("n_nodes", C.c_uint32), ("top_root", C.c_uint32),
        ("leaf_triangles", C.POINTER(C.c_uint32)), ("n_leaf_triangles", C.c_uint64),
        ("leaf_meshes", C.POINTER(C.c_uint32)), ("n_leaf_meshes", C.c_uint32),
        ("triangles", C.POINTER(Triangle)), ("n_triangles", C.c_uint32),
        ("triangle_vertices", C.POINTER(C.c_uint32)),
        ("vertex_normals", C.POINTER(C.c_float)), ("vertex_uvs", C.POINTER(C.c_float)), ("n_vertices", C.c_uint32),
        ("meshes", C.POINTER(MeshRec)), ("n_meshes", C.c_uint32),
        ("materials", C.POINTER(MaterialRec)), ("n_materials", C.c_uint32),
        ("textures", C.POINTER(TextureRec)), ("n_textures", C.c_uint32),
        ("texels", C.POINTER(C.c_uint8)), ("n_texels", C.c_uint64),
        ("lights", C.POINTER(LightRec)), ("n_lights", C.c_uint32),
    ]


class Options(C.Structure):
    _fields_ = [("max_depth", C.c_uint32), ("shadow_bias", C.c_float), ("reflection_bias", C.c_float),
                ("refraction_bias", C.c_float), ("use_gi", C.c_uint32), ("collect_counters", C.c_uint32),
                ("gi_sample_size", C.c_uint32), ("rays_per_pixel", C.c_uint32), ("monte_carlo_bias", C.c_float), ("gi_seed", C.c_uint32)]


class Rect(C.Structure):
    _fields_ = [("row", C.c_uint32), ("col", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("kernel_ms", C.c_double), ("total_ms", C.c_double), ("box_tests", C.c_uint64),
                ("tri_tests", C.c_uint64), ("leaf_index_reads", C.c_uint64), ("shaded_hits", C.c_uint64),
                ("light_evals", C.c_uint64), ("texel_fetches", C.c_uint64), ("primary_rays", C.c_uint64),
                ("secondary_rays", C.c_uint64), ("shadow_rays", C.c_uint64), ("pixels", C.c_uint64),
                ("counters_valid", C.c_uint32), ("fallback_frames", C.c_uint32), ("queue_bytes", C.c_uint64),
                ("queue_regrows", C.c_uint64)]

    def counters(self):
        return {k: int(getattr(self, k)) for k in ("box_tests", "tri_tests", "leaf_index_reads", "shaded_hits",
                                                   "light_evals", "texel_fetches", "primary_rays", "secondary_rays",
                                                   "shadow_rays")}


MODE_STREAM, MODE_LANES = range(2)


class Tuning(C.Structure):
    """crt_tuning (include/crt_hip.h): kernel selection and sizing; no setting changes a pixel."""
    _fields_ = [(n, C.c_uint32) for n in (
        "size", "mode", "step_budget", "shadow_budget", "level0_budget", "heavy_level", "side_blocks",
        "node_cap", "ray_cap", "shadow_cap", "bvh", "level_queue", "fetch_chunk")]


def make_tuning(**fields):
    """Default tuning (crt_tuning_defaults) with the given fields replaced, e.g. make_tuning(mode=MODE_LANES)."""
    t = Tuning()
    lib().crt_tuning_defaults(C.byref(t))
    for k, v in fields.items():
        if k not in dict(Tuning._fields_) or k in ("size", "reserved"):
            raise KeyError("crt_tuning has no field %r" % k)
        setattr(t, k, int(v))
    return t


def tuning_from_string(text):
    """'quad=0 heavy_level=0' -> make_tuning(quad=0, heavy_level=0); '' -> None.  Used by the development tools
    (tools/*.py read it from their command line or from the CRT_TUNING variable); the library itself reads no
    environment variables."""
    text = (text or "").strip()
    if not text:
        return None
    return make_tuning(**{k: int(v, 0) for k, v in (kv.split("=") for kv in text.split())})


# every symbol include/crt_hip.h and include/crt_host.h declare; the test hooks are exported by libcrt_hip_test.so only
TEST_HOOK_SYMBOLS = ["crt_bvh_selftest", "crt_test_pow5", "crt_test_gi", "crt_debug_multi_force_staged", "crt_debug_multi_fail_next_alloc"]
DEVICE_SYMBOLS = ["crt_bvh_selftest", "crt_tuning_defaults", "crt_create_tuned", "crt_create", "crt_set_camera", "crt_render", "crt_render_tiles_device", "crt_packed_tile_count",
                  "crt_unpack_tiles_device", "crt_quantize_device", "crt_read_quantized", "crt_kernel_elapsed_ms", "crt_kernel_times_ms",
                  "crt_get_stats", "crt_get_kernel_counters", "crt_synchronize", "crt_destroy", "crt_last_error", "crt_device_count", "crt_test_pow5", "crt_test_gi",
                  "crt_describe_kernels", "crt_debug_stream_counts", "crt_get_executed_counters", "crt_get_executed_plan_tests",
                  "crt_render_async", "crt_wait", "crt_alloc_pinned", "crt_free_pinned",
                  "crt_build_tree_device", "crt_built_tree_node_count", "crt_built_tree_index_total", "crt_built_tree_boxes",
                  "crt_built_tree_links", "crt_built_tree_indexes", "crt_built_tree_free", "crt_build_last_error",
                  "crt_multi_create", "crt_multi_set_camera", "crt_multi_render", "crt_multi_read_quantized", "crt_multi_get_stats",
                  "crt_multi_device_count", "crt_multi_context", "crt_multi_last_error", "crt_multi_destroy",
                  "crt_multi_staged_parts", "crt_multi_peer_note", "crt_debug_multi_force_staged", "crt_debug_multi_fail_next_alloc"]
HOST_SYMBOLS = ["crt_host_tracer_note", "crt_host_scene_parse_file", "crt_host_scene_parse_text", "crt_host_scene_parse_text_ex",
                "crt_host_scene_build_seconds", "crt_host_scene_free", "crt_host_scene_desc",
                "crt_host_scene_settings", "crt_host_scene_camera", "crt_host_scene_mesh_count",
                "crt_host_tree_node_count", "crt_host_tree_index_total", "crt_host_tree_dump", "crt_host_mesh_sizes",
                "crt_host_mesh_normals", "crt_host_bucket_rects", "crt_host_camera_apply", "crt_host_tracer_create", "crt_host_tracer_create_tuned",
                "crt_host_tracer_create_multi", "crt_host_tracer_stats",
                "crt_host_tracer_free", "crt_host_tracer_set_camera", "crt_host_tracer_render", "crt_host_tracer_ctx", "crt_host_tracer_multi",
                "crt_host_export_ppm", "crt_host_last_error"]

_lib = None


def lib():
    """Load libcrt_hip.so (built by `make -C course-assignment-danielhalachev_amd` / __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("the library is not built (%s): run __graft_entry__.build(); there is no fallback path"
                          % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int
    L.crt_create.argtypes = [C.POINTER(SceneDesc), i32, C.POINTER(vp)]
    L.crt_set_camera.argtypes = [vp, vp, vp]
    L.crt_render.argtypes = [vp, C.POINTER(Options), C.POINTER(Rect), u32, vp]
    L.crt_render_tiles_device.argtypes = [vp, C.POINTER(Options), u32, u32, vp, vp]
    L.crt_render_async.argtypes = [vp, C.POINTER(Options), C.POINTER(Rect), u32, vp, vp]
    L.crt_wait.argtypes = [vp]
    L.crt_alloc_pinned.restype = vp
    L.crt_alloc_pinned.argtypes = [C.c_size_t]
    L.crt_free_pinned.restype = None
    L.crt_free_pinned.argtypes = [vp]
    L.crt_packed_tile_count.restype = u32
    L.crt_packed_tile_count.argtypes = [vp, u32, u32]
    L.crt_unpack_tiles_device.argtypes = [vp, vp, u32, C.c_uint64, vp, vp]
    L.crt_quantize_device.argtypes = [vp, vp, C.c_uint64, vp, vp]
    L.crt_read_quantized.argtypes = [vp, vp]
    L.crt_kernel_elapsed_ms.argtypes = [vp, C.POINTER(C.c_double)]
    L.crt_kernel_times_ms.argtypes = [vp, C.POINTER(C.c_double), u32, C.POINTER(u32)]
    L.crt_get_stats.argtypes = [vp, C.POINTER(Stats)]
    L.crt_get_kernel_counters.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.crt_synchronize.argtypes = [vp]
    L.crt_destroy.argtypes = [vp]
    L.crt_destroy.restype = None
    L.crt_last_error.restype = C.c_char_p
    L.crt_last_error.argtypes = [vp]
    L.crt_host_scene_parse_file.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(vp)]
    L.crt_host_scene_parse_text.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.POINTER(vp)]
    L.crt_host_scene_parse_text_ex.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, i32, C.POINTER(vp)]
    L.crt_host_scene_build_seconds.restype = C.c_double
    L.crt_host_scene_build_seconds.argtypes = [vp]
    L.crt_host_scene_free.argtypes = [vp]
    L.crt_host_scene_free.restype = None
    L.crt_host_scene_desc.restype = C.POINTER(SceneDesc)
    L.crt_host_scene_desc.argtypes = [vp]
    L.crt_host_scene_settings.argtypes = [vp, C.POINTER(u32), C.POINTER(u32), C.POINTER(u32)]
    L.crt_host_scene_settings.restype = None
    L.crt_host_scene_camera.argtypes = [vp, vp, vp]
    L.crt_host_scene_camera.restype = None
    L.crt_host_scene_mesh_count.restype = u32
    L.crt_host_scene_mesh_count.argtypes = [vp]
    L.crt_host_tree_node_count.restype = u32
    L.crt_host_tree_node_count.argtypes = [vp, i32]
    L.crt_host_tree_index_total.restype = C.c_uint64
    L.crt_host_tree_index_total.argtypes = [vp, i32]
    L.crt_host_tree_dump.argtypes = [vp, i32, vp, vp, vp]
    L.crt_host_tree_dump.restype = None
    L.crt_host_mesh_sizes.argtypes = [vp, u32, C.POINTER(u32), C.POINTER(u32)]
    L.crt_host_mesh_sizes.restype = None
    L.crt_host_mesh_normals.argtypes = [vp, u32, vp, vp]
    L.crt_host_mesh_normals.restype = None
    L.crt_host_bucket_rects.restype = u32
    L.crt_host_bucket_rects.argtypes = [u32, u32, u32, i32, u32, C.POINTER(Rect), u32]
    L.crt_host_camera_apply.argtypes = [vp, vp, i32, vp]
    L.crt_host_tracer_create.argtypes = [vp, i32, C.POINTER(vp)]
    L.crt_host_tracer_create_tuned.argtypes = [vp, i32, C.POINTER(Tuning), C.POINTER(vp)]
    L.crt_host_tracer_create_multi.argtypes = [vp, C.POINTER(C.c_int), u32, C.POINTER(Tuning), C.POINTER(vp)]
    L.crt_host_tracer_stats.argtypes = [vp, C.POINTER(Stats)]
    L.crt_tuning_defaults.argtypes = [C.POINTER(Tuning)]
    L.crt_tuning_defaults.restype = None
    L.crt_create_tuned.argtypes = [C.POINTER(SceneDesc), i32, C.POINTER(Tuning), C.POINTER(vp)]
    L.crt_host_tracer_free.argtypes = [vp]
    L.crt_host_tracer_free.restype = None
    L.crt_host_tracer_set_camera.argtypes = [vp, vp, vp]
    L.crt_host_tracer_render.argtypes = [vp, C.c_char_p, i32, C.POINTER(Options), vp]
    L.crt_host_tracer_note.argtypes = [vp]
    L.crt_host_tracer_note.restype = C.c_char_p
    L.crt_host_tracer_ctx.restype = vp
    L.crt_host_tracer_ctx.argtypes = [vp]
    L.crt_host_export_ppm.argtypes = [C.c_char_p, vp, u32, u32]
    L.crt_host_last_error.restype = C.c_char_p
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def make_options(max_depth=5, shadow_bias=1e-4, reflection_bias=1e-4, refraction_bias=1e-4, counters=False, use_gi=False,
                 gi_sample_size=2, rays_per_pixel=1, monte_carlo_bias=1e-4, gi_seed=0):
    # counters: False / True (the counting build, reference semantics) / 2 (production kernels tallying executed tests)
    # use_gi: the reference's GI / multi-sample mode (RayTracer.h:27-30) with the counter-based generator of csrc/gi_random.h
    return Options(max_depth, shadow_bias, reflection_bias, refraction_bias, int(bool(use_gi)), int(counters), gi_sample_size,
                   rays_per_pixel, monte_carlo_bias, gi_seed)


def _host_check(rc):
    if rc != CRT_OK:
        raise CrtError(rc, lib().crt_host_last_error().decode(errors="replace"))


class Scene:
    """A parsed `.crtscene` with its tree built and flattened (host side only; no GPU needed)."""

    def __init__(self, json_text=None, path=None, folder="", build_device=-1):
        """build_device >= 0: the big meshes' trees are built on that GPU (same trees, node for node)."""
        L = lib()
        h = C.c_void_p()
        if json_text is not None:
            data = json_text.encode() if isinstance(json_text, str) else json_text
            _host_check(L.crt_host_scene_parse_text_ex(data, len(data), folder.encode(), build_device, C.byref(h)))
        else:
            _host_check(L.crt_host_scene_parse_file(path.encode(), folder.encode(), C.byref(h)))
        self._h = h
        w, hh, b = C.c_uint32(), C.c_uint32(), C.c_uint32()
        L.crt_host_scene_settings(h, C.byref(w), C.byref(hh), C.byref(b))
        self.width, self.height, self.bucket_size = w.value, hh.value, b.value
        self.build_seconds = L.crt_host_scene_build_seconds(h)

    def close(self):
        if getattr(self, "_h", None):
            lib().crt_host_scene_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def desc(self):
        return lib().crt_host_scene_desc(self._h).contents

    @property
    def mesh_count(self):
        return lib().crt_host_scene_mesh_count(self._h)

    def camera(self):
        pos = np.zeros(3, dtype=np.float32)
        mat = np.zeros(9, dtype=np.float32)
        lib().crt_host_scene_camera(self._h, _p(pos), _p(mat))
        return pos, mat

    def tree(self, mesh=-1):
        """Tree in the reference's node numbering: (boxes [n,6], links [n,4], indexes [total])."""
        L = lib()
        n = L.crt_host_tree_node_count(self._h, mesh)
        total = L.crt_host_tree_index_total(self._h, mesh)
        boxes = np.zeros((n, 6), dtype=np.float32)
        links = np.zeros((n, 4), dtype=np.uint32)
        idx = np.zeros(max(total, 1), dtype=np.uint32)
        L.crt_host_tree_dump(self._h, mesh, _p(boxes), _p(links), _p(idx))
        return boxes, links, idx[:total]

    def mesh_normals(self, mesh):
        nv, nt = C.c_uint32(), C.c_uint32()
        lib().crt_host_mesh_sizes(self._h, mesh, C.byref(nv), C.byref(nt))
        fn = np.zeros((nt.value, 3), dtype=np.float32)
        vn = np.zeros((nv.value, 3), dtype=np.float32)
        lib().crt_host_mesh_normals(self._h, mesh, _p(fn), _p(vn))
        return fn, vn

    def flat_nodes(self):
        d = self.desc
        return np.ctypeslib.as_array(C.cast(d.nodes, C.POINTER(C.c_uint32)), shape=(d.n_nodes, 8)).copy()

    def flat_leaf_triangles(self):
        d = self.desc
        if d.n_leaf_triangles == 0:
            return np.zeros(0, dtype=np.uint32)
        return np.ctypeslib.as_array(d.leaf_triangles, shape=(d.n_leaf_triangles,)).copy()

    def flat_leaf_meshes(self):
        d = self.desc
        if d.n_leaf_meshes == 0:
            return np.zeros(0, dtype=np.uint32)
        return np.ctypeslib.as_array(d.leaf_meshes, shape=(d.n_leaf_meshes,)).copy()


def bucket_rects(width, height, bucket_size, optimization=OPT_BVH_BUCKETS_POOL, hardware_concurrency=8):
    out = (Rect * 65536)()
    n = lib().crt_host_bucket_rects(width, height, bucket_size, optimization, hardware_concurrency, out, 65536)
    return np.array([[r.row, r.col, r.width, r.height] for r in out[:min(n, 65536)]], dtype=np.uint32).reshape(-1, 4)


def camera_apply(position, matrix, op, v):
    pos = np.ascontiguousarray(position, dtype=np.float32).copy()
    mat = np.ascontiguousarray(matrix, dtype=np.float32).reshape(9).copy()
    vv = np.zeros(3, dtype=np.float32)
    vv[:len(np.atleast_1d(v))] = np.atleast_1d(v)
    _host_check(lib().crt_host_camera_apply(_p(pos), _p(mat), {"truck": 0, "pan": 1, "tilt": 2, "roll": 3}[op], _p(vv)))
    return pos, mat


def export_ppm(path, rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    _host_check(lib().crt_host_export_ppm(path.encode(), _p(rgb), rgb.shape[1], rgb.shape[0]))


class Tracer:
    """crt::RayTracer on one GPU: scene + tree resident in HBM, re-renderable with a new camera."""

    def __init__(self, scene: Scene, device=0, tuning: Tuning = None, devices=None):
        """devices: a list of GPU indices -> the frame's tiles are dealt over them (crt_multi), devices[0] gathers."""
        L = lib()
        self.scene = scene
        h = C.c_void_p()
        tp = C.byref(tuning) if tuning is not None else None
        if devices:
            arr = (C.c_int * len(devices))(*devices)
            _host_check(L.crt_host_tracer_create_multi(scene._h, arr, len(devices), tp, C.byref(h)))
        else:
            _host_check(L.crt_host_tracer_create_tuned(scene._h, device, tp, C.byref(h)))
        self.devices = list(devices) if devices else [device]
        self._h = h
        self.ctx = C.c_void_p(L.crt_host_tracer_ctx(h))
        self.width, self.height = scene.width, scene.height

    def close(self):
        if getattr(self, "_h", None):
            lib().crt_host_tracer_free(self._h)
            self._h = None

    # ---- several devices: which parts reach devices[0] over xGMI, which are staged through pinned host memory
    def _multi(self):
        L = lib()
        L.crt_host_tracer_multi.restype = C.c_void_p
        L.crt_host_tracer_multi.argtypes = [C.c_void_p]
        m = L.crt_host_tracer_multi(self._h)
        if not m:
            raise RuntimeError("not a multi-device tracer")
        return C.c_void_p(m)

    def staged_parts(self):
        L = lib()
        L.crt_multi_staged_parts.restype = C.c_uint32
        L.crt_multi_staged_parts.argtypes = [C.c_void_p]
        return int(L.crt_multi_staged_parts(self._multi()))

    def peer_note(self):
        L = lib()
        L.crt_multi_peer_note.restype = C.c_char_p
        L.crt_multi_peer_note.argtypes = [C.c_void_p]
        return L.crt_multi_peer_note(self._multi()).decode()

    def fail_next_alloc(self):
        """Tests: the next re-partition of a multi-device tracer stops half-way with CRT_ERR_NOMEM."""
        L = lib()
        L.crt_debug_multi_fail_next_alloc.argtypes = [C.c_void_p]
        self._check(L.crt_debug_multi_fail_next_alloc(self._multi()))

    def force_staged(self, on=True):
        """Tests: every part but the first copies its tiles through pinned host memory, as if no device had peer access."""
        L = lib()
        L.crt_debug_multi_force_staged.argtypes = [C.c_void_p, C.c_int]
        self._check(L.crt_debug_multi_force_staged(self._multi(), 1 if on else 0))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_camera(self, position, matrix):
        p = np.ascontiguousarray(position, dtype=np.float32)
        m = np.ascontiguousarray(matrix, dtype=np.float32).reshape(9)
        _host_check(lib().crt_host_tracer_set_camera(self._h, _p(p), _p(m)))

    def render(self, max_depth=5, optimization=OPT_BVH_BUCKETS_POOL, ppm_path=None, counters=False, out=None,
               options=None):
        """RayTracer::render: returns the H x W x 3 float32 colour buffer (and writes a PPM if asked)."""
        o = options or make_options(max_depth, counters=counters)
        rgb = out if out is not None else np.zeros((self.height, self.width, 3), dtype=np.float32)
        _host_check(lib().crt_host_tracer_render(self._h, ppm_path.encode() if ppm_path else None, optimization,
                                                 C.byref(o), _p(rgb)))
        return rgb

    def note(self) -> str:
        """What the last render did differently from the reference ("" when nothing): a non-tree RenderOptimization renders with the
        tree modes' semantics (crt_host.h: crt_host_tracer_note)."""
        return (lib().crt_host_tracer_note(self._h) or b"").decode()

    def synchronize(self):
        """crt_synchronize on the first context: waits for everything enqueued on its device; afterwards stats() reflects the
        last frame (fallback_frames included)."""
        self._check(lib().crt_synchronize(self.ctx))

    def stats(self) -> Stats:
        s = Stats()
        lib().crt_host_tracer_stats(self._h, C.byref(s))
        return s

    def executed_counters(self):
        """{box_tests, tri_tests} the production kernels executed in the last render made with counters=2."""
        a = (C.c_uint64 * 4)()
        self._check(lib().crt_get_executed_counters(self.ctx, a))
        b = (C.c_uint64 * 2)()
        L = lib()
        L.crt_get_executed_plan_tests.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        self._check(L.crt_get_executed_plan_tests(self.ctx, b))
        return {"box_tests": int(a[0]), "tri_tests": int(a[1]), "shadow_pass0_box_tests": int(a[2]), "shadow_pass0_tri_tests": int(a[3]),
                "plan_tests": int(b[0]), "shadow_pass0_plan_tests": int(b[1])}

    def kernel_counters(self):
        """(recursion levels' counters, bulk shadow pass's counters) of the last counted render, as dicts."""
        names = ("box_tests", "tri_tests", "leaf_index_reads", "shaded_hits", "light_evals", "texel_fetches",
                 "primary_rays", "secondary_rays", "shadow_rays")
        a = (C.c_uint64 * 9)()
        b = (C.c_uint64 * 9)()
        self._check(lib().crt_get_kernel_counters(self.ctx, a, b))
        return ({k: int(a[i]) for i, k in enumerate(names)}, {k: int(b[i]) for i, k in enumerate(names)})

    def _check(self, rc):
        if rc != CRT_OK:
            raise CrtError(rc, lib().crt_last_error(self.ctx).decode(errors="replace"))

    # ---- device-resident API (multi-GPU tile partition, bench)
    def packed_tile_count(self, first, stride):
        return lib().crt_packed_tile_count(self.ctx, first, stride)

    def render_tiles_device(self, options, first, stride, d_packed_ptr, stream_ptr=None):
        self._check(lib().crt_render_tiles_device(self.ctx, C.byref(options), first, stride, C.c_void_p(d_packed_ptr),
                                                  C.c_void_p(stream_ptr or 0)))

    def unpack_tiles_device(self, d_packed_all_ptr, n_parts, part_stride_floats, d_frame_ptr, stream_ptr=None):
        self._check(lib().crt_unpack_tiles_device(self.ctx, C.c_void_p(d_packed_all_ptr), n_parts, part_stride_floats,
                                                  C.c_void_p(d_frame_ptr), C.c_void_p(stream_ptr or 0)))

    def quantize_device(self, d_rgb_ptr, n_values, d_out_ptr, stream_ptr=None):
        self._check(lib().crt_quantize_device(self.ctx, C.c_void_p(d_rgb_ptr), n_values, C.c_void_p(d_out_ptr),
                                              C.c_void_p(stream_ptr or 0)))

    def kernel_elapsed_ms(self):
        ms = C.c_double()
        self._check(lib().crt_kernel_elapsed_ms(self.ctx, C.byref(ms)))
        return ms.value

    def kernel_times_ms(self, max_count=64):
        """[(total, recursion levels, shadow pass 0 [overlapped], shadow pass 1 + heavy, resolve)] in ms of the
        most recent renders, oldest first."""
        a = (C.c_double * (5 * max_count))()
        n = C.c_uint32()
        self._check(lib().crt_kernel_times_ms(self.ctx, a, max_count, C.byref(n)))
        return [tuple(a[5 * i + j] for j in range(5)) for i in range(n.value)]

    # ---- a frame without waiting for it (crt_render_async / crt_wait), through the device-level ABI of the first context
    def render_async(self, options, rgb=None, rgb8=None, optimization=OPT_BVH_BUCKETS_POOL):
        """Enqueue a frame (coverage: the reference's bucket rectangles); rgb / rgb8 are numpy arrays the copies land in
        once wait() has returned."""
        r = bucket_rects(self.width, self.height, self.scene.bucket_size, optimization)
        rects = (Rect * len(r))(*[Rect(*map(int, x)) for x in r])
        self._check(lib().crt_render_async(self.ctx, C.byref(options), rects, len(r), _p(rgb) if rgb is not None else None,
                                           _p(rgb8) if rgb8 is not None else None))

    def wait(self):
        self._check(lib().crt_wait(self.ctx))

    def kernels(self):
        """{'level0': ..., 'shadow0': ..., 'levels': ...}: the kernels a production frame runs (names as rocprofv3 prints them)."""
        buf = C.create_string_buffer(512)
        L = lib()
        L.crt_describe_kernels.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        self._check(L.crt_describe_kernels(self.ctx, buf, 512))
        return dict(kv.split("=", 1) for kv in buf.value.decode().split(";"))

    def stream_counts(self):
        """Diagnostics: the ray-stream pass's counter block of the last frame (SC_* layout of csrc/kernel_stream.h)."""
        out = (C.c_uint32 * 512)()
        L = lib()
        L.crt_debug_stream_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32]
        self._check(L.crt_debug_stream_counts(self.ctx, out, 512))
        return np.array(out[:], dtype=np.uint32)

    def read_quantized(self):
        out = np.zeros((self.height, self.width, 3), dtype=np.uint8)
        self._check(lib().crt_read_quantized(self.ctx, _p(out)))
        return out


def test_pow5(x, device=0):
    """Device build of the restated glibc powf(x, 5) on a float32 array (unit-test hook)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.empty_like(x)
    L = lib()
    L.crt_test_pow5.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_uint64]
    rc = L.crt_test_pow5(device, _p(x), _p(out), x.size)
    if rc != CRT_OK:
        raise CrtError(rc, "crt_test_pow5 failed")
    return out


def csrc_sha256():
    """SHA-256 over the kernel sources (csrc/*, sorted by name): profiles/ records it at collection time and bench.py compares it
    with the running tree's, so that counters of other kernels are never passed off as this build's."""
    import hashlib
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".h", ".hip", ".cpp")):
            h.update(name.encode())
            with open(os.path.join(d, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()


def device_count():
    return lib().crt_device_count()
