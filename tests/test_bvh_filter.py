"""The candidate filter (csrc/crt_bvh.h, csrc/kernel_bvh.h) on the CPU: its hierarchy is built by the library's host code and its two
promises are checked against brute force by crt_bvh_selftest -- every triangle the reference's test accepts with a finite distance is
reached by the conservative walk; every triangle it accepts with an infinite or NaN distance in a leaf the ray's line passes is
reached by the miss check.  No GPU: the device kernels use the same expressions (tests/test_gpu_parity.py holds them to the oracle)."""
import ctypes as C

import numpy as np
import pytest

from helpers import small_case


def _selftest(pkg, scenes, scene, rays, primary=False, folder=""):
    if scene.get("textures") and folder:
        scenes.write_bitmaps(scene, folder)
    hs = pkg.Scene(json_text=scenes.to_json(scene), folder=folder)
    L = pkg.lib()
    L.crt_bvh_selftest.argtypes = [C.POINTER(pkg.SceneDesc), C.c_void_p, C.c_uint32, C.c_int, C.POINTER(C.c_uint64)]
    out = (C.c_uint64 * 8)()
    rays = np.ascontiguousarray(rays, dtype=np.float32)
    rc = L.crt_bvh_selftest(C.byref(hs.desc), rays.ctypes.data_as(C.c_void_p), len(rays), 1 if primary else 0, out)
    assert rc == 0
    return dict(zip(("rays", "finite_hits", "finite_missed", "other_hits", "other_missed", "nodes_1", "nodes_2", "errors"), [int(v) for v in out]))


def _random_rays(rng, n, lo, hi):
    o = rng.uniform(lo, hi, (n, 3))
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], axis=1).astype(np.float32)


@pytest.mark.parametrize("name", ["hw08", "hw11", "hw14", "hw12"])
def test_filter_reaches_every_accepted_triangle(pkg, scenes, name, tmp_path):
    scene, _, _ = small_case(scenes, name)
    rng = np.random.default_rng(11)
    rays = _random_rays(rng, 300, [-3.5, -2.0, -8.5], [3.5, 3.0, 1.5])       # in and around the room, any direction
    r = _selftest(pkg, scenes, scene, rays, folder=str(tmp_path))
    assert r["errors"] == 0
    assert r["finite_hits"] > (200 if name != "hw08" else 50) and r["finite_missed"] == 0     # (some origins lie outside the room, and hw08 is no room at all)
    assert r["other_missed"] == 0
    assert r["nodes_1"] < 0.2 * r["finite_hits"] * 400                         # a walk, not a sweep of the hierarchy


def test_rays_parallel_to_planes_are_seen_by_the_miss_check(pkg, scenes):
    """Directions with exact zeros: d . n is exactly 0 for the room's walls, the reference's test divides by it, and whatever it then
    accepts (infinite and NaN distances) must be among what the miss check reaches -- and zero components must not turn the
    conservative walk into a sweep."""
    scene, _, _ = small_case(scenes, "hw11")
    rng = np.random.default_rng(5)
    rays = []
    for axis in range(3):
        for _ in range(60):
            o = rng.uniform([-2.9, -1.4, -7.9], [2.9, 2.4, 0.9])
            d = rng.normal(size=3)
            d[axis] = 0.0
            d /= np.linalg.norm(d)
            rays.append(np.concatenate([o, d]))
        for sign in (1.0, -1.0):                                               # along an axis: two zero components
            d = np.zeros(3)
            d[axis] = sign
            rays.append(np.concatenate([rng.uniform([-2.9, -1.4, -7.9], [2.9, 2.4, 0.9]), d]))
    # origins ON the walls' planes, directions in them: the numerator is zero too (0 / 0)
    rays.append([-3.0, 0.0, -4.0, 0.0, 0.6, -0.8])
    rays.append([0.5, -1.5, -3.0, 0.6, 0.0, -0.8])
    r = _selftest(pkg, scenes, scene, np.array(rays))
    assert r["errors"] == 0 and r["finite_missed"] == 0 and r["other_missed"] == 0
    assert r["other_hits"] > 0                                                 # the case exists: some of these rays are "hit" at infinity
    assert r["nodes_2"] < len(rays) * 400


def test_primary_rays_of_the_benchmark_scene(pkg, scenes):
    scene = scenes.make("hw14", width=64, height=36, detail=0.2)               # ~8k triangles
    cam = np.array(scene["camera"]["position"], dtype=np.float32)
    rng = np.random.default_rng(3)
    d = np.stack([rng.uniform(-0.9, 0.9, 200), rng.uniform(-0.5, 0.5, 200), -np.ones(200)], axis=1)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([np.tile(cam, (200, 1)), d], axis=1)
    r = _selftest(pkg, scenes, scene, rays, primary=True)
    assert r["errors"] == 0 and r["finite_hits"] >= 200 and r["finite_missed"] == 0 and r["other_missed"] == 0
