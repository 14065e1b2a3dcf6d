"""The oracle against the real reference binary, on scenes that are NOT in tests/golden/.  Runs where
oracle/_ref/ exists (built from /root/reference in the development container; shipped to the GPU box as a
binary); skipped elsewhere."""
import numpy as np
import pytest

from helpers import assert_same_floats


def _have(oracle, textured):
    return oracle.reference_available(textured=textured)


@pytest.mark.parametrize("name,w,h,detail,depth", [
    ("hw08", 96, 64, 0.3, 1),
    ("hw11", 128, 72, 0.2, 5),
    ("hw14", 112, 63, 0.03, 8),
    ("hw12", 96, 54, 0.06, 3),
    # the benchmark scenes at FULL detail (the deep-tree paths: depth-25 cut-off, 8.6x leaf duplication, KDTree.cpp:10-46)
    ("hw14", 160, 90, 1.0, 8),     # 207,954 triangles
    ("hw12", 160, 90, 1.0, 8),     # 60,156 triangles, 1024x1024 bitmap
    ("hw11", 160, 90, 1.0, 8),
])
def test_restatement_is_bit_exact(oracle, scenes, name, w, h, detail, depth):
    kw = {"bitmap_size": 32} if name == "hw12" and detail < 1.0 else {}
    scene = scenes.make(name, width=w, height=h, detail=detail, **kw)
    if not _have(oracle, bool(scene.get("textures"))):
        pytest.skip("oracle/_ref not built here")
    blob = scenes.to_blob(scene)
    want, _ = oracle.reference_render(blob, max_depth=depth)
    got, _ = oracle.OracleScene(blob).render(depth)
    assert_same_floats(got, want, name)


def test_camera_change_matches_reference(oracle, scenes):
    if not _have(oracle, False):
        pytest.skip("oracle/_ref not built here")
    scene = scenes.make("hw11", width=96, height=54, detail=0.2)
    scene["camera"]["position"] = [0.35, 0.6, 0.2]
    c, s = np.float32(np.cos(0.2)), np.float32(np.sin(0.2))
    scene["camera"]["matrix"] = [c, 0, -s, 0, 1, 0, s, 0, c]
    blob = scenes.to_blob(scene)
    want, _ = oracle.reference_render(blob, max_depth=4)
    o = oracle.OracleScene(scenes.to_blob(scenes.make("hw11", width=96, height=54, detail=0.2)))
    o.set_camera(scene["camera"]["position"], scene["camera"]["matrix"])  # same scene, camera set afterwards
    got, _ = o.render(4)
    assert_same_floats(got, want, "camera")
