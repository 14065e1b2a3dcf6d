"""tests/golden/fullsize.json on the CPU: the scene generators still produce the scenes the reference rendered, and the oracle
reproduces the reference's whole full-size frame where that takes seconds (HW07 640x480, HW08 1280x720)."""
import hashlib
import importlib.util
import json
import os

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("make_fullsize", os.path.join(HERE, "golden", "make_fullsize.py"))
make_fullsize = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(make_fullsize)

with open(os.path.join(HERE, "golden", "fullsize.json")) as _f:
    FULL = json.load(_f)


@pytest.mark.parametrize("name", ["hw07", "hw08", "hw11", "hw14", "hw12"])
def test_generators_produce_the_scenes_the_reference_rendered(scenes, name):
    scene = scenes.make(name)
    assert hashlib.sha256(scenes.to_blob(scene)).hexdigest() == FULL[name]["scene_sha256"]
    assert scenes.triangle_count(scene) == FULL[name]["triangles"]


@pytest.mark.parametrize("name", ["hw07", "hw08"])
def test_oracle_reproduces_the_reference_full_size_frame(oracle, scenes, name):
    want = FULL[name]
    got, _ = oracle.OracleScene(scenes.to_blob(scenes.make(name))).render(want["depth"])
    d = make_fullsize.digests(got)
    assert d["frame_sha256"] == want["frame_sha256"]
    assert d["quant_sha256"] == want["quant_sha256"]
