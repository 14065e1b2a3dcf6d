"""The oracle (and the host-side writer) against the committed fixtures generated from the REAL reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from helpers import assert_same_floats, blob_to_scene, load_golden

CASES = ["hw07", "hw08", "hw11", "hw14", "hw12", "coverage", "uvwrap"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_reference_frame(oracle, name):
    g = load_golden(name)
    got, _ = oracle.OracleScene(g["blob"]).render(g["depth"])
    assert_same_floats(got, g["rgb"], name)


def test_uvwrap_fixture_has_coordinates_outside_the_unit_square():
    g = load_golden("uvwrap")
    uv = np.concatenate([np.asarray(o["uvs"])[:, :2].ravel() for o in blob_to_scene(g["blob"])["objects"] if "uvs" in o])
    assert uv.min() <= -1.0 and uv.max() > 1.0   # negative float -> unsigned (Texture.cpp:38-39), clamped bitmap index (:67-70)


def test_coverage_fixture_has_unrendered_pixels():
    # bucket_size 7 on 100x60: nx = 3, ny = 2 -> 33x30 rectangles, the 7th wraps; 1042 pixels stay (0,0,0)
    g = load_golden("coverage")
    assert int((g["rgb"].sum(axis=2) == 0).sum()) == 1042


def test_ppm_bytes(oracle, pkg, tmp_path):
    g = load_golden("hw07")
    p1, p2 = str(tmp_path / "o.ppm"), str(tmp_path / "h.ppm")
    oracle.write_ppm(p1, g["rgb"])
    pkg.export_ppm(p2, g["rgb"])  # the product's host-side writer (RayTracer::exportPPM replacement)
    assert open(p1, "rb").read() == g["ppm"]
    assert open(p2, "rb").read() == g["ppm"]


def test_quantiser_truncates(oracle):
    v = np.array([-1.0, 0.0, 0.0039, 0.00392157, 0.5, 0.999, 1.0, 7.0, 254.9999 / 255], dtype=np.float32)
    q = oracle.quantize(v)
    assert q.tolist() == [0, 0, 0, int(np.float32(0.00392157) * np.float32(255)), 127, 254, 255, 255,
                          int(np.float32(254.9999 / 255) * np.float32(255))]


def test_blob_roundtrip(scenes):
    # the helper that turns fixtures back into `.crtscene` scenes is lossless
    g = load_golden("hw12")
    scene = blob_to_scene(g["blob"])
    assert scenes.to_blob(scene) == g["blob"]
