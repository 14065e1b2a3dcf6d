"""Every BASELINE.json configuration at its FULL size and full detail, the GPU's whole frame against the REAL reference's whole
frame (tests/golden/fullsize.json: digests of what oracle/_ref rendered in the development container, make_fullsize.py) --
float frame bit for bit, quantised pixels, and the bytes of the PPM file (RayTracer.cpp:204-298, 540-552)."""
import hashlib
import importlib.util
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
_spec = importlib.util.spec_from_file_location("make_fullsize", os.path.join(HERE, "golden", "make_fullsize.py"))
make_fullsize = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(make_fullsize)

with open(os.path.join(HERE, "golden", "fullsize.json")) as _f:
    FULL = json.load(_f)


@pytest.mark.parametrize("name", ["hw07", "hw08", "hw11", "hw14", "hw12"])
def test_whole_frame_equals_the_reference_frame(pkg, scenes, name, tmp_path):
    want = FULL[name]
    scene = scenes.make(name)
    assert hashlib.sha256(scenes.to_blob(scene)).hexdigest() == want["scene_sha256"]   # the very scene the reference rendered
    folder = ""
    if scene.get("textures"):
        folder = str(tmp_path)
        scenes.write_bitmaps(scene, folder)
    tracer = pkg.Tracer(pkg.Scene(json_text=scenes.to_json(scene), folder=folder))
    assert (tracer.width, tracer.height) == (want["width"], want["height"])
    ppm = str(tmp_path / "frame.ppm")
    got = tracer.render(max_depth=want["depth"], ppm_path=ppm)
    d = make_fullsize.digests(got)
    bad = [k for k in range(len(d["band_sha256"])) if d["band_sha256"][k] != want["band_sha256"][k]]
    assert not bad, "%s: the frame differs from the reference's in row bands %s of %d" % (name, bad, len(d["band_sha256"]))
    assert d["frame_sha256"] == want["frame_sha256"]
    assert d["quant_sha256"] == want["quant_sha256"]
    q = tracer.read_quantized()                                            # PPMColor on the device
    assert hashlib.sha256(np.ascontiguousarray(q, dtype=np.uint8).tobytes()).hexdigest() == want["quant_sha256"]
    h = hashlib.sha256()
    with open(ppm, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 24), b""):
            h.update(chunk)
    assert os.path.getsize(ppm) == want["ppm_bytes"]
    assert h.hexdigest() == want["ppm_sha256"]                             # the file the reference's own exportPPM wrote
    os.remove(ppm)
