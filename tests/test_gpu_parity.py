"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on the same inputs.
Bar: bit-exact floats (stricter than the 1e-5 north-star tolerance) and identical work counters."""
import numpy as np
import pytest

from helpers import assert_same_floats, small_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", ["hw07", "hw08", "hw11", "hw14", "hw12"])
def test_frame_matches_oracle(pkg, scenes, oracle, name, tmp_path):
    scene, depth, folder = small_case(scenes, name, tmp_path)
    hs = pkg.Scene(json_text=scenes.to_json(scene), folder=folder)
    tracer = pkg.Tracer(hs)
    got = tracer.render(max_depth=depth, counters=True)
    stats = tracer.stats()
    want, counters = oracle.OracleScene(scenes.to_blob(scene)).render(depth)
    assert_same_floats(got, want, name)
    assert stats.counters_valid == 1
    assert stats.counters() == counters
    # the non-counting build of the kernel must give the same pixels
    again = tracer.render(max_depth=depth)
    assert_same_floats(again, want, name + " (plain kernel)")
    # quantised bytes (PPMColor) from the device
    q = tracer.read_quantized()
    assert np.array_equal(q.astype(np.uint16), oracle.quantize(want))
