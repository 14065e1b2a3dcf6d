"""GPU unit checks that do not need a scene."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def libm_pow5(x):
    libm = C.CDLL("libm.so.6")
    libm.powf.restype = C.c_float
    libm.powf.argtypes = [C.c_float, C.c_float]
    return np.array([libm.powf(float(v), 5.0) for v in x], dtype=np.float32)


def test_device_pow5_matches_libm(pkg):
    # Fresnel domain [0, 1] incl. subnormal results, negative tiny arguments, the underflow seam and specials
    rng = np.random.RandomState(11)
    x = np.concatenate([
        rng.randint(0, 0x3F800001, size=200_000, dtype=np.int64).astype(np.uint32).view(np.float32),
        np.arange(0x3F000000, 0x3F000000 + 50_000, dtype=np.uint32).view(np.float32),
        np.arange(int(np.float32(2.0 ** -30).view(np.uint32)) - 2000, int(np.float32(2.0 ** -30).view(np.uint32)) + 2000,
                  dtype=np.uint32).view(np.float32),
        np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 2.6e-8, -5.9604645e-08, 1.5, 300.0, 7131.6],
                 dtype=np.float32),
    ])
    got = pkg.test_pow5(x)
    want = libm_pow5(x)
    bad = (got.view(np.uint32) != want.view(np.uint32)) & ~(np.isnan(got) & np.isnan(want))
    assert not bad.any(), "x=%r device %r libm %r (%d mismatches)" % (x[bad][0], got[bad][0], want[bad][0], int(bad.sum()))


def test_abi_rejects_inconsistent_arguments(pkg, scenes):
    """The C ABI answers a malformed call with an error code and a message, never with a launch."""
    import torch
    scene = scenes.make("hw08", width=64, height=48, detail=0.2)
    tracer = pkg.Tracer(pkg.Scene(json_text=scenes.to_json(scene)))
    frame = torch.zeros(64 * 48 * 3, dtype=torch.float32, device="cuda:0")
    packed = torch.zeros(48 * 192, dtype=torch.float32, device="cuda:0")   # 8 x 6 tiles
    # a stride that cannot hold one part's tiles (the kernel would read past the buffer)
    with pytest.raises(pkg.CrtError) as e:
        tracer.unpack_tiles_device(packed.data_ptr(), 1, 47 * 192, frame.data_ptr())
    assert "part_stride_floats" in str(e.value)
    with pytest.raises(pkg.CrtError):
        tracer.unpack_tiles_device(packed.data_ptr(), 0, 48 * 192, frame.data_ptr())
    with pytest.raises(pkg.CrtError):
        tracer.render(options=pkg.make_options(3, use_gi=True, gi_sample_size=65))   # more GI samples per hit than the ABI takes
    with pytest.raises(pkg.CrtError):
        tracer.render(options=pkg.make_options(3, use_gi=True, counters=2))          # the executed-test tallies belong to the ray-stream kernels
    tracer.unpack_tiles_device(packed.data_ptr(), 1, 48 * 192, frame.data_ptr())   # the consistent call goes through
    torch.cuda.synchronize()
