"""The restated glibc powf(x, 5) (csrc/glibc_powf.h) against libm's powf itself, on the CPU.
The same header is compiled into the HIP kernels; tests/test_gpu_units.py checks the device build."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def shim():
    out = os.path.join(tempfile.gettempdir(), "crt_powf_shim_%d.so" % os.getuid())
    src = os.path.join(HERE, "powf_shim.cpp")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off", src, "-o", out, "-lm"])
    lib = C.CDLL(out)
    for f in (lib.shim_pow5, lib.libm_pow5):
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    return lib


def both(shim, x):
    x = np.ascontiguousarray(x, dtype=np.float32)
    a = np.empty_like(x)
    b = np.empty_like(x)
    shim.shim_pow5(x.ctypes.data, a.ctypes.data, x.size)
    shim.libm_pow5(x.ctypes.data, b.ctypes.data, x.size)
    return a, b


def assert_bits_equal(a, b, x):
    ua, ub = a.view(np.uint32), b.view(np.uint32)
    bad = (ua != ub) & ~(np.isnan(a) & np.isnan(b))
    assert not bad.any(), "first mismatch at x=%r: restated %r libm %r (%d mismatches)" % (
        x[bad][0], a[bad][0], b[bad][0], int(bad.sum()))


def test_fresnel_domain_dense(shim):
    # the Fresnel argument is 1 - cos(alpha) in [0, 1]: every binary32 exponent, 4M random mantissas + a dense run
    rng = np.random.RandomState(5)
    bits = rng.randint(0, 0x3F800001, size=4_000_000, dtype=np.int64).astype(np.uint32)
    x = bits.view(np.float32)
    a, b = both(shim, x)
    assert_bits_equal(a, b, x)
    run = np.arange(0x3F000000, 0x3F000000 + 2_000_000, dtype=np.uint32).view(np.float32)  # 0.5 .. 0.56, consecutive floats
    a, b = both(shim, run)
    assert_bits_equal(a, b, run)


def test_whole_range_and_specials(shim):
    rng = np.random.RandomState(6)
    x = rng.randint(0, 2**32, size=2_000_000, dtype=np.int64).astype(np.uint32).view(np.float32)  # any bit pattern
    a, b = both(shim, x)
    assert_bits_equal(a, b, x)
    specials = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 1.1754944e-38, 2.6e-8,
                         9.3e-10, 9.4e-10, 1.0e-9, 1.1e-9, 3.4e38, 170.0, 180.0, 7131.5, 7131.6, -2.5, 0.999999940395],
                        dtype=np.float32)
    a, b = both(shim, specials)
    assert_bits_equal(a, b, specials)
    # the under/overflow seams: consecutive floats around 2^-30 (5*log2 x = -150) and 2^25.6 (overflow)
    for centre in (2.0 ** -30, 2.0 ** -29.8, 2.0 ** -25.2, 2.0 ** 25.6):
        c = int(np.float32(centre).view(np.uint32))
        seam = np.arange(c - 200_000, c + 200_000, dtype=np.uint32).view(np.float32)
        a, b = both(shim, seam)
        assert_bits_equal(a, b, seam)
