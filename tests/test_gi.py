"""The GI / multi-sample mode (RayTracer.cpp:90-104, 331-354).

The reference seeds that mode's generator from clock() ^ thread id, so none of its frames can be pinned; what is defined is the
DISTRIBUTION of its frames.  The tests therefore come in three layers:
  1. the arithmetic the mode adds -- glibc's sinf / cosf, restated in oracle/cpu_ref.c and (independently) in
     csrc/glibc_sincosf.h, and the counter-based generator both sides define -- bit for bit against libm / each other;
  2. the oracle with that generator against per-pixel moments of the REAL reference's frames (tests/golden/gi_stats.npz, made by
     tests/golden/make_golden.py from oracle/_ref/ref_render --gi): same mean image, same variance;
  3. (-m gpu) the HIP path against the oracle for the same seed: identical floats, identical ray counters.
"""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import assert_same_floats

HERE = os.path.dirname(os.path.abspath(__file__))
TWO_PI = 6.2831860


def crt_test_gi(pkg, device, what, a, b=None):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    out = np.empty_like(a)
    bp = None
    if b is not None:
        b = np.ascontiguousarray(b, dtype=np.uint32)
        bp = b.ctypes.data_as(C.c_void_p)
    L = pkg.lib()
    L.crt_test_gi.argtypes = [C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    rc = L.crt_test_gi(device, what, a.ctypes.data_as(C.c_void_p), bp, out.ctypes.data_as(C.c_void_p), a.size)
    assert rc == 0, rc
    return out


def angle_samples(n, seed):
    """Floats of [0, 2 pi] by bit pattern (every exponent the angles can have), plus the interval's ends and the quadrant edges."""
    rng = np.random.RandomState(seed)
    hi = int(np.float32(TWO_PI).view(np.uint32)) + 16
    bits = rng.randint(0, hi + 1, size=n, dtype=np.int64).astype(np.uint32)
    edges = np.array([0.0, 1e-45, 2.4e-4, 2.5e-4, 0.78539816, 0.7853982, 1.5707963, 1.5707964, 2.3561945, 3.1415925, 3.1415927,
                      3.926991, 4.712389, 5.4977875, 6.2831855, TWO_PI], dtype=np.float32).view(np.uint32)
    return np.concatenate([bits, edges])


# ------------------------------------------------------------------------------------------------------------ 1. arithmetic
def test_oracle_sincos_equals_this_machines_libm_over_the_whole_angle_range(oracle):
    """EVERY float of [0, 2 pi] (1.09e9 values, a few seconds with OpenMP): the oracle's restatement of glibc's sinf / cosf
    against libm itself."""
    tested, bad_sin, bad_cos = oracle.sincos_vs_libm(0.0, TWO_PI, 1)
    assert tested > 1_086_000_000
    assert (bad_sin, bad_cos) == (0, 0)


def test_product_sincos_host_build_equals_the_oracles(pkg, oracle):
    """csrc/glibc_sincosf.h (compiled for the host by the same compiler flags as for the GPU) against the oracle's restatement."""
    bits = angle_samples(4_000_000, 11)
    s, c = oracle.sincos_array(bits.view(np.float32))
    assert np.array_equal(crt_test_gi(pkg, -1, 0, bits), s.view(np.uint32))
    assert np.array_equal(crt_test_gi(pkg, -1, 1, bits), c.view(np.uint32))


def test_generator_host_build_equals_the_oracles(pkg, oracle):
    rng = np.random.RandomState(12)
    a = rng.randint(0, 2**32, size=1_000_000, dtype=np.int64).astype(np.uint32)
    b = np.concatenate([rng.randint(0, 2**32, size=500_000, dtype=np.int64), rng.randint(0, 200, size=500_000)]).astype(np.uint32)
    mix, uni = oracle.gi_array(a, b)
    assert np.array_equal(crt_test_gi(pkg, -1, 3, a, b), mix)
    assert np.array_equal(crt_test_gi(pkg, -1, 2, a, b), uni.view(np.uint32))


def test_generator_is_uniform_and_uncorrelated(oracle):
    """u(K, d) over consecutive keys / draws: range [0, 1), flat histogram, no correlation between the draws of one key or between
    neighbouring pixels' keys (what a hash-based generator can get wrong)."""
    n = 1 << 20
    seed_keys = oracle.gi_array(np.full(n, 7, dtype=np.uint32), np.arange(n, dtype=np.uint32))[0]  # mix(seed, pixel)
    keys = oracle.gi_array(seed_keys, np.zeros(n, dtype=np.uint32))[0]                                 # ... sample 0
    draws = [oracle.gi_array(keys, np.full(n, d, dtype=np.uint32))[1].astype(np.float64) for d in range(4)]
    for u in draws:
        assert u.min() >= 0.0 and u.max() < 1.0
        assert abs(u.mean() - 0.5) < 4 * np.sqrt(1 / 12 / n)
        assert abs(u.var() - 1 / 12) < 1e-3
        hist = np.bincount((u * 64).astype(np.int64), minlength=64)
        chi2 = ((hist - n / 64) ** 2 / (n / 64)).sum()
        assert chi2 < 130, chi2   # 63 degrees of freedom: P(chi2 > 130) ~ 1e-6
    for i in range(4):
        for j in range(i + 1, 4):
            assert abs(np.corrcoef(draws[i], draws[j])[0, 1]) < 5 / np.sqrt(n)
        assert abs(np.corrcoef(draws[i][:-1], draws[i][1:])[0, 1]) < 5 / np.sqrt(n)   # neighbouring pixels
    assert len(np.unique(keys)) > n * 0.999   # keys of different pixels (almost) never collide


# ------------------------------------------------------------------------------------- 2. the oracle against the real reference
def moments(oracle, blob, depth, n, r, frames, first_seed):
    o = oracle.OracleScene(blob)
    acc = np.zeros((o.height, o.width, 3), dtype=np.float64)
    acc2 = np.zeros_like(acc)
    for k in range(frames):
        rgb, _ = o.render(options=oracle.make_options(depth, use_gi=1, gi_sample_size=n, rays_per_pixel=r, gi_seed=first_seed + k))
        acc += rgb
        acc2 += rgb.astype(np.float64) ** 2
    mean = acc / frames
    return mean, (acc2 - frames * mean ** 2) / (frames - 1)


@pytest.mark.parametrize("name", ["hw11", "hw08"])
def test_oracle_gi_frames_have_the_references_distribution(oracle, name):
    """Per pixel and channel: the mean over `frames` oracle frames (seeds 0, 1, ...) against the mean over the reference's frames,
    in units of the standard error of their difference.  Were the two distributions the same, z would be standard normal."""
    g = np.load(os.path.join(HERE, "golden", "gi_stats.npz"))
    depth, n, r, ref_frames = (int(v) for v in g[name + "_params"])
    ref_mean, ref_var = g[name + "_mean"].astype(np.float64), g[name + "_var"].astype(np.float64)
    frames = 300
    mean, var = moments(oracle, g[name + "_blob"].tobytes(), depth, n, r, frames, 0)
    noisy = (ref_var > 1e-10) | (var > 1e-10)      # pixels the random numbers reach at all
    assert noisy.mean() > 0.3
    # pixels without randomness (background, mirrors that see only background): the very same float in every frame, on both sides
    assert np.abs(mean - ref_mean)[~noisy].max() < 1e-6
    z = (mean - ref_mean)[noisy] / np.sqrt(ref_var[noisy] / ref_frames + var[noisy] / frames)
    assert abs(z.mean()) < 0.12, z.mean()          # no bias (a bias of 1 % of the pixel noise would show as ~0.13)
    assert 0.85 < z.std() < 1.15, z.std()
    assert (np.abs(z) > 4.5).mean() < 1e-3
    assert abs(mean.mean() / ref_mean.mean() - 1) < 4e-3           # the frame's overall brightness
    assert abs(var[noisy].mean() / ref_var[noisy].mean() - 1) < 0.08   # the same noise level: same sampling density per pixel


def test_oracle_against_the_reference_binary_live(oracle, scenes):
    """The same comparison against frames rendered now by oracle/_ref/ref_render (development container only)."""
    if not oracle.reference_available():
        pytest.skip("oracle/_ref/ref_render is not built here")
    scene = scenes.make("hw11", width=32, height=24, detail=0.15)
    blob = scenes.to_blob(scene)
    frames = 120
    ref, _ = oracle.reference_render(blob, max_depth=3, gi=(1, 2), repeat=frames, all_frames=True)
    ref = ref.astype(np.float64)
    mean, var = moments(oracle, blob, 3, 1, 2, frames, 5000)
    ref_mean, ref_var = ref.mean(axis=0), ref.var(axis=0, ddof=1)
    noisy = (ref_var > 1e-10) | (var > 1e-10)
    z = (mean - ref_mean)[noisy] / np.sqrt((ref_var[noisy] + var[noisy]) / frames)
    assert abs(z.mean()) < 0.2 and 0.8 < z.std() < 1.2, (z.mean(), z.std())


def test_gi_without_samples_is_the_deterministic_frame(oracle, scenes):
    """USE_GI with GI_SAMPLE_SIZE 0 and one ray per pixel draws no random number: (direct + 0) * (1 / 1) per diffuse hit -- the
    non-GI frame, except that shadow rays no longer pass through refractive meshes (AccelerationStructure.cpp:67-71)."""
    scene = scenes.make("hw08", width=64, height=48, detail=0.3)   # no refractive mesh
    o = oracle.OracleScene(scenes.to_blob(scene))
    plain, _ = o.render(max_depth=3)
    gi, _ = o.render(options=oracle.make_options(3, use_gi=1, gi_sample_size=0, rays_per_pixel=1, gi_seed=9))
    assert_same_floats(gi + 0.0, plain + 0.0, "GI with no samples")
    scene = scenes.make("hw11", width=64, height=48, detail=0.2)   # a refractive sphere: its shadow appears
    o = oracle.OracleScene(scenes.to_blob(scene))
    plain, _ = o.render(max_depth=3)
    gi, _ = o.render(options=oracle.make_options(3, use_gi=1, gi_sample_size=0, rays_per_pixel=1, gi_seed=9))
    assert (gi <= plain + 1e-6).all() and (gi < plain - 1e-3).any()
    assert_same_floats(gi, o.render(options=oracle.make_options(3, use_gi=1, gi_sample_size=0, rays_per_pixel=1, gi_seed=10))[0], "seed")


# ------------------------------------------------------------------------------------------------------------ 3. the HIP path
GI_FRAMES = [  # scene, width, height, detail, depth, GI_SAMPLE_SIZE, RAYS_PER_PIXEL, seed
    ("hw11", 96, 64, 0.2, 2, 2, 2, 1),
    ("hw11", 64, 40, 0.2, 4, 1, 3, 77),
    ("hw08", 80, 60, 0.3, 1, 3, 1, 0xFFFFFFFF),
    ("hw12", 64, 40, 0.08, 2, 2, 2, 5),
    ("hw14", 64, 36, 0.04, 3, 1, 1, 123456),
    ("hw11", 64, 40, 0.2, 3, 0, 4, 3),     # no GI samples: only the pixel jitter is random
    ("hw07", 80, 60, 0.5, 0, 2, 0, 8),     # depth 0: every GI ray is background; RAYS_PER_PIXEL 0 renders like 1
]


@pytest.mark.gpu
def test_device_arithmetic_equals_the_host_build(pkg):
    bits = angle_samples(2_000_000, 21)
    for what in (0, 1):
        assert np.array_equal(crt_test_gi(pkg, 0, what, bits), crt_test_gi(pkg, -1, what, bits))
    rng = np.random.RandomState(22)
    a = rng.randint(0, 2**32, size=1_000_000, dtype=np.int64).astype(np.uint32)
    b = rng.randint(0, 2**32, size=1_000_000, dtype=np.int64).astype(np.uint32)
    for what in (2, 3):
        assert np.array_equal(crt_test_gi(pkg, 0, what, a, b), crt_test_gi(pkg, -1, what, a, b))


@pytest.mark.gpu
@pytest.mark.parametrize("case", GI_FRAMES, ids=lambda c: "%s-d%d-n%d-r%d" % (c[0], c[4], c[5], c[6]))
def test_gi_frame_equals_the_oracles(pkg, scenes, oracle, case, tmp_path):
    name, w, h, detail, depth, n, r, seed = case
    kw = dict(bitmap_size=32) if name == "hw12" else {}
    scene = scenes.make(name, width=w, height=h, detail=detail, **kw)
    folder = ""
    if scene.get("textures"):
        folder = str(tmp_path)
        scenes.write_bitmaps(scene, folder)
    tracer = pkg.Tracer(pkg.Scene(json_text=scenes.to_json(scene), folder=folder))
    o = oracle.OracleScene(scenes.to_blob(scene))
    want, want_counters = o.render(options=oracle.make_options(depth, use_gi=1, gi_sample_size=n, rays_per_pixel=r, gi_seed=seed))
    opts = dict(use_gi=True, gi_sample_size=n, rays_per_pixel=r, gi_seed=seed)
    got = tracer.render(options=pkg.make_options(depth, **opts))
    assert_same_floats(got, want, "GI frame")
    counted = tracer.render(options=pkg.make_options(depth, counters=True, **opts))   # the counting build: the reference's ray counts
    assert_same_floats(counted, want, "GI frame, counting build")
    assert tracer.stats().counters() == want_counters
    if (n and depth > 0) or r > 1:   # (the depth-0 case draws no random number)
        other = tracer.render(options=pkg.make_options(depth, use_gi=True, gi_sample_size=n, rays_per_pixel=r, gi_seed=seed ^ 1))
        assert not np.array_equal(other, got)   # another seed, another frame
    plain = tracer.render(max_depth=depth)      # ... and the deterministic path is untouched by a GI frame before it
    assert_same_floats(plain, o.render(max_depth=depth)[0], "non-GI frame after a GI frame")


GI_PATHS = [dict(mode=1),                                                  # render_lanes: the whole recursion per lane
            dict(bvh=0, step_budget=0),                                    # faithful ray-stream kernels
            dict(bvh=0, step_budget=8, shadow_budget=8, level0_budget=8),  # nearly every walk through the wave-per-ray kernels
            dict(bvh=0, heavy_level=0), dict(bvh=0, heavy_level=1000000),  # the levels per lane first / by the wave-per-ray kernel whole
            dict(bvh=0, step_budget=100000, shadow_budget=100000, heavy_level=0), # nothing evicted
            dict(bvh=0, side_blocks=0),
            dict(bvh=0),                                                   # the reference-order kernels' defaults
            dict(),                                                        # the defaults: the candidate-filter kernels, level by level
            dict(bvh=1, side_blocks=0),                                    # ... without a side stream
            dict(bvh=1, fetch_chunk=0x400040, side_blocks=4)]              # ... cursor claims of 64, the bulk shadow pass on four workgroups per CU


# every path on the room; the knot scene and the wide plan on the paths that differ there
GI_PATH_CASES = [("hw11", 0, t) for t in GI_PATHS] + [("hw14", 0, GI_PATHS[k]) for k in (2, 3, 5, 8, 9)] + [("hw11", 100, GI_PATHS[k]) for k in (1, 2, 5, 8)]


@pytest.mark.gpu
@pytest.mark.parametrize("name,meshes,tuning", GI_PATH_CASES,
                         ids=lambda v: v if isinstance(v, str) else (str(v) if isinstance(v, int) else ",".join("%s=%s" % kv for kv in v.items())))
def test_gi_frame_on_every_kernel_path(pkg, scenes, oracle, name, meshes, tuning):
    """The GI mode on the ray stream (jittered level-0 samples, gi_sample_size child rays per diffuse hit through the level queues,
    shadow rays that do not skip refractive meshes, the N-ary post-order in stream_resolve) and on every other path: the oracle's
    frame for the seed, bit for bit; 100 meshes: the wide plan."""
    scene = scenes.make(name, width=96, height=64, detail=0.04 if name == "hw14" else 0.2)
    if meshes:
        scene = scenes.scatter_meshes(scene, meshes)
    opts = dict(use_gi=1, gi_sample_size=2, rays_per_pixel=3, gi_seed=77)
    want, _ = oracle.OracleScene(scenes.to_blob(scene)).render(options=oracle.make_options(3, **opts))
    tracer = pkg.Tracer(pkg.Scene(json_text=scenes.to_json(scene)), tuning=pkg.make_tuning(**tuning))
    for frame in range(2):   # (the second frame is not probed and sizes its launches by the first)
        got = tracer.render(options=pkg.make_options(3, use_gi=True, gi_sample_size=2, rays_per_pixel=3, gi_seed=77))
        assert_same_floats(got, want, "GI %s %r frame %d" % (name, tuning, frame))
    assert tracer.stats().fallback_frames == 0


@pytest.mark.gpu
def test_gi_frame_that_overflows_fixed_queues_is_redone_by_the_lane_kernel(pkg, scenes, oracle):
    """Explicit capacities (crt_tuning) are never regrown: a GI frame that overflows them is redone in the same call by
    render_lanes<., true>, the queue-less last resort -- same frame, and crt_stats says so."""
    scene = scenes.make("hw11", width=96, height=64, detail=0.2)
    want, _ = oracle.OracleScene(scenes.to_blob(scene)).render(options=oracle.make_options(2, use_gi=1, gi_sample_size=2, rays_per_pixel=2, gi_seed=9))
    tracer = pkg.Tracer(pkg.Scene(json_text=scenes.to_json(scene)), tuning=pkg.make_tuning(ray_cap=512))
    for frame in (1, 2):
        got = tracer.render(options=pkg.make_options(2, use_gi=True, gi_sample_size=2, rays_per_pixel=2, gi_seed=9))
        assert_same_floats(got, want, "GI frame through the last resort")
        assert tracer.stats().fallback_frames == frame


@pytest.mark.gpu
def test_gi_frame_does_not_depend_on_how_the_image_is_split(pkg, scenes, oracle):
    """Keys are (seed, pixel, sample, position in the ray tree): the same frame from one context, from two contexts sharing the
    tiles (crt_multi with the same device twice), and under every bucket mode of RayTracer::render."""
    scene = scenes.make("hw11", width=80, height=48, detail=0.2)
    hs = pkg.Scene(json_text=scenes.to_json(scene))
    opts = pkg.make_options(2, use_gi=True, gi_sample_size=2, rays_per_pixel=2, gi_seed=31)
    one = pkg.Tracer(hs).render(options=opts)
    two = pkg.Tracer(hs, devices=[0, 0]).render(options=opts)
    assert_same_floats(two, one, "two contexts")
    for optimization in (pkg.OPT_REGIONS, pkg.OPT_BUCKETS_POOL, pkg.OPT_BVH):
        assert_same_floats(pkg.Tracer(hs).render(options=opts, optimization=optimization), one, "mode %d" % optimization)


@pytest.mark.gpu
def test_gi_mean_image_matches_the_references(pkg, scenes):
    """End to end against the real reference's moments (the fixture of layer 2), through the C ABI on the GPU."""
    g = np.load(os.path.join(HERE, "golden", "gi_stats.npz"))
    name = "hw11"
    depth, n, r, ref_frames = (int(v) for v in g[name + "_params"])
    scene = scenes.make("hw11", width=48, height=32, detail=0.15)   # tests/golden/make_golden.py GI_CASES[0]
    assert scenes.to_blob(scene) == g[name + "_blob"].tobytes()
    tracer = pkg.Tracer(pkg.Scene(json_text=scenes.to_json(scene)))
    frames = 300
    acc = np.zeros((tracer.height, tracer.width, 3), dtype=np.float64)
    acc2 = np.zeros_like(acc)
    for k in range(frames):
        rgb = tracer.render(options=pkg.make_options(depth, use_gi=True, gi_sample_size=n, rays_per_pixel=r, gi_seed=100000 + k))
        acc += rgb
        acc2 += rgb.astype(np.float64) ** 2
    mean = acc / frames
    var = (acc2 - frames * mean ** 2) / (frames - 1)
    ref_mean, ref_var = g[name + "_mean"].astype(np.float64), g[name + "_var"].astype(np.float64)
    noisy = (ref_var > 1e-10) | (var > 1e-10)
    z = (mean - ref_mean)[noisy] / np.sqrt(ref_var[noisy] / ref_frames + var[noisy] / frames)
    assert abs(z.mean()) < 0.12 and 0.85 < z.std() < 1.15, (z.mean(), z.std())


@pytest.mark.gpu
@pytest.mark.parametrize("name,depth", [("hw14", 3), ("hw11", 4), ("hw12", 2)])
def test_gi_filter_kernels_equal_the_reference_order_kernels_at_full_detail(pkg, scenes, name, depth, tmp_path):
    """The GI mode through the candidate filter (kernel_bvh.h: bvh_trace_shade_gi) against the reference-order GI kernels (crt_tuning::bvh = 0,
    which the other tests pin to the oracle) on the FULL-detail benchmark scenes (207,954 / 60,156 triangles) at 480x270: the same floats,
    two seeds, with and without the side stream (tools/gi_big.py is the same comparison as a tool)."""
    scene = scenes.make(name, width=480, height=270)
    folder = ""
    if scene.get("textures"):
        folder = str(tmp_path) + "/"
        scenes.write_bitmaps(scene, folder)
    hs = pkg.Scene(json_text=scenes.to_json(scene), folder=folder)
    frames = {}
    for tag, tuning in (("reference-order", dict(bvh=0)), ("filter", dict()), ("filter, no side stream", dict(side_blocks=0))):
        tracer = pkg.Tracer(hs, tuning=pkg.make_tuning(**tuning))
        for seed in (5, 6):
            frames[(tag, seed)] = tracer.render(options=pkg.make_options(depth, use_gi=True, gi_sample_size=2, rays_per_pixel=2, gi_seed=seed)).copy()
        assert tracer.stats().fallback_frames == 0
    for seed in (5, 6):
        for tag in ("filter", "filter, no side stream"):
            assert_same_floats(frames[(tag, seed)], frames[("reference-order", seed)], "GI %s seed %d: %s vs reference-order kernels" % (name, seed, tag))
