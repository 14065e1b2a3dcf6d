"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle and the committed reference
fixtures on the same inputs.  Bar: bit-exact floats (stricter than the 1e-5 north-star tolerance), identical
quantised bytes, identical work counters."""
import os

import numpy as np
import pytest

from helpers import assert_same_floats, blob_to_scene, load_golden, small_case

pytestmark = pytest.mark.gpu

CONFIGS = ["hw07", "hw08", "hw11", "hw14", "hw12"]


def make_tracer(pkg, scenes, scene, folder="", tuning=None):
    if scene.get("textures") and folder:
        scenes.write_bitmaps(scene, folder)
    return pkg.Tracer(pkg.Scene(json_text=scenes.to_json(scene), folder=folder),
                      tuning=pkg.make_tuning(**tuning) if tuning else None)


@pytest.mark.parametrize("name", CONFIGS)
def test_frame_matches_oracle(pkg, scenes, oracle, name, tmp_path):
    scene, depth, folder = small_case(scenes, name, tmp_path)
    tracer = make_tracer(pkg, scenes, scene, folder)
    got = tracer.render(max_depth=depth, counters=True)   # counting build: every ray walked the reference's way
    stats = tracer.stats()
    want, counters = oracle.OracleScene(scenes.to_blob(scene)).render(depth)
    assert_same_floats(got, want, name)
    assert stats.counters_valid == 1
    assert stats.counters() == counters
    again = tracer.render(max_depth=depth)                # production build: lean walk + heavy_trace
    assert_same_floats(again, want, name + " (production kernels)")
    q = tracer.read_quantized()                           # PPMColor on the device
    assert np.array_equal(q.astype(np.uint16), oracle.quantize(want))


@pytest.mark.parametrize("name", CONFIGS + ["coverage", "uvwrap"])
def test_frame_matches_reference_fixture(pkg, scenes, name, tmp_path):
    g = load_golden(name)
    scene = blob_to_scene(g["blob"])
    tracer = make_tracer(pkg, scenes, scene, str(tmp_path))
    got = tracer.render(max_depth=g["depth"])
    assert_same_floats(got, g["rgb"], name)               # the REAL reference's frame, bit for bit


def test_ppm_file_matches_reference_bytes(pkg, scenes, tmp_path):
    g = load_golden("hw07")
    tracer = make_tracer(pkg, scenes, blob_to_scene(g["blob"]))
    path = str(tmp_path / "frame.ppm")
    tracer.render(max_depth=g["depth"], ppm_path=path)    # RayTracer::render(pathToImage, ...) writes the file
    assert open(path, "rb").read() == g["ppm"]


# kernel paths, selected through crt_tuning (include/crt_hip.h); mode: 1 = lanes; bvh: 1 = the candidate-filter kernels (the default
# where the scene has a filter), 0 = the reference-order kernels alone, 2 = the filter kernels' bounds-checked build
KERNEL_PATHS = [
    dict(mode=1),                                                # recursive one-ray-per-lane kernel for every pixel (the queue-less fallback)
    dict(bvh=1),                                                 # the filter kernels (csrc/kernel_bvh.h), the levels below level 0 through the level queue
    dict(bvh=1, side_blocks=0),                                  # ... without a side stream
    dict(bvh=1, side_blocks=4, level_queue=2),                   # ... the bulk shadow pass on four workgroups per CU, the queue on two
    dict(bvh=1, fetch_chunk=0x400040),                           # ... cursor claims of 64 (a wave's refill) in the bulk shadow pass
    dict(bvh=1, fetch_chunk=0x1000800, side_blocks=2),           # ... of 2048 slots and 256 primary rays
    dict(bvh=2),                                                 # ... their bounds-checked build
    dict(bvh=1, level_queue=0),                                  # ... one launch per level
    dict(bvh=1, level_queue=4097),                               # ... level 0 held back until the queue's first launch has given up (a serialising profiler does that)
    dict(bvh=1, level_queue=513),                                # ... the queue's launch behind level 0 instead of beside it
    dict(bvh=1, level_queue=258),                                # ... every child ray through the queue (none continues in its parent's lane)
    dict(bvh=2, level_queue=0, side_blocks=0),
    dict(bvh=0),                                                 # the reference-order kernels: plan kernels + wave-per-ray kernels
    dict(bvh=0, step_budget=8, shadow_budget=8, level0_budget=8),       # nearly every walk through the wave-per-ray kernels
    dict(bvh=0, step_budget=0),                                         # wave-per-ray kernels off: faithful stream kernels
    dict(bvh=0, heavy_level=1000000),                                   # deeper levels entirely by the wave-per-ray kernel
    dict(bvh=0, heavy_level=0),                                         # ... entirely by the per-lane kernel first
    dict(bvh=0, heavy_level=3000),                                      # ... some levels straddle the threshold
    dict(bvh=0, side_blocks=0),                                         # no side stream
    dict(bvh=0, step_budget=100000, shadow_budget=100000, heavy_level=0),  # nothing evicted: per-lane kernels alone
    dict(bvh=0, side_blocks=4),                                         # the bulk shadow pass on four workgroups per CU
]


@pytest.mark.parametrize("tuning", KERNEL_PATHS, ids=lambda t: ",".join("%s=%s" % kv for kv in t.items()))
@pytest.mark.parametrize("name", ["hw11", "hw12", "hw14"])
def test_every_kernel_path_gives_the_same_frame(pkg, scenes, oracle, name, tuning, tmp_path):
    scene, depth, folder = small_case(scenes, name, tmp_path)
    tracer = make_tracer(pkg, scenes, scene, folder, tuning=tuning)
    got = tracer.render(max_depth=depth)
    want, _ = oracle.OracleScene(scenes.to_blob(scene)).render(depth)
    assert_same_floats(got, want, "%s %r" % (name, tuning))
    assert tracer.stats().fallback_frames == 0
    # the second frame of a context sizes (or leaves out) launches by the first frame's counters: the same pixels again
    assert_same_floats(tracer.render(max_depth=depth), want, "%s %r, second frame" % (name, tuning))
    assert tracer.stats().fallback_frames == 0


MANY_PATHS = [dict(bvh=0), dict(bvh=0, step_budget=8, shadow_budget=8, level0_budget=8), dict(bvh=0, heavy_level=0),
              dict(bvh=0, step_budget=100000, shadow_budget=100000, heavy_level=0), dict(bvh=0, step_budget=0), dict(mode=1), None, dict(bvh=2)]


@pytest.mark.parametrize("n_meshes,tuning", [(100, t) for t in MANY_PATHS] + [(150, MANY_PATHS[k]) for k in (0, 1, 3, 6)] + [(300, MANY_PATHS[k]) for k in (0, 1, 6, 7)],
                         ids=lambda v: str(v) if isinstance(v, int) else ("defaults" if v is None else ",".join("%s=%s" % kv for kv in v.items())))
def test_many_meshes_stay_on_the_plan_kernels(pkg, scenes, oracle, n_meshes, tuning):
    """A top-level tree with hundreds of leaves (the reference's scenes have a handful of meshes): up to 256 meshes the wide
    plan kernels render it (eight-word mesh masks, leaf groups; the wave-per-ray kernels walk the top-level leaf sequence 64
    leaves at a time, and know repeats among the first 128 meshes only), beyond that the faithful kernels.  Same frame."""
    scene = scenes.scatter_meshes(scenes.make("hw11", width=192, height=108, detail=0.25), n_meshes)
    want, _ = oracle.OracleScene(scenes.to_blob(scene)).render(4)
    tracer = make_tracer(pkg, scenes, scene, tuning=tuning)
    if tuning == dict(bvh=0):
        assert ("plan_wide" in tracer.kernels()["level0"]) == (n_meshes <= 256)
    if tuning is None:
        assert tracer.kernels()["level0"].startswith("bvh_trace_")   # the filter kernels take any number of meshes
    for frame in range(2):
        assert_same_floats(tracer.render(max_depth=4), want, "%d meshes %r frame %d" % (n_meshes, tuning, frame))
    assert tracer.stats().fallback_frames == 0


@pytest.mark.parametrize("caps", [dict(ray_cap=256), dict(node_cap=1), dict(shadow_cap=1 << 14)])
def test_queue_overflow_hands_the_frame_to_the_fallback(pkg, scenes, oracle, caps):
    """Ray queues too small for the frame: the stream kernels raise the overflow word and stop, render_lanes redoes the
    whole frame without queues.  Same pixels, and crt_stats says that the fallback ran."""
    scene, depth, _ = small_case(scenes, "hw11")
    want, _ = oracle.OracleScene(scenes.to_blob(scene)).render(depth)
    tracer = make_tracer(pkg, scenes, scene, tuning=caps)
    assert_same_floats(tracer.render(max_depth=depth), want, "overflow %r" % (caps,))
    assert tracer.stats().fallback_frames == 1
    assert_same_floats(tracer.render(max_depth=depth), want, "overflow again %r" % (caps,))
    assert tracer.stats().fallback_frames == 2
    roomy = make_tracer(pkg, scenes, scene)
    assert_same_floats(roomy.render(max_depth=depth), want, "default capacities")
    assert roomy.stats().fallback_frames == 0


def test_depth_rule_and_bias_options(pkg, scenes, oracle):
    scene, _, _ = small_case(scenes, "hw11")
    tracer = make_tracer(pkg, scenes, scene)
    o = oracle.OracleScene(scenes.to_blob(scene))
    for depth in (0, 1, 2, 5):                            # depth == MAX_DEPTH is traced, depth > MAX_DEPTH is background
        assert_same_floats(tracer.render(max_depth=depth), o.render(depth)[0], "depth %d" % depth)
    opts = pkg.make_options(3, shadow_bias=1e-3, reflection_bias=2e-4, refraction_bias=5e-4)
    want, _ = o.render(3, options=oracle.make_options(3, 1e-3, 2e-4, 5e-4))
    assert_same_floats(tracer.render(options=opts), want, "biases")


def test_camera_updates_keep_the_scene_resident(pkg, scenes, oracle):
    # the animation driver's loop (app/animation.cpp:24-38): new camera, same tracer
    scene, depth, _ = small_case(scenes, "hw14")
    tracer = make_tracer(pkg, scenes, scene)
    o = oracle.OracleScene(scenes.to_blob(scene))
    pos0, mat0 = np.array(scene["camera"]["position"], dtype=np.float32), np.array(scene["camera"]["matrix"], dtype=np.float32)
    for deg in (0.0, 17.5, -40.0):
        pos, mat = pkg.camera_apply(pos0, mat0, "pan", deg)
        pos, mat = pkg.camera_apply(pos, mat, "truck", [0.1, 0.05, -0.3])
        tracer.set_camera(pos, mat)
        o.set_camera(pos, mat)
        assert_same_floats(tracer.render(max_depth=depth), o.render(depth)[0], "pan %g" % deg)


def test_uncovered_pixels_keep_their_previous_value(pkg, scenes, oracle):
    # bucket_size 7 on 100x60 covers only part of the frame; the colour buffer persists between renders
    scene = scenes.make("hw08", width=100, height=60, detail=0.3)
    scene["settings"]["image_settings"]["bucket_size"] = 7
    tracer = make_tracer(pkg, scenes, scene)
    o = oracle.OracleScene(scenes.to_blob(scene))
    first = tracer.render(max_depth=1)
    want = o.render(1)[0]
    assert_same_floats(first, want, "partial coverage")
    full = tracer.render(max_depth=1, optimization=pkg.OPT_BVH)           # one rectangle: the whole frame
    pos, mat = pkg.camera_apply(scene["camera"]["position"], scene["camera"]["matrix"], "pan", 25.0)
    tracer.set_camera(pos, mat)
    o.set_camera(pos, mat)
    mixed = tracer.render(max_depth=1)                                    # partial again: uncovered pixels keep `full`
    covered = np.zeros((60, 100), dtype=bool)
    for row, col, w, h in oracle.bucket_grid(100, 60, 7):
        covered[row:min(60, row + h), col:min(100, col + w)] = True
    moved = o.render(1, buffer=np.ascontiguousarray(full.copy()))[0]
    assert_same_floats(mixed, moved, "persistence")
    assert np.array_equal(mixed[~covered], full[~covered]) and not np.array_equal(mixed[covered], full[covered])


def test_empty_and_degenerate_scenes(pkg, scenes, oracle):
    # no objects at all; a mesh with a degenerate (zero-area) triangle; a constant-material mesh; no lights
    base = scenes.make("hw07", width=48, height=32, detail=0.3)
    empty = dict(base, objects=[])
    t = make_tracer(pkg, scenes, empty)
    assert_same_floats(t.render(max_depth=2), oracle.OracleScene(scenes.to_blob(empty)).render(2)[0], "empty scene")
    const = dict(base, materials=[{"type": "constant", "albedo": [1, 0, 0], "smooth_shading": False}])
    t = make_tracer(pkg, scenes, const)
    assert_same_floats(t.render(max_depth=2), oracle.OracleScene(scenes.to_blob(const)).render(2)[0], "constant material")
    dark = dict(base, lights=[])
    t = make_tracer(pkg, scenes, dark)
    assert_same_floats(t.render(max_depth=2), oracle.OracleScene(scenes.to_blob(dark)).render(2)[0], "no lights")
    odd = scenes.make("hw08", width=50, height=30, detail=0.3)           # frame not a multiple of the 8x8 tile
    t = make_tracer(pkg, scenes, odd)
    assert_same_floats(t.render(max_depth=1), oracle.OracleScene(scenes.to_blob(odd)).render(1)[0], "ragged tiles")


def test_axis_parallel_rays(pkg, scenes, oracle):
    # camera square-on to the room: rays through the image centre column/row have direction components below
    # FLT_EPSILON only for odd sizes; an odd frame makes the centre pixel's x and y exactly 0 -> the parallel-axis
    # branch of the box test (BoundingBox.h:90-93) and shadow rays straight up from the floor under a light
    scene = scenes.make("hw11", width=65, height=37, detail=0.2)
    scene["lights"].append({"intensity": 20, "position": [0.0, 2.0, -4.0]})
    scene["camera"]["position"] = [0.0, 0.0, 0.5]
    t = make_tracer(pkg, scenes, scene)
    assert_same_floats(t.render(max_depth=4), oracle.OracleScene(scenes.to_blob(scene)).render(4)[0], "axis-parallel rays")


# ---------------------------------------------------------------------------------------------- full-size properties
@pytest.fixture(scope="module")
def full_hw14(pkg, scenes):
    scene = scenes.make("hw14")                                           # 1920x1080, ~208k triangles
    tracer = pkg.Tracer(pkg.Scene(json_text=scenes.to_json(scene)))
    return scene, tracer


def test_full_size_render_is_reproducible_and_quantises(full_hw14, oracle):
    scene, tracer = full_hw14
    a = tracer.render(max_depth=8).copy()
    b = tracer.render(max_depth=8)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))           # no run-to-run variation (no float atomics)
    assert np.isfinite(a).all()
    q = tracer.read_quantized()
    assert np.array_equal(q.astype(np.uint16), oracle.quantize(a))        # device quantiser == PPMColor rule
    assert len(np.unique(q.reshape(-1, 3), axis=0)) > 1000                # an actual image, not a constant


def test_full_size_tile_partition_reassembles_the_frame(pkg, full_hw14):
    import torch
    scene, tracer = full_hw14
    want = tracer.render(max_depth=8).copy()
    tiles = __import__("importlib").import_module("course-assignment-danielhalachev_amd.tiles")
    W, H, world = tracer.width, tracer.height, 4
    per = tiles.tiles_per_rank(W, H, world)
    dev = torch.device("cuda", 0)
    gathered = torch.zeros(world * per * 192, dtype=torch.float32, device=dev)
    frame = torch.zeros(H * W * 3, dtype=torch.float32, device=dev)
    opts = pkg.make_options(8)
    for rank in range(world):                                             # the four "GPUs" one after another on this one
        part = gathered[rank * per * 192:(rank + 1) * per * 192]
        tracer.render_tiles_device(opts, rank, world, part.data_ptr())
        torch.cuda.synchronize()
    tracer.unpack_tiles_device(gathered.data_ptr(), world, per * 192, frame.data_ptr())
    torch.cuda.synchronize()
    got = frame.cpu().numpy().reshape(H, W, 3)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    # and the numpy mirror of the packed layout agrees with the device kernels
    host = tiles.unpack_tiles(gathered.cpu().numpy().reshape(world, per, 64, 3), W, H, world)
    assert np.array_equal(host.view(np.uint32), want.view(np.uint32))


def test_full_size_sample_against_oracle(full_hw14, oracle, scenes):
    # the oracle is too slow for the whole 1080p frame in a test, so check a band of rows: same scene, same camera
    scene, tracer = full_hw14
    got = tracer.render(max_depth=8)
    o = oracle.OracleScene(scenes.to_blob(scene))
    band = dict(scene)
    rows = slice(520, 544)                                                # 24 rows through the knot and both spheres
    buf = np.zeros((1080, 1920, 3), dtype=np.float32)
    # render only those rows with the oracle by giving it a one-bucket grid covering them: emulate via per-ray shoots
    for row in range(rows.start, rows.stop, 4):
        for col in range(0, 1920, 3):
            org, d = o.camera_ray(row, col)
            buf[row, col] = o.shoot(org, d, ray_type=0, depth=0, max_depth=8)
            assert np.array_equal(buf[row, col].view(np.uint32), got[row, col].view(np.uint32)), (row, col)


@pytest.mark.parametrize("name", ["hw08", "hw11", "hw12", "hw14"])
def test_full_size_configs_against_oracle_samples_and_counting_build(pkg, scenes, oracle, name, tmp_path):
    """BASELINE.json's other configurations at their full sizes (HW12: 3840x2160 with its bitmap texture): the production
    kernels must give the counting build's frame bit for bit over the WHOLE frame, and the oracle's colour on a grid of
    sample pixels (the oracle is too slow for 8 Mpixels inside a test)."""
    scene = scenes.make(name)
    depth = scenes.CONFIGS[name][3]
    tracer = make_tracer(pkg, scenes, scene, str(tmp_path))
    got = tracer.render(max_depth=depth).copy()
    ref = tracer.render(max_depth=depth, counters=True)                   # every ray walked the reference's way
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    o = oracle.OracleScene(scenes.to_blob(scene))
    H, W = got.shape[:2]
    for row in range(H // 16, H, H // 8):
        for col in range(W // 64, W, W // 32):
            org, d = o.camera_ray(row, col)
            want = o.shoot(org, d, ray_type=0, depth=0, max_depth=depth)
            assert np.array_equal(want.view(np.uint32), got[row, col].view(np.uint32)), (name, row, col)


def _random_camera(rng, scene_radius):
    # a camera anywhere in (and around) the scene looking anywhere: rays that start inside meshes, graze boxes, run along axes
    pos = rng.uniform(-scene_radius, scene_radius, 3).astype(np.float32)
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    rot = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                    [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                    [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]], dtype=np.float32)
    return pos, rot.reshape(9)


@pytest.mark.parametrize("name", ["hw11", "hw14", "hw12"])
def test_early_exits_change_nothing_from_random_viewpoints(pkg, scenes, name, tmp_path):
    """Distance pruning and the shadow early exit skip work the reference does; the frames must not move by a bit.
    The counting build walks every ray to the end the reference's way (and is itself checked against the oracle
    above), so it is the yardstick here, from viewpoints the fixed camera never shows."""
    scene = scenes.make(name, width=192, height=108, detail=0.5)
    folder = str(tmp_path)
    tracer = make_tracer(pkg, scenes, scene, folder)
    depth = scenes.CONFIGS[name][3]
    rng = np.random.default_rng(1234)
    for k in range(6):
        pos, mat = _random_camera(rng, 2.5)
        if k == 0:                                        # axis-aligned view from the origin: rays parallel to box faces
            pos, mat = np.zeros(3, np.float32), np.eye(3, dtype=np.float32).reshape(9)
        tracer.set_camera(pos, mat)
        want = tracer.render(max_depth=depth, counters=True).copy()
        got = tracer.render(max_depth=depth)
        assert_same_floats(got, want, "%s view %d" % (name, k))


@pytest.mark.parametrize("name", ["hw11", "hw14"])
def test_executed_work_is_a_subset_of_the_reference_work(pkg, scenes, name):
    """collect_counters == 2 runs the production kernels with tallies of the tests they execute: the frame is the same,
    and the triangle tally is no larger than the reference's count (the counting build's)."""
    scene, depth, _ = small_case(scenes, name)
    tracer = make_tracer(pkg, scenes, scene)
    plain = tracer.render(max_depth=depth).copy()
    counted = tracer.render(max_depth=depth, counters=True).copy()
    ref = tracer.stats().counters()
    tallied = tracer.render(max_depth=depth, counters=2)
    ex = tracer.executed_counters()
    assert_same_floats(tallied, plain, name + " (tallying run)")
    assert_same_floats(counted, plain, name + " (counting build)")
    assert 0 < ex["tri_tests"] <= ref["tri_tests"]       # triangles: only leaves the reference visits too, each mesh once
    assert 0 < ex["box_tests"] <= 8 * ref["box_tests"]   # boxes: the wide kernels test 4 / 64 at a time, so not necessarily fewer


@pytest.mark.parametrize("n_objects", [72, 140, 280])
def test_more_meshes_than_the_seen_mask_holds(pkg, scenes, oracle, n_objects):
    """Many objects: the top-level tree is several levels deep and lists meshes in many leaves.  72: beyond the 64 top-level
    nodes / meshes the register and plan forms hold (the LDS copy takes over); 140: beyond the 128-bit one-walk-per-mesh mask
    (meshes 128.. are simply walked at every listed occurrence, like the reference does); 280: beyond the LDS copy as well."""
    base = scenes.make("hw11", width=96, height=64, detail=0.15)
    rng = np.random.default_rng(7)
    objects = list(base["objects"])
    n_mat = len(base["materials"])
    while len(objects) < n_objects:                                        # small tetrahedra scattered through the room
        c = rng.uniform([-1.6, -0.8, -5.5], [1.6, 1.2, -2.0]).astype(np.float32)
        v = (c + rng.uniform(-0.25, 0.25, (4, 3))).astype(np.float32)
        objects.append({"material_index": int(rng.integers(0, n_mat)), "vertices": v,
                        "triangles": np.array([[0, 1, 2], [0, 3, 1], [1, 3, 2], [2, 3, 0]], dtype=np.uint32)})
    scene = dict(base, objects=objects)
    tracer = make_tracer(pkg, scenes, scene)
    want, counters = oracle.OracleScene(scenes.to_blob(scene)).render(4)
    got = tracer.render(max_depth=4, counters=True)
    assert tracer.stats().counters() == counters
    assert_same_floats(got, want, "%d meshes (counting build)" % n_objects)
    assert_same_floats(tracer.render(max_depth=4), want, "%d meshes (production kernels)" % n_objects)
    heavy = make_tracer(pkg, scenes, scene, tuning=dict(bvh=0, step_budget=8, shadow_budget=8))   # nearly every walk by the wave-per-ray kernels
    assert_same_floats(heavy.render(max_depth=4), want, "%d meshes (wave-per-ray kernels)" % n_objects)


# ---------------------------------------------------------------------------------------------- several devices, one tracer
@pytest.mark.parametrize("parts", [2, 3])
def test_multi_device_tracer_renders_the_same_frame(pkg, scenes, oracle, parts):
    """crt::RayTracer(scene, devices[]) / crt_multi_*: one context per listed device, tiles dealt round-robin, packed tiles
    copied to devices[0] and scattered there.  A one-GPU box lists its device several times: same code path, peer copy onto
    itself.  Same pixels as the single-device tracer, same work counters in total."""
    scene, depth, _ = small_case(scenes, "hw14")
    want, counters = oracle.OracleScene(scenes.to_blob(scene)).render(depth)
    tracer = pkg.Tracer(pkg.Scene(json_text=scenes.to_json(scene)), devices=[0] * parts)
    assert_same_floats(tracer.render(max_depth=depth), want, "%d parts" % parts)
    assert np.array_equal(tracer.read_quantized().astype(np.uint16), oracle.quantize(want))
    counted = tracer.render(max_depth=depth, counters=True)               # counting build on every part
    assert_same_floats(counted, want, "%d parts, counting build" % parts)
    st = tracer.stats()
    assert st.counters_valid == 1 and st.counters() == counters and st.pixels == want.shape[0] * want.shape[1]
    assert st.fallback_frames == 0
    # a new camera reaches every part
    pos, mat = pkg.camera_apply(scene["camera"]["position"], scene["camera"]["matrix"], "pan", 20.0)
    tracer.set_camera(pos, mat)
    o = oracle.OracleScene(scenes.to_blob(scene))
    o.set_camera(pos, mat)
    assert_same_floats(tracer.render(max_depth=depth), o.render(depth)[0], "%d parts, moved camera" % parts)


def test_multi_device_tracer_without_peer_access_stages_its_tiles(pkg, scenes, oracle):
    """A part whose device cannot store into devices[0]'s memory copies its tiles through pinned host memory (crt_multi_create
    checks hipDeviceCanAccessPeer / hipDeviceEnablePeerAccess and says which parts those are); forced here, on one device."""
    scene, depth, _ = small_case(scenes, "hw14")
    want, _ = oracle.OracleScene(scenes.to_blob(scene)).render(depth)
    tracer = pkg.Tracer(pkg.Scene(json_text=scenes.to_json(scene)), devices=[0, 0, 0])
    assert tracer.staged_parts() == 0 and tracer.peer_note() == ""        # the same device three times: nothing to stage
    assert_same_floats(tracer.render(max_depth=depth), want, "peer copies")
    tracer.force_staged()
    assert tracer.staged_parts() == 2
    for frame in range(2):
        assert_same_floats(tracer.render(max_depth=depth), want, "staged copies, frame %d" % frame)
    tracer.force_staged(False)
    assert_same_floats(tracer.render(max_depth=depth), want, "peer copies again")


def test_multi_device_tracer_redoes_a_partition_that_failed_half_way(pkg, scenes, oracle):
    """A re-partition that stops at a failed allocation leaves no cached rectangles behind: the call reports the error, and the
    next call with the very same rectangles rebuilds everything instead of launching into half-built buffers."""
    scene, depth, _ = small_case(scenes, "hw11")
    want, _ = oracle.OracleScene(scenes.to_blob(scene)).render(depth)
    tracer = pkg.Tracer(pkg.Scene(json_text=scenes.to_json(scene)), devices=[0, 0])
    tracer.fail_next_alloc()
    with pytest.raises(Exception):
        tracer.render(max_depth=depth)
    assert_same_floats(tracer.render(max_depth=depth), want, "after the failed partition")
    assert_same_floats(tracer.render(max_depth=depth), want, "and with the partition cached")


def test_multi_device_tracer_keeps_uncovered_pixels(pkg, scenes, oracle):
    # bucket_size 7 on 100x60 covers part of the frame only: coverage masks travel with the tiles
    scene = scenes.make("hw08", width=100, height=60, detail=0.3)
    scene["settings"]["image_settings"]["bucket_size"] = 7
    tracer = pkg.Tracer(pkg.Scene(json_text=scenes.to_json(scene)), devices=[0, 0])
    o = oracle.OracleScene(scenes.to_blob(scene))
    assert_same_floats(tracer.render(max_depth=1), o.render(1)[0], "partial coverage on two parts")
    full = tracer.render(max_depth=1, optimization=pkg.OPT_BVH).copy()    # one rectangle: the whole frame
    pos, mat = pkg.camera_apply(scene["camera"]["position"], scene["camera"]["matrix"], "pan", 25.0)
    tracer.set_camera(pos, mat)
    o.set_camera(pos, mat)
    mixed = tracer.render(max_depth=1)                                    # partial again: uncovered pixels keep `full`
    moved = o.render(1, buffer=np.ascontiguousarray(full.copy()))[0]
    assert_same_floats(mixed, moved, "persistence on two parts")


def test_multi_device_full_size(pkg, full_hw14):
    scene, tracer = full_hw14
    want = tracer.render(max_depth=8).copy()
    multi = pkg.Tracer(tracer.scene, devices=[0, 0, 0, 0])
    got = multi.render(max_depth=8)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_queue_capacities_follow_the_frames(pkg, scenes, oracle):
    """Four of the room's walls are mirrors: four secondary rays per pixel, more ray-tree nodes than the queues are first
    sized for.  The first frame's attempt overflows and is repeated with larger queues inside the same call
    (crt_stats::queue_regrows); no frame is redone by the queue-less kernel, and the first one costs about two settled ones."""
    import time
    scene = scenes.make("hw11", width=256, height=192, detail=0.15)
    kinds = [m["type"] for m in scene["materials"]]
    mirror = kinds.index("reflective")
    for o in scene["objects"][0:4]:                                       # four of the room's walls become mirrors
        o["material_index"] = mirror
    want, counters = oracle.OracleScene(scenes.to_blob(scene)).render(8)
    assert counters["secondary_rays"] > 4 * 256 * 192                     # > 5 nodes per pixel: beyond the starting capacity
    tracer = make_tracer(pkg, scenes, scene)
    regrows, sizes, device_ms = [], [], []
    for frame in range(6):
        assert_same_floats(tracer.render(max_depth=8), want, "frame %d" % frame)
        st = tracer.stats()
        assert st.fallback_frames == 0                                    # never the last resort
        regrows.append(st.queue_regrows)
        sizes.append(st.queue_bytes)
    assert regrows[0] == 1                                                # the first frame did not fit at once ...
    assert regrows[-1] == regrows[0]                                      # ... the later ones do, without probing
    assert sizes == sorted(sizes) and sizes[-1] == sizes[0]               # grown inside the first call
    # device time of the attempts of a fresh context's first frame against a settled frame's, at a size where the frame and not
    # the launches is what costs (1920x1080: 1.76x, tools/regrow_time.py)
    big = scenes.make("hw11", width=1280, height=720, detail=0.15)
    for o in big["objects"][0:4]:
        o["material_index"] = mirror
    fresh = make_tracer(pkg, scenes, big)
    fresh.render(max_depth=8)
    attempts = int(fresh.stats().queue_regrows) + 1
    for _ in range(4):
        fresh.render(max_depth=8)
    # one failed attempt, one repeat, never the queue-less redo (what the first frame COSTS against a settled one is measured by
    # tools/regrow_time.py, not asserted here: wall-clock bounds on a shared box are flaky)
    assert attempts == 2 and fresh.stats().fallback_frames == 0 and int(fresh.stats().queue_regrows) == 1


def test_async_frames_equal_synchronous_ones(pkg, scenes, oracle):
    """crt_render_async / crt_wait: frames enqueued without waiting, two contexts alternating (two frames in flight), give
    the synchronous call's pixels; the quantised copy follows PPMColor."""
    scene, depth, _ = small_case(scenes, "hw14")
    hs = pkg.Scene(json_text=scenes.to_json(scene))
    o = oracle.OracleScene(scenes.to_blob(scene))
    ring = [pkg.Tracer(hs), pkg.Tracer(hs)]
    opts = pkg.make_options(depth)
    H, W = hs.height, hs.width
    bufs = [(np.zeros((H, W, 3), np.float32), np.zeros((H, W, 3), np.uint8)) for _ in ring]
    cams = [pkg.camera_apply(scene["camera"]["position"], scene["camera"]["matrix"], "pan", 7.0 * k) for k in range(5)]
    pending = []
    for k, (pos, mat) in enumerate(cams):
        t = ring[k % 2]
        if len(pending) == 2:                                             # the ring is full: collect the oldest frame first
            j = pending.pop(0)
            ring[j % 2].wait()
            o.set_camera(*cams[j])
            want, _ = o.render(depth)
            assert_same_floats(bufs[j % 2][0], want, "async frame %d" % j)
            assert np.array_equal(bufs[j % 2][1].astype(np.uint16), oracle.quantize(want))
        t.set_camera(pos, mat)
        t.render_async(opts, rgb=bufs[k % 2][0], rgb8=bufs[k % 2][1])
        pending.append(k)
    for j in pending:
        ring[j % 2].wait()
        o.set_camera(*cams[j])
        want, _ = o.render(depth)
        assert_same_floats(bufs[j % 2][0], want, "async frame %d" % j)
    assert ring[0].stats().pixels == H * W and ring[0].stats().kernel_ms > 0


def test_top_level_nodes_interleaved_with_a_mesh_tree(pkg, scenes, oracle):
    """crt_scene_desc only asks for forward links: the top-level tree's nodes need not be one index range.  With a mesh tree spliced
    into the middle of them there is no plan table (it is read off that range) -- the frame must come from the faithful kernels
    and still be the oracle's, and the counters the oracle's."""
    import ctypes as C
    scene, depth, _ = small_case(scenes, "hw11")
    hs = pkg.Scene(json_text=scenes.to_json(scene))
    d = hs.desc
    n = d.n_nodes
    nodes = [d.nodes[i] for i in range(n)]
    roots = sorted([d.meshes[m].root for m in range(d.n_meshes)] + [d.top_root, n])
    top_lo, top_hi = d.top_root, roots[roots.index(d.top_root) + 1]          # the top-level tree's range
    assert top_hi - top_lo >= 3
    m0 = min(range(d.n_meshes), key=lambda m: d.meshes[m].root if d.meshes[m].root >= top_hi else 1 << 40)
    m_lo = d.meshes[m0].root
    m_hi = roots[roots.index(m_lo) + 1]
    assert m_lo >= top_hi                                                     # (the host stores the top-level tree first)
    cut = top_lo + 2
    order = list(range(0, cut)) + list(range(m_lo, m_hi)) + list(range(cut, m_lo)) + list(range(m_hi, n))
    assert sorted(order) == list(range(n))
    new_of = {old: new for new, old in enumerate(order)}
    END, LEAF = 0xFFFFFFFF, 0x80000000

    def remap(link, is_link):
        if link == END or (is_link and (link & LEAF)):
            return link
        return new_of[link]

    arr = (pkg.Node * n)()
    for new, old in enumerate(order):
        src = nodes[old]
        arr[new].lo[:] = src.lo[:]
        arr[new].hi[:] = src.hi[:]
        arr[new].miss = remap(src.miss, False)
        arr[new].link = remap(src.link, True)
    meshes = (pkg.MeshRec * d.n_meshes)()
    for m in range(d.n_meshes):
        meshes[m].root, meshes[m].material, meshes[m].flags, meshes[m].pad = new_of[d.meshes[m].root], d.meshes[m].material, d.meshes[m].flags, 0
    d2 = pkg.SceneDesc()
    C.memmove(C.byref(d2), C.byref(d), C.sizeof(d2))
    d2.nodes = C.cast(arr, C.POINTER(pkg.Node))
    d2.meshes = C.cast(meshes, C.POINTER(pkg.MeshRec))
    d2.top_root = new_of[d.top_root]
    L = pkg.lib()
    ctx = C.c_void_p()
    rc = L.crt_create(C.byref(d2), 0, C.byref(ctx))
    assert rc == 0, L.crt_last_error(None)
    try:
        pos, mat = hs.camera()
        assert L.crt_set_camera(ctx, pos.ctypes.data_as(C.c_void_p), mat.ctypes.data_as(C.c_void_p)) == 0
        want, counters = oracle.OracleScene(scenes.to_blob(scene)).render(depth)
        rect = pkg.Rect(0, 0, hs.width, hs.height)
        for counting in (False, True):
            got = np.zeros((hs.height, hs.width, 3), dtype=np.float32)
            o = pkg.make_options(depth, counters=counting)
            assert L.crt_render(ctx, C.byref(o), C.byref(rect), 1, got.ctypes.data_as(C.c_void_p)) == 0, L.crt_last_error(ctx)
            assert_same_floats(got, want, "interleaved layout, counting=%r" % counting)
        st = pkg.Stats()
        assert L.crt_get_stats(ctx, C.byref(st)) == 0
        assert st.counters() == counters
    finally:
        L.crt_destroy(ctx)


@pytest.mark.gpu
def test_tuning_that_the_kernels_cannot_run_is_refused(pkg, scenes):
    """crt_create_tuned: a cursor chunk below 64 (a refill of a whole wave must fit in one claim, kernel_stream.h: wave_fetch_chunked)
    or not a multiple of 64, and an unknown mode."""
    scene, _, _ = small_case(scenes, "hw07")
    hs = pkg.Scene(json_text=scenes.to_json(scene))
    for bad in (dict(fetch_chunk=0x400020), dict(fetch_chunk=0x100), dict(fetch_chunk=0x410100), dict(mode=7)):
        with pytest.raises(pkg.CrtError) as e:
            pkg.Tracer(hs, tuning=pkg.make_tuning(**bad))
        assert e.value.code == pkg.CRT_ERR_INVALID, bad


@pytest.mark.gpu
def test_a_non_tree_mode_says_that_it_renders_with_tree_semantics(pkg, scenes):
    """RayTracer.cpp:459-478: the reference's brute-force / single-box modes accept a handful of hits its tree modes drop; this
    path renders every mode with the tree's semantics -- same pixels as the BVH modes -- and says so instead of staying silent."""
    scene, depth, _ = small_case(scenes, "hw07")
    tracer = make_tracer(pkg, scenes, scene)
    tree = tracer.render(max_depth=depth, optimization=pkg.OPT_BVH).copy()
    assert tracer.note() == ""
    for mode in (0, 4):   # NoOptimization, AABB
        got = tracer.render(max_depth=depth, optimization=mode)
        assert_same_floats(got, tree, "mode %d" % mode)
        assert "tree" in tracer.note() and "RenderOptimization %d" % mode in tracer.note()
