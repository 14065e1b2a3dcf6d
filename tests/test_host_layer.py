"""Host-side mirror of the reference interface (no GPU): ABI surface, `.crtscene` loader, tree builder and
its flattened form, bucket arithmetic, camera helpers, PPM writer, error behaviour."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from helpers import bits, small_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------------------------------------- ABI
def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(crt_[a-z0-9_]+)\s*\(", text)))


@pytest.mark.parametrize("header", ["crt_hip.h", "crt_host.h"])
def test_library_exports_every_declared_symbol(pkg, header):
    lib = pkg.lib()
    names = declared_functions(header)
    assert len(names) >= 10
    for name in names:
        assert hasattr(lib, name), "%s declared in include/%s is not exported" % (name, header)


def test_the_product_library_exports_no_test_hook(pkg):
    """libcrt_hip.so (what crt_main, crt_animation and bench.py load) is the test flavour minus the hooks."""
    plain = C.CDLL(os.path.join(os.path.dirname(pkg.LIB_PATH), "libcrt_hip.so"))
    for name in declared_functions("crt_hip.h"):
        assert hasattr(plain, name) == (name not in pkg.TEST_HOOK_SYMBOLS), name
    for name in pkg.TEST_HOOK_SYMBOLS:
        assert hasattr(pkg.lib(), name)


def test_python_symbol_lists_match_headers(pkg):
    assert sorted(pkg.DEVICE_SYMBOLS) == declared_functions("crt_hip.h")
    assert sorted(pkg.HOST_SYMBOLS) == sorted(set(declared_functions("crt_host.h")) - set(declared_functions("crt_hip.h")))


def test_struct_sizes_match_the_header(pkg):
    assert C.sizeof(pkg.Node) == 32 and C.sizeof(pkg.Triangle) == 64 and C.sizeof(pkg.MeshRec) == 16
    assert C.sizeof(pkg.MaterialRec) == 32 and C.sizeof(pkg.LightRec) == 16 and C.sizeof(pkg.Rect) == 16


def test_no_device_is_an_error_not_a_fallback(pkg, scenes):
    if pkg.device_count() > 0:
        pytest.skip("a GPU is present")
    scene, _, _ = small_case(scenes, "hw07")
    hs = pkg.Scene(json_text=scenes.to_json(scene))
    with pytest.raises(pkg.CrtError) as e:
        pkg.Tracer(hs)
    assert e.value.code == pkg.CRT_ERR_NO_DEVICE


def test_tuning_defaults(pkg):
    t = pkg.make_tuning()
    assert (t.fetch_chunk & 0xFFFF, t.fetch_chunk >> 16, t.level_queue, t.side_blocks, t.bvh) == (256, 64, 1, 3, 1)


# ---------------------------------------------------------------------------------------------- loader
@pytest.mark.parametrize("name", ["hw07", "hw11", "hw14", "hw12"])
def test_loader_builder_match_the_oracle(pkg, scenes, oracle, name, tmp_path):
    scene, _, folder = small_case(scenes, name, tmp_path)
    hs = pkg.Scene(json_text=scenes.to_json(scene), folder=folder)
    o = oracle.OracleScene(scenes.to_blob(scene))
    assert (hs.width, hs.height, hs.bucket_size) == (o.width, o.height, o.bucket_count)
    pos, mat = hs.camera()
    assert np.array_equal(bits(pos), bits(scene["camera"]["position"]))
    assert np.array_equal(bits(mat), bits(scene["camera"]["matrix"]))
    assert hs.mesh_count == o.mesh_count
    for m in range(-1, hs.mesh_count):  # -1 = the object tree; same topology, same box floats, same leaf lists
        hb, hl, hi = hs.tree(m)
        ob, ol, oi = o.tree(m)
        assert np.array_equal(bits(hb), bits(ob)) and np.array_equal(hl, ol) and np.array_equal(hi, oi), "tree %d" % m
    for m in range(hs.mesh_count):  # Mesh constructor normals (Scene.cpp:5-30)
        hf, hv = hs.mesh_normals(m)
        of, ov = o.mesh_normals(m)
        assert np.array_equal(bits(hf), bits(of)) and np.array_equal(bits(hv), bits(ov))


def reference_leaf_order(links):
    """Leaf nodes in the order the reference's stack DFS reaches them (KDTree.cpp:53-74: push children[0], then
    children[1]; pop -> children[1] first), assuming every box test passes."""
    order, stack = [], [0]
    while stack:
        n = stack.pop()
        c0, c1, _, cnt = (int(v) for v in links[n])
        if cnt:
            order.append(n)
        else:
            if c0 != 0xFFFFFFFF:
                stack.append(c0)
            if c1 != 0xFFFFFFFF:
                stack.append(c1)
    return order


def test_flattened_tree_encodes_the_reference_visit_order(pkg, scenes):
    scene, _, _ = small_case(scenes, "hw14")
    hs = pkg.Scene(json_text=scenes.to_json(scene))
    d = hs.desc
    nodes = hs.flat_nodes()
    miss, link = nodes[:, 3], nodes[:, 7]
    idx = np.arange(d.n_nodes, dtype=np.uint32)
    inner = (link & pkg.LINK_LEAF) == 0
    assert np.all((miss == pkg.LINK_END) | (miss > idx))           # links point forward
    assert np.all((link[inner] == pkg.LINK_END) | (link[inner] > idx[inner]))
    leaf_tris = hs.flat_leaf_triangles()
    tri_base = 0
    roots = [d.meshes[m].root for m in range(d.n_meshes)]
    for m in range(d.n_meshes):
        boxes, links, indexes = hs.tree(m)
        # walk the flattened mesh tree taking every `link` (all boxes pass): leaves must come out in the
        # reference's DFS order with the reference's index lists
        want_leaves = reference_leaf_order(links)
        starts = np.concatenate([[0], np.cumsum(links[:, 3].astype(np.int64))]).astype(np.int64)
        n, got = roots[m], []
        while n != pkg.LINK_END:
            if link[n] & pkg.LINK_LEAF:
                e, lst = int(link[n] & ~np.uint32(pkg.LINK_LEAF)), []
                while True:
                    lst.append(int(leaf_tris[e] & ~np.uint32(pkg.ENTRY_LAST)) - tri_base)
                    if leaf_tris[e] & pkg.ENTRY_LAST:
                        break
                    e += 1
                got.append(lst)
                n = miss[n]
            else:
                n = link[n]
        want = [indexes[int(starts[k]):int(starts[k + 1])].tolist() for k in want_leaves]
        assert got == want, "mesh %d" % m
        nv, nt = C.c_uint32(), C.c_uint32()
        pkg.lib().crt_host_mesh_sizes(hs._h, m, C.byref(nv), C.byref(nt))
        tri_base += nt.value


def test_json_numbers_are_narrowed_like_getfloat(pkg):
    # RapidJSON GetFloat = static_cast<float>(double): 0.1 (double) -> float, not a float parse of "0.1"
    js = ('{"settings":{"background_color":[0.1,0.2,1e-3],"image_settings":{"width":16,"height":8,"bucket_size":2}},'
          '"camera":{"matrix":[1,0,0,0,1,0,0,0,1],"position":[0.30000000000000004,0,5e-1]},"lights":[],'
          '"materials":[{"type":"diffuse","albedo":[1,1,1],"smooth_shading":false}],'
          '"objects":[{"material_index":0,"vertices":[0,0,-1, 1,0,-1, 0,1,-1],"triangles":[0,1,2]}]}')
    hs = pkg.Scene(json_text=js)
    pos, _ = hs.camera()
    assert np.array_equal(bits(pos), bits(np.array([0.30000000000000004, 0.0, 0.5]).astype(np.float32)))
    assert np.array_equal(bits(np.array(list(hs.desc.background))), bits(np.array([0.1, 0.2, 1e-3]).astype(np.float32)))
    assert (hs.width, hs.height, hs.bucket_size) == (16, 8, 2)


def test_default_bucket_size_follows_core_count(pkg):
    js = ('{"settings":{"background_color":[0,0,0],"image_settings":{"width":16,"height":8}},'
          '"camera":{"matrix":[1,0,0,0,1,0,0,0,1],"position":[0,0,0]},"lights":[],"materials":[],"objects":[]}')
    n = os.cpu_count()
    assert pkg.Scene(json_text=js).bucket_size == (1 if n == 1 else n * 6)  # SceneParser.cpp:104-105


@pytest.mark.parametrize("js,code", [
    ("{", 6), ('{"settings": 3}', 6), ("[1,2,3]", 6),
    ('{"settings":{"background_color":[0,0,0],"image_settings":{"width":4,"height":4}},"camera":{"matrix":[1,0,0,0,1,0,0,0,1],'
     '"position":[0,0,0]},"materials":[{"type":"shiny","albedo":[1,1,1],"smooth_shading":false}]}', 6),
    ('{"settings":{"background_color":[0,0,0],"image_settings":{"width":4,"height":4}},"camera":{"matrix":[1,0,0,0,1,0,0,0,1],'
     '"position":[0,0,0]},"materials":[{"type":"diffuse","albedo":[1,1,1],"smooth_shading":false}],'
     '"objects":[{"material_index":0,"vertices":[0,0,0,1,0,0,0,1,0],"triangles":[0,1,7]}]}', 6),
])
def test_malformed_scenes_are_status_codes(pkg, js, code):
    with pytest.raises(pkg.CrtError) as e:
        pkg.Scene(json_text=js)
    assert e.value.code == code


def test_unknown_material_keeps_the_reference_message(pkg):
    js = ('{"settings":{"background_color":[0,0,0],"image_settings":{"width":4,"height":4}},"camera":{"matrix":'
          '[1,0,0,0,1,0,0,0,1],"position":[0,0,0]},"materials":[{"type":"shiny","albedo":[1,1,1],"smooth_shading":false}]}')
    with pytest.raises(pkg.CrtError, match="Invalid material"):  # `throw "Invalid material"`, SceneParser.cpp:237
        pkg.Scene(json_text=js)


def test_png_and_pnm_bitmaps_decode_to_the_same_texels(pkg, scenes, tmp_path):
    import struct
    from PIL import Image
    rng = np.random.RandomState(3)
    px = rng.randint(0, 256, size=(9, 13, 3), dtype=np.uint8)
    Image.fromarray(px).save(str(tmp_path / "t.png"))
    Image.fromarray(px).save(str(tmp_path / "t.bmp"))                          # 24-bit, bottom-up, rows padded to 4 bytes
    Image.fromarray(np.dstack([px, np.full((9, 13), 200, np.uint8)])).save(str(tmp_path / "t32.bmp"))   # 32-bit with alpha (dropped)
    Image.fromarray(px).save(str(tmp_path / "t.tga"))                          # uncompressed true colour, bottom-up
    with open(str(tmp_path / "top.tga"), "wb") as f:                           # top-down 32-bit TGA written by hand
        f.write(struct.pack("<BBBHHBHHHHBB", 0, 0, 2, 0, 0, 0, 0, 0, 13, 9, 32, 0x28))
        f.write(np.dstack([px[:, :, ::-1], np.full((9, 13), 7, np.uint8)]).tobytes())
    with open(str(tmp_path / "t.ppm"), "wb") as f:
        f.write(b"P6\n13 9\n255\n" + px.tobytes())
    base = ('{"settings":{"background_color":[0,0,0],"image_settings":{"width":8,"height":8,"bucket_size":1}},'
            '"camera":{"matrix":[1,0,0,0,1,0,0,0,1],"position":[0,0,0]},"lights":[],'
            '"textures":[{"name":"b","type":"bitmap","file_path":"/%s"}],'
            '"materials":[{"type":"diffuse","albedo":"b","smooth_shading":false}],'
            '"objects":[{"material_index":0,"vertices":[0,0,-1,1,0,-1,0,1,-1],"uvs":[0,0,0,1,0,0,0,1,0],"triangles":[0,1,2]}]}')
    for fn in ("t.png", "t.ppm", "t.bmp", "t32.bmp", "t.tga", "top.tga"):
        hs = pkg.Scene(json_text=base % fn, folder=str(tmp_path))
        d = hs.desc
        assert (d.textures[0].width, d.textures[0].height, d.n_texels) == (13, 9, 117)
        got = np.ctypeslib.as_array(d.texels, shape=(117 * 3,)).reshape(9, 13, 3)
        assert np.array_equal(got, px), fn
    Image.fromarray(px).save(str(tmp_path / "t.jpg"))                          # what this loader leaves to the caller: the error says how
    with pytest.raises(pkg.CrtError, match="stb_image"):
        pkg.Scene(json_text=base % "t.jpg", folder=str(tmp_path))


# ---------------------------------------------------------------------------------------------- buckets / camera / PPM
@pytest.mark.parametrize("w,h,count", [(640, 480, 48), (1920, 1080, 48), (100, 60, 7), (33, 17, 5), (64, 64, 1), (50, 50, 100),
                                       (31, 9, 65536 + 6)])
def test_bucket_rectangles_match_the_oracle(pkg, oracle, w, h, count):
    got = pkg.bucket_rects(w, h, count, pkg.OPT_BVH_BUCKETS_POOL)
    want = oracle.bucket_grid(w, h, count)
    assert np.array_equal(got, want)
    assert np.array_equal(pkg.bucket_rects(w, h, count, pkg.OPT_BVH_BUCKETS_QUEUE), want)  # same set, shuffled in the reference


def test_single_rectangle_modes_and_regions(pkg):
    for mode in (pkg.OPT_NONE, pkg.OPT_AABB, pkg.OPT_BVH):
        assert pkg.bucket_rects(640, 480, 48, mode).tolist() == [[0, 0, 640, 480]]  # RayTracer.cpp:209-215,251-258
    r = pkg.bucket_rects(640, 480, 48, pkg.OPT_REGIONS, hardware_concurrency=8)     # RayTracer.cpp:216-222: 8 regions
    assert len(r) == 8 and r[0].tolist() == [0, 0, 160, 240] and r[5].tolist() == [240, 160, 160, 240]


def test_camera_helpers_follow_the_reference_formulas(pkg):
    libm = C.CDLL("libm.so.6")
    libm.cosf.restype = libm.sinf.restype = C.c_float
    libm.cosf.argtypes = libm.sinf.argtypes = [C.c_float]
    f = np.float32
    deg = f(33.5)
    rad = deg * (f(22) / (f(7) * f(180.0)))  # degreesToRadians uses 22/7 (Camera.cpp:10-12)
    c, s = f(libm.cosf(rad)), f(libm.sinf(rad))
    ident = np.eye(3, dtype=np.float32).ravel()
    pos, mat = pkg.camera_apply([0, 0, 0], ident, "pan", deg)
    assert np.array_equal(bits(mat), bits(np.array([c, 0, -s, 0, 1, 0, s, 0, c], dtype=np.float32)))
    pos, mat2 = pkg.camera_apply([1, 2, 3], mat, "truck", [0.5, 0.25, -1.0])
    d = np.array([0.5, 0.25, -1.0], dtype=np.float32)
    m = mat.reshape(3, 3)
    moved = np.array([f(f(d[0] * m[0, j]) + f(d[1] * m[1, j])) + f(d[2] * m[2, j]) for j in range(3)], dtype=np.float32)
    assert np.array_equal(bits(pos), bits(np.array([1, 2, 3], dtype=np.float32) + moved))
    _, t = pkg.camera_apply([0, 0, 0], ident, "tilt", deg)
    assert np.array_equal(bits(t), bits(np.array([1, 0, 0, 0, c, -s, 0, s, c], dtype=np.float32)))
    _, r = pkg.camera_apply([0, 0, 0], ident, "roll", deg)
    assert np.array_equal(bits(r), bits(np.array([c, -s, 0, s, c, 0, 0, 0, 1], dtype=np.float32)))


def test_camera_helpers_match_the_reference_camera(pkg):
    """tests/golden/camera_ops.npz: a chain of 68 truck / pan / tilt / roll calls on the REAL reference's Camera
    (Camera.cpp:33-70, pi = 22/7) and the camera after each; crt_host_camera_apply must reproduce every bit."""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "camera_ops.npz"))
    pos, mat = z["start"][:3].copy(), z["start"][3:].copy()
    assert set(z["op_names"].tolist()) == {"truck", "pan", "tilt", "roll"}
    for name, arg, want in zip(z["op_names"], z["op_args"], z["after"]):
        pos, mat = pkg.camera_apply(pos, mat, str(name), arg)
        got = np.concatenate([pos, mat])
        assert np.array_equal(bits(got), bits(want)), (str(name), arg.tolist())


def test_ppm_writer_is_byte_identical_to_the_oracle_writer(pkg, oracle, tmp_path):
    rng = np.random.RandomState(1)
    rgb = (rng.rand(37, 53, 3) * 1.3 - 0.15).astype(np.float32)
    rgb[0, 0] = [np.float32(1.0), np.float32(0.0), np.float32(254.5 / 255)]
    a, b = str(tmp_path / "a.ppm"), str(tmp_path / "b.ppm")
    pkg.export_ppm(a, rgb)
    oracle.write_ppm(b, rgb)
    data = open(a, "rb").read()
    assert data == open(b, "rb").read()
    assert data.startswith(b"P3\n53 37\n255\n") and data.endswith(b"\t\n")  # RayTracer.cpp:541-551


def test_unknown_answer_vectors_of_the_box_test(oracle):
    lo, hi = [-1, -1, -1], [1, 1, 1]
    assert oracle.box_hit(lo, hi, [0, 0, 5], [0, 0, -1])          # pointing at the box
    assert oracle.box_hit(lo, hi, [0, 0, 5], [0, 0, 1])           # pointing AWAY: still "hit" (no t >= 0 test, BoundingBox.h:85-108)
    assert not oracle.box_hit(lo, hi, [3, 0, 5], [0, 0, -1])      # parallel axis, origin outside the slab
    assert oracle.box_hit(lo, hi, [1, 0, 5], [1e-8, 0, -1])       # |d.x| < FLT_EPSILON counts as parallel; origin on the face
    assert not oracle.box_hit(lo, hi, [0, 3, 5], [0.0, 0.2, -1])  # misses above
    assert oracle.box_hit([0, -1, 0], [1, -1, 1], [0.5, 0, 0.5], [0, -1, 0])  # zero-thickness box
