"""GPU tree build (crt_build_tree_device) against the host builder, which tests/test_host_layer.py pins to the oracle's --
and through the golden frames to the reference's own -- builder: same creation-order numbering, same boxes bit for bit,
same leaf lists (KDTree.cpp:10-46, BoundingBox.h:60-83)."""
import numpy as np
import pytest

from helpers import small_case

pytestmark = pytest.mark.gpu


def _trees(scene_handle):
    out = []
    for m in range(scene_handle.mesh_count):
        boxes, links, idx = scene_handle.tree(m)
        out.append((boxes.view(np.uint32).copy(), links.copy(), idx.copy()))
    boxes, links, idx = scene_handle.tree(-1)
    out.append((boxes.view(np.uint32).copy(), links.copy(), idx.copy()))
    return out


@pytest.mark.parametrize("name,detail", [("hw07", 1.0), ("hw08", 1.0), ("hw11", 1.0), ("hw14", 0.3), ("hw12", 0.5)])
def test_device_built_trees_equal_the_host_builders(pkg, scenes, name, detail, tmp_path):
    # every mesh of 4096 triangles and more goes through the GPU builder, the small ones (and HW07 / HW08 whole) stay on the host
    scene = scenes.make(name, detail=detail, **({"bitmap_size": 64} if name == "hw12" else {}))
    folder = str(tmp_path)
    if scene.get("textures"):
        scenes.write_bitmaps(scene, folder)
    text = scenes.to_json(scene)
    host = pkg.Scene(json_text=text, folder=folder)
    dev = pkg.Scene(json_text=text, folder=folder, build_device=0)
    big = [m for m in range(host.mesh_count) if len(scene["objects"][m]["triangles"]) >= 4096]
    assert big or name in ("hw07", "hw08")
    for m, (a, b) in enumerate(zip(_trees(host), _trees(dev))):
        assert a[0].shape == b[0].shape, (name, m)
        assert np.array_equal(a[0], b[0]), (name, m, "boxes")
        assert np.array_equal(a[1], b[1]), (name, m, "links")
        assert np.array_equal(a[2], b[2]), (name, m, "leaf lists")
    assert np.array_equal(host.flat_nodes(), dev.flat_nodes())             # hence the same flattened scene


def test_device_built_scene_renders_the_same_frame(pkg, scenes, oracle):
    scene = scenes.make("hw14", width=192, height=108, detail=0.3)
    dev = pkg.Scene(json_text=scenes.to_json(scene), build_device=0)
    got = pkg.Tracer(dev).render(max_depth=8)
    want, _ = oracle.OracleScene(scenes.to_blob(scene)).render(8)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_raw_builder_edge_cases(pkg):
    import ctypes as C
    L = pkg.lib()
    L.crt_build_tree_device.argtypes = [C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
    L.crt_built_tree_node_count.restype = C.c_uint32
    L.crt_built_tree_node_count.argtypes = [C.c_void_p]
    L.crt_built_tree_index_total.restype = C.c_uint64
    L.crt_built_tree_index_total.argtypes = [C.c_void_p]
    L.crt_built_tree_links.restype = C.POINTER(C.c_uint32)
    L.crt_built_tree_links.argtypes = [C.c_void_p]
    L.crt_built_tree_free.argtypes = [C.c_void_p]
    root = np.array([0, 0, 0, 1, 1, 1], np.float32)
    # no elements: a root that is a leaf without indexes
    t = C.c_void_p()
    assert L.crt_build_tree_device(0, None, 0, root.ctypes.data_as(C.c_void_p), 25, 8, C.byref(t)) == 0
    assert L.crt_built_tree_node_count(t) == 1 and L.crt_built_tree_index_total(t) == 0
    L.crt_built_tree_free(t)
    # 20 identical boxes spanning the split plane: both children get all of them at every level down to the depth limit
    boxes = np.tile(np.array([0.4, 0.4, 0.4, 0.6, 0.6, 0.6], np.float32), (20, 1))
    t = C.c_void_p()
    assert L.crt_build_tree_device(0, boxes.ctypes.data_as(C.c_void_p), 20, root.ctypes.data_as(C.c_void_p), 3, 8, C.byref(t)) == 0
    n = L.crt_built_tree_node_count(t)
    links = np.ctypeslib.as_array(L.crt_built_tree_links(t), shape=(n, 4)).copy()
    assert n == 15 and int((links[:, 3] > 0).sum()) == 8 and L.crt_built_tree_index_total(t) == 160   # a full tree of depth 3
    assert links[0, 0] == 1 and links[0, 1] == 8 and links[1, 0] == 2 and links[1, 1] == 5            # pre-order numbering
    L.crt_built_tree_free(t)
