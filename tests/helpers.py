"""Shared helpers for the parity tests."""
import os

import numpy as np


def small_case(scenes, name, tmpdir=None):
    """Reduced-size instance of a BASELINE config: (scene dict, max_depth, scene_folder)."""
    params = {
        "hw07": dict(width=160, height=120, detail=0.5),
        "hw08": dict(width=192, height=108, detail=0.4),
        "hw11": dict(width=192, height=108, detail=0.25),
        "hw14": dict(width=192, height=108, detail=0.04),
        "hw12": dict(width=192, height=108, detail=0.08, bitmap_size=64),
    }[name]
    scene = scenes.make(name, **params)
    depth = scenes.CONFIGS[name][3]
    folder = ""
    if scene.get("textures") and tmpdir is not None:
        folder = str(tmpdir)
        scenes.write_bitmaps(scene, folder)
    return scene, depth, folder


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_same_floats(got, want, what=""):
    g, w = bits(got), bits(want)
    diff = g != w
    # NaNs with different payloads still count as equal colours
    both_nan = np.isnan(got) & np.isnan(want)
    diff &= ~both_nan
    n = int(diff.sum())
    if n:
        idx = np.argwhere(diff)[:5]
        raise AssertionError("%s: %d of %d floats differ, first at %s: got %r want %r" % (
            what, n, diff.size, idx.tolist(), got[tuple(idx[0])], want[tuple(idx[0])]))


def blob_to_scene(blob):
    """Parse a CRTS blob (oracle/scene_blob.h) back into the scene dict of scenes.py."""
    import struct
    p = [0]

    def take(fmt):
        n = struct.calcsize(fmt)
        v = struct.unpack_from(fmt, blob, p[0])
        p[0] += n
        return v

    def arr(dtype, n):
        a = np.frombuffer(blob, dtype=dtype, count=n, offset=p[0]).copy()
        p[0] += a.nbytes
        return a

    assert blob[:4] == b"CRTS"
    p[0] = 4
    (version,) = take("<I")
    assert version == 1
    w, h, bucket = take("<III")
    bg = arr(np.float32, 3)
    cam_pos = arr(np.float32, 3)
    cam_mat = arr(np.float32, 9)
    (n_tex,) = take("<I")
    textures = []
    kinds = {0: "albedo", 1: "edges", 2: "checker", 3: "bitmap"}
    for i in range(n_tex):
        (kind,) = take("<I")
        a = arr(np.float32, 3)
        b = arr(np.float32, 3)
        scalar, bw, bh = take("<fII")
        t = {"name": "tex%d" % i, "type": kinds[kind]}
        if kind == 0:
            t["albedo"] = a
        elif kind == 1:
            t.update(inner_color=a, edge_color=b, edge_width=scalar)
        elif kind == 2:
            t.update(color_A=a, color_B=b, square_size=scalar)
        else:
            t["file_path"] = "/tex%d.ppm" % i
            t["_pixels"] = arr(np.uint8, bw * bh * 3).reshape(bh, bw, 3)
        textures.append(t)
    (n_mat,) = take("<I")
    materials = []
    types = {0: "diffuse", 1: "reflective", 2: "constant", 3: "refractive"}
    for _ in range(n_mat):
        (mtype,) = take("<I")
        albedo = arr(np.float32, 3)
        smooth, ior, tex = take("<Ifi")
        m = {"type": types[mtype], "albedo": ("tex%d" % tex) if tex >= 0 else albedo, "smooth_shading": bool(smooth)}
        if mtype == 3:
            m["ior"] = ior
        materials.append(m)
    (n_lights,) = take("<I")
    lights = []
    for _ in range(n_lights):
        pos = arr(np.float32, 3)
        (inten,) = take("<I")
        lights.append({"intensity": inten, "position": pos})
    (n_mesh,) = take("<I")
    objects = []
    for _ in range(n_mesh):
        mat, nv, nt, has_uv = take("<IIII")
        o = {"material_index": mat, "vertices": arr(np.float32, nv * 3).reshape(-1, 3)}
        if has_uv:
            o["uvs"] = arr(np.float32, nv * 3).reshape(-1, 3)
        o["triangles"] = arr(np.uint32, nt * 3).reshape(-1, 3)
        objects.append(o)
    scene = {"settings": {"background_color": bg, "image_settings": {"width": w, "height": h, "bucket_size": bucket}},
             "camera": {"matrix": cam_mat, "position": cam_pos}, "lights": lights, "materials": materials,
             "objects": objects}
    if textures:
        scene["textures"] = textures
    return scene


def load_golden(name):
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz")
    z = np.load(here)
    out = {"blob": z["blob"].tobytes(), "depth": int(z["depth"]), "rgb": z["rgb"]}
    if "ppm_gz" in z.files:
        import gzip
        out["ppm"] = gzip.decompress(z["ppm_gz"].tobytes())
    return out
