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
