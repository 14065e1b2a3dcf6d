import importlib, os, sys
import numpy as np
import ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
pkg = importlib.import_module('course-assignment-danielhalachev_amd'); sc = pkg.scenes
from helpers import small_case
scene, depth, folder = small_case(sc, sys.argv[1], '/tmp')
res = {}
for bfs in (0, 1):
    tr = pkg.Tracer(pkg.Scene(json_text=sc.to_json(scene), folder=folder), tuning=pkg.make_tuning(bfs=bfs, autotune=0))
    tr.render(max_depth=0)
    out = (C.c_uint32 * 512)()
    L = pkg.lib(); L.crt_debug_stream_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32]
    L.crt_debug_stream_counts(tr.ctx, out, 512)
    n = out[128]
    hits = np.zeros((n, 4), np.float32); ids = np.zeros(n, np.uint32)
    L.crt_debug_read_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    L.crt_debug_read_hits(tr.ctx, hits.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p), n)
    res[bfs] = {int(i): tuple(h.view(np.uint32)) for i, h in zip(ids, hits)}
    print('bfs', bfs, 'evicted', n)
a, b = res[0], res[1]
print('same ids', set(a) == set(b))
bad = [i for i in a if i in b and a[i] != b[i]]
print('differing records', len(bad))
for i in bad[:16]:
    ha, hb = a[i], b[i]
    f = lambda u: np.array([u], np.uint32).view(np.float32)[0]
    print(' ray', i, 'heavy t=%g tri=%d mesh=%d have=%d' % (f(ha[0]), ha[1], ha[2], ha[3]), '| bfs t=%g tri=%d mesh=%d have=%d' % (f(hb[0]), hb[1], hb[2], hb[3]))
