"""Dev helper: render one small case under CRT_TUNING and list where it differs from the oracle."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
pkg = importlib.import_module('course-assignment-danielhalachev_amd'); sc = pkg.scenes
from oracle import oracle_api as oa
from helpers import small_case
import ctypes as C
name = sys.argv[1]
scene, depth, folder = small_case(sc, name, '/tmp')
tr = pkg.Tracer(pkg.Scene(json_text=sc.to_json(scene), folder=folder), tuning=pkg.tuning_from_string(os.environ.get('CRT_TUNING', '')))
got = tr.render(max_depth=depth).copy()
want, _ = oa.OracleScene(sc.to_blob(scene)).render(depth)
bad = np.argwhere((got.view(np.uint32) != want.view(np.uint32)).any(axis=2))
print(name, 'depth', depth, 'differing pixels:', len(bad), 'of', got.shape[0] * got.shape[1], 'fallbacks', tr.stats().fallback_frames, tr.kernels())
for (y, x) in bad[:12]:
    print('  pixel', y, x, 'got', got[y, x], 'want', want[y, x])
out = (C.c_uint32 * 512)()
L = pkg.lib(); L.crt_debug_stream_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32]
L.crt_debug_stream_counts(tr.ctx, out, 512)
print('rays per level', [out[g] for g in range(depth + 2)], 'evicted', [out[128 + g] for g in range(depth + 1)], 'shadow', out[320], 'evicted shadow', out[323], 'split', out[329], 'overflow', out[322])
