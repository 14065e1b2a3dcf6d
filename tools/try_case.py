"""Dev helper: render one small case under CRT_TUNING (crt_tuning fields, e.g. 'mode=2') and compare with the oracle."""
import importlib, sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
pkg = importlib.import_module('course-assignment-danielhalachev_amd'); sc = pkg.scenes
from oracle import oracle_api as oa
from helpers import small_case
name = sys.argv[1]
scene, depth, folder = small_case(sc, name, '/tmp')
hs = pkg.Scene(json_text=sc.to_json(scene), folder=folder); tr = pkg.Tracer(hs, tuning=pkg.tuning_from_string(__import__('os').environ.get('CRT_TUNING', '')))
t = time.time(); got = tr.render(max_depth=depth); print('render s', time.time() - t, flush=True)
want, _ = oa.OracleScene(sc.to_blob(scene)).render(depth)
print(name, 'differing floats:', int((got.view(np.uint32) != want.view(np.uint32)).sum()), 'times', tr.kernel_times_ms(1), flush=True)
import ctypes as C
out = (C.c_uint32 * 512)()
L = pkg.lib(); L.crt_debug_stream_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32]
L.crt_debug_stream_counts(tr.ctx, out, 512)
_, cnt = oa.OracleScene(sc.to_blob(scene)).render(depth)
print('shadow rays queued', out[320], 'oracle light evals', cnt['light_evals'], 'todo tiles', out[328], 'todo shadow', out[329], 'split', out[326], flush=True)
