"""Dev helper (GPU box): one small case on the bounds-checked build of the filter kernels (crt_tuning::bvh = 2): prints the violations it
recorded (kernel_bvh.h: bvh_at codes) and compares the frame with the oracle."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
pkg = importlib.import_module('course-assignment-danielhalachev_amd'); sc = pkg.scenes
from oracle import oracle_api as oa
from helpers import small_case
for name in sys.argv[1:] or ['hw07']:
    scene, depth, folder = (sc.make(name), sc.CONFIGS[name][3], '/tmp') if os.environ.get('FULL') else small_case(sc, name, '/tmp')
    hs = pkg.Scene(json_text=sc.to_json(scene), folder=folder if scene.get('textures') else '')
    tr = pkg.Tracer(hs, tuning=pkg.tuning_from_string('bvh=' + os.environ.get('BVH', '2')))
    got = tr.render(max_depth=depth).copy()
    c = tr.stream_counts()
    print(name, 'diag', [(k, int(c[400 + 2 * k + 1])) for k in range(16) if c[400 + 2 * k]], 'rays', [int(c[g]) for g in range(depth + 1)], 'evicted', [int(c[128 + g]) for g in range(depth + 1)], flush=True)
    if os.environ.get('FULL'): continue
    want, _ = oa.OracleScene(sc.to_blob(scene)).render(depth)
    print(name, 'differing pixels vs oracle:', int((got.view(np.uint32) != want.view(np.uint32)).any(axis=2).sum()), flush=True)
