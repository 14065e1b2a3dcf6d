"""Dev helper (GPU box): the filter kernels (crt_tuning::bvh = 1) against the reference-order kernels (bvh = 0) and, on the small
cases, the oracle: differing pixels, rays and evicted rays per level, frame times.   usage: python tools/bvh_check.py [full]"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
pkg = importlib.import_module('course-assignment-danielhalachev_amd'); sc = pkg.scenes
from oracle import oracle_api as oa
from helpers import small_case
full = len(sys.argv) > 1 and sys.argv[1] == 'full'
extra = os.environ.get('CRT_TUNING', '')


def counts(tr, depth):
    c = tr.stream_counts()
    return 'rays %s shadow %d misses %d miss_hit %d overflow %d' % ([int(c[g]) for g in range(depth + 1)], int(c[320]), int(c[330]), int(c[332]), int(c[322]))


for name in ['hw07', 'hw08', 'hw11', 'hw14', 'hw12']:
    if full:
        scene, depth, folder = sc.make(name), sc.CONFIGS[name][3], '/tmp/bvhchk_' + name
        os.makedirs(folder, exist_ok=True)
        if scene.get('textures'):
            sc.write_bitmaps(scene, folder)
    else:
        scene, depth, folder = small_case(sc, name, '/tmp')
    hs = pkg.Scene(json_text=sc.to_json(scene), folder=folder if scene.get('textures') else '')
    res = {}
    for bvh in (0, 1):
        tr = pkg.Tracer(hs, tuning=pkg.tuning_from_string(('bvh=%d ' % bvh) + extra))
        got = tr.render(max_depth=depth).copy()
        for _ in range(3):
            tr.render(max_depth=depth)
        ms = [t[0] for t in tr.kernel_times_ms(3)]
        res[bvh] = got
        print(name, 'bvh', bvh, tr.kernels().get('level0'), tr.kernels().get('filter') if bvh else '', 'kernel ms %.3f' % min(ms), counts(tr, depth), 'fallbacks', tr.stats().fallback_frames, flush=True)
    bad = np.argwhere((res[0].view(np.uint32) != res[1].view(np.uint32)).any(axis=2))
    print(name, 'bvh=1 vs bvh=0 differing pixels:', len(bad), 'of', res[0].shape[0] * res[0].shape[1], flush=True)
    for (y, x) in bad[:8]:
        print('  pixel', y, x, 'bvh1', res[1][y, x], 'bvh0', res[0][y, x])
    if not full:
        want, _ = oa.OracleScene(sc.to_blob(scene)).render(depth)
        print(name, 'bvh=1 vs oracle differing pixels:', int((res[1].view(np.uint32) != want.view(np.uint32)).any(axis=2).sum()), flush=True)
