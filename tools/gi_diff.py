"""Dev helper (GPU box): GI frames of the ray-stream kernels against render_lanes' (crt_tuning::mode = lanes), case by case.
usage: python tools/gi_diff.py [scene]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
pkg = importlib.import_module('course-assignment-danielhalachev_amd'); sc = pkg.scenes
name = sys.argv[1] if len(sys.argv) > 1 else 'hw11'
scene = sc.make(name, width=96, height=64, detail=0.2)
hs = pkg.Scene(json_text=sc.to_json(scene))
a, b = pkg.Tracer(hs, tuning=pkg.tuning_from_string(os.environ.get('CRT_TUNING', ''))), pkg.Tracer(hs, tuning=pkg.make_tuning(mode=1))
for d in (0, 1, 2):
    x, y = a.render(max_depth=d), b.render(max_depth=d)
    print('no GI, depth %d: %d differing pixels' % (d, int((x.view(np.uint32) != y.view(np.uint32)).any(axis=2).sum())), a.kernels(), flush=True)
for depth, n, r in [(0, 0, 1), (1, 1, 1), (2, 2, 2)]:
    o = pkg.make_options(depth, use_gi=True, gi_sample_size=n, rays_per_pixel=r, gi_seed=5)
    x, y = a.render(options=o), b.render(options=o)
    bad = np.argwhere((x.view(np.uint32) != y.view(np.uint32)).any(axis=2))
    print('depth %d samples %d rays/px %d: %d differing pixels of %d%s; fallback frames %d, regrows %d' % (depth, n, r, len(bad), x.shape[0] * x.shape[1],
          '' if not len(bad) else ' first %s stream %s lanes %s' % (bad[0].tolist(), x[tuple(bad[0])], y[tuple(bad[0])]), a.stats().fallback_frames, a.stats().queue_regrows), flush=True)
