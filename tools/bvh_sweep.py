"""Dev helper (GPU box): frame times of full-size scenes under a list of crt_tuning settings.
usage: python tools/bvh_sweep.py hw14 "side_blocks=2" "side_blocks=3" ..."""
import importlib, os, sys, statistics
import numpy as np
sys.path.insert(0, '.')
pkg = importlib.import_module('course-assignment-danielhalachev_amd'); sc = pkg.scenes
name = sys.argv[1]
scene = sc.make(name); depth = sc.CONFIGS[name][3]
folder = '/tmp/sweep_' + name; os.makedirs(folder, exist_ok=True)
if scene.get('textures'): sc.write_bitmaps(scene, folder)
hs = pkg.Scene(json_text=sc.to_json(scene), folder=folder if scene.get('textures') else '')
ref = None
for t in sys.argv[2:] or ['']:
    tr = pkg.Tracer(hs, tuning=pkg.tuning_from_string(t))
    got = tr.render(max_depth=depth).copy()
    if ref is None: ref = got
    same = bool(np.array_equal(got.view(np.uint32), ref.view(np.uint32)))
    for _ in range(12): tr.render(max_depth=depth)
    ms = tr.kernel_times_ms(10)
    print('%-6s %-40s total %.3f (min %.3f)  levels %.3f  shadow0 %.3f  tail %.3f  same=%s fallbacks=%d' % (
        name, t or 'defaults', statistics.median(m[0] for m in ms), min(m[0] for m in ms), statistics.median(m[1] for m in ms),
        statistics.median(m[2] for m in ms), statistics.median(m[3] + m[4] for m in ms), same, tr.stats().fallback_frames), flush=True)
