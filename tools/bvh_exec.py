"""Dev helper (GPU box): executed box / triangle tests of a full-size frame, filter kernels vs reference-order kernels."""
import importlib, os, sys
sys.path.insert(0, '.')
pkg = importlib.import_module('course-assignment-danielhalachev_amd'); sc = pkg.scenes
name = sys.argv[1] if len(sys.argv) > 1 else 'hw14'
w = int(sys.argv[2]) if len(sys.argv) > 2 else None
scene = sc.make(name) if w is None else sc.make(name, width=w, height=w * 9 // 16)
depth = sc.CONFIGS[name][3]
hs = pkg.Scene(json_text=sc.to_json(scene))
for bvh in (0, 1):
    tr = pkg.Tracer(hs, tuning=pkg.tuning_from_string('bvh=%d %s' % (bvh, os.environ.get('CRT_TUNING', ''))))
    tr.render(max_depth=depth)
    tr.render(max_depth=depth, counters=2)
    e = tr.executed_counters()
    c = tr.stream_counts()
    rays0 = int(c[327]); 
    c2 = tr.stream_counts(); print('  shadow walks: longest', int(c2[440]), 'steps (ray', int(c2[441]), ') total/16', int(c2[442]), 'walks >= 1024 steps', int(c2[443]))
    print('  closest-hit walks per level (longest, mean steps, walks):', [(int(c2[448 + 3 * g]), round(int(c2[449 + 3 * g]) / max(int(c2[450 + 3 * g]), 1), 1), int(c2[450 + 3 * g])) for g in range(depth + 1)])
    print(name, 'bvh', bvh, e, 'pass-0 slots', rays0, 'per slot: box %.1f tri %.2f' % (e['shadow_pass0_box_tests'] / max(rays0, 1), e['shadow_pass0_tri_tests'] / max(rays0, 1)), 'phase ms', tr.kernel_times_ms(1), flush=True)
