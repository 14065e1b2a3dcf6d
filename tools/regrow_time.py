"""Dev helper (GPU box): a context's first frame when its queues do not fit at once -- the attempts' device times against a
settled frame's (tests/test_gpu_parity.py::test_queue_capacities_follow_the_frames is the same room).
usage: python tools/regrow_time.py [width height]"""
import importlib, sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pkg = importlib.import_module('course-assignment-danielhalachev_amd'); sc = pkg.scenes
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 192)
scene = sc.make("hw11", width=W, height=H, detail=0.15)
mirror = [m["type"] for m in scene["materials"]].index("reflective")
for o in scene["objects"][0:4]:
    o["material_index"] = mirror
hs = pkg.Scene(json_text=sc.to_json(scene))
for trial in range(3):
    tr = pkg.Tracer(hs)
    t0 = time.perf_counter(); tr.render(max_depth=8); wall = (time.perf_counter() - t0) * 1e3
    st = tr.stats(); n = int(st.queue_regrows) + 1
    first = tr.kernel_times_ms(n)
    for _ in range(5): tr.render(max_depth=8)
    t0 = time.perf_counter(); tr.render(max_depth=8); wall_s = (time.perf_counter() - t0) * 1e3
    settled = tr.kernel_times_ms(1)[-1]
    print('trial %d: first call %.2f ms wall, %d attempt(s): %s ms on the device; settled call %.2f ms wall, %.3f ms on the device; fallback frames %d, queues %.1f MB'
          % (trial, wall, n, ' + '.join('%.3f' % t[0] for t in first), wall_s, settled[0], tr.stats().fallback_frames, tr.stats().queue_bytes / 1e6), flush=True)
    print('   phases of the attempts:', [tuple(round(x, 3) for x in t) for t in first], 'settled:', tuple(round(x, 3) for x in settled))
