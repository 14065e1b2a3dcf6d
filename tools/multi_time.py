"""Dev helper: time whole frames through the host tracer (crt_host_tracer_render, no PPM) for one context and for several
contexts on ONE device (crt_multi: the frame's tiles dealt over the contexts, each with its own streams and queues)."""
import importlib, sys, time, os
sys.path.insert(0, '.')
import numpy as np
pkg = importlib.import_module('course-assignment-danielhalachev_amd'); sc = pkg.scenes
name = sys.argv[1] if len(sys.argv) > 1 else 'hw14'
s = sc.make(name); hs = pkg.Scene(json_text=sc.to_json(s))
depth = sc.CONFIGS[name][3]
tun = pkg.tuning_from_string(os.environ.get('CRT_TUNING', ''))
ref = None
for parts in (1, 2, 3, 4):
    tr = pkg.Tracer(hs, tuning=tun, devices=[0] * parts) if parts > 1 else pkg.Tracer(hs, tuning=tun)
    out = np.zeros((hs.height, hs.width, 3), dtype=np.float32)
    for _ in range(3): tr.render(max_depth=depth, optimization=pkg.OPT_BVH, out=out)
    if ref is None: ref = out.copy()
    same = bool(np.array_equal(ref.view(np.uint32), out.view(np.uint32)))
    n = 10; t0 = time.perf_counter()
    for _ in range(n): tr.render(max_depth=depth, optimization=pkg.OPT_BVH, out=out)
    ms = (time.perf_counter() - t0) / n * 1e3
    st = tr.stats()
    print('%d context(s): %.2f ms per frame incl. the copy to the host; device %.2f ms; same frame %s' % (parts, ms, st.kernel_ms, same), flush=True)
    tr.close()
