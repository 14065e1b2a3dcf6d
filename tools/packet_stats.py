import importlib, sys, ctypes as C, time
sys.path.insert(0,'.')
pkg = importlib.import_module('course-assignment-danielhalachev_amd'); sc = pkg.scenes
name = sys.argv[1] if len(sys.argv)>1 else 'hw14'
s = sc.make(name); hs = pkg.Scene(json_text=sc.to_json(s)); tr = pkg.Tracer(hs)
depth = sc.CONFIGS[name][3]
img = tr.render(max_depth=depth, counters=True)
pk, ln = tr.kernel_counters()
out = (C.c_uint64*3)(); pkg.lib().crt_debug_packet_counters.argtypes=[C.c_void_p, C.POINTER(C.c_uint64)]; pkg.lib().crt_debug_packet_counters(tr.ctx, out)
print('packet counters', pk); print('lane counters', ln)
print('debug counters', out[0], out[1], out[2])
for i in range(3):
    img = tr.render(max_depth=depth); print(tr.kernel_times_ms(1))
