"""Dev helper: per-level ray counts / evictions of the ray-stream pass on a full-size BASELINE scene.
CRT_TUNING="quad=0 heavy_level=0" (crt_tuning fields) selects kernels; the tool, not the library, reads it."""
import importlib, sys, ctypes as C
sys.path.insert(0, '.')
pkg = importlib.import_module('course-assignment-danielhalachev_amd'); sc = pkg.scenes
name = sys.argv[1] if len(sys.argv) > 1 else 'hw14'
import tempfile
s = sc.make(name); folder = tempfile.mkdtemp() + '/'
if s.get('textures'): sc.write_bitmaps(s, folder)
hs = pkg.Scene(json_text=sc.to_json(s), folder=folder); tr = pkg.Tracer(hs, tuning=pkg.tuning_from_string(__import__('os').environ.get('CRT_TUNING', '')))
depth = sc.CONFIGS[name][3]
for i in range(3):
    tr.render(max_depth=depth)
print('phase ms', tr.kernel_times_ms(3))
out = (C.c_uint32 * 512)()
L = pkg.lib(); L.crt_debug_stream_counts.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32]
L.crt_debug_stream_counts(tr.ctx, out, 512)
print('rays per level   ', [out[g] for g in range(depth + 2)])
print('evicted per level', [out[128 + g] for g in range(depth + 1)])
print('todo tiles', out[328], 'todo shadow slots', out[329], 'split', out[326])
print('shadow rays', out[320], 'evicted shadow', out[323], 'overflow', out[322], 'guard', out[325])
print('deep queue: level-1 rays', out[332], 'slots', out[1], 'claimed', out[65], 'done', out[384], 'chained rays', out[333], 'nodes', out[416], 'claims that waited', out[448], 'fallback frames', tr.stats().fallback_frames)

# what the wave-per-ray walks of a frame consist of (a render with counters=2 tallies them)
tr.render(max_depth=depth, counters=2)
big = (C.c_uint32 * 1024)()
L.crt_debug_stream_counts(tr.ctx, big, 1024)
for label, o in (('closest-hit', 704), ('shadow', 712)):
    n = max(big[o], 1)
    print('wave-per-ray %s walks: %d rays; per ray %.1f chunk tests (64 boxes each), %.1f triangle batches, %.1f mesh trees entered, %.1f loop trips'
          % (label, big[o], big[o + 1] / n, big[o + 2] / n, big[o + 3] / n, big[o + 4] / n))
