"""Dev helper (GPU box): crt_tuning::autotune -- ms per frame before it starts, while it measures, and once it has settled, against
the same frames with the tuner off.  usage: python tools/autotune_time.py hw14"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module('course-assignment-danielhalachev_amd'); sc = pkg.scenes
tiles = importlib.import_module('course-assignment-danielhalachev_amd.tiles')
name = sys.argv[1] if len(sys.argv) > 1 else 'hw14'
s = sc.make(name)
folder = ''
if s.get('textures'):
    import tempfile
    folder = tempfile.mkdtemp() + '/'; sc.write_bitmaps(s, folder)
hs = pkg.Scene(json_text=sc.to_json(s), folder=folder)
depth = sc.CONFIGS[name][3]
opts = pkg.make_options(depth)
dev = torch.device('cuda', 0)
def timed(tr, buf, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): tr.render_tiles_device(opts, 0, 1, buf.data_ptr())
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
for label, tuning in (('autotune', None), ('off', pkg.make_tuning(autotune=0))):
    tr = pkg.Tracer(hs, tuning=tuning)
    buf = torch.zeros(tiles.tiles_per_rank(tr.width, tr.height, 1) * 192, dtype=torch.float32, device=dev)
    for _ in range(3): tr.render_tiles_device(opts, 0, 1, buf.data_ptr())
    out = []
    for block in range(10):
        out.append('%.2f' % timed(tr, buf, 25))
    print(name, label, 'ms per frame in blocks of 25 frames:', ' '.join(out), '|', tr.kernels()['autotune'], flush=True)
