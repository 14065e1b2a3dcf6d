"""Dev helper: time one rank's share of the frame (tiles rank, rank+N, ...) for several world sizes on one GPU.
usage: python tools/rank_time.py hw14 [1,2,4,8]"""
import importlib, sys, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
pkg = importlib.import_module('course-assignment-danielhalachev_amd'); sc = pkg.scenes
tiles = importlib.import_module('course-assignment-danielhalachev_amd.tiles')
name = sys.argv[1] if len(sys.argv) > 1 else 'hw14'
s = sc.make(name); tr = pkg.Tracer(pkg.Scene(json_text=sc.to_json(s)), tuning=pkg.tuning_from_string(__import__('os').environ.get('CRT_TUNING', '')))
depth = sc.CONFIGS[name][3]
opts = pkg.make_options(depth)
dev = torch.device('cuda', 0)
worlds = [int(w) for w in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1, 2, 4, 8]
base_ms = None
for world in worlds:
    per = tiles.tiles_per_rank(tr.width, tr.height, world)
    buf = torch.zeros(per * 192, dtype=torch.float32, device=dev)
    for _ in range(3):
        tr.render_tiles_device(opts, 0, world, buf.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        tr.render_tiles_device(opts, 0, world, buf.data_ptr())
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    base_ms = base_ms or ms * world
    print('world %d: rank 0 renders its share in %.2f ms -> %.1f Mpx/s whole job before the gather, strong-scaling ceiling %.2f' % (world, ms, tr.width * tr.height / ms / 1e3, base_ms / (ms * world)), flush=True)
    print('   phases', ['%.2f' % x for x in tr.kernel_times_ms(1)[-1]])
