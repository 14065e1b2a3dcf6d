"""Dev helper (GPU box): frames per second of rank 0's share of the frame (world = 1, 2, 4, 8 simulated on one GPU) with 1, 2, 3 frames in
flight -- contexts of their own on streams of their own, frame k on context k % K.  usage: python tools/inflight_time.py hw14"""
import importlib, sys, time
sys.path.insert(0, '.')
import torch
pkg = importlib.import_module('course-assignment-danielhalachev_amd'); sc = pkg.scenes
tiles = importlib.import_module('course-assignment-danielhalachev_amd.tiles')
name = sys.argv[1] if len(sys.argv) > 1 else 'hw14'
s = sc.make(name); hs = pkg.Scene(json_text=sc.to_json(s))
depth = sc.CONFIGS[name][3]
opts = pkg.make_options(depth)
dev = torch.device('cuda', 0)
K_MAX = 3
trs = [pkg.Tracer(hs) for _ in range(K_MAX)]
streams = [torch.cuda.Stream(dev) for _ in range(K_MAX)]
for world in (1, 2, 4, 8):
    per = tiles.tiles_per_rank(trs[0].width, trs[0].height, world)
    bufs = [torch.zeros(per * 192, dtype=torch.float32, device=dev) for _ in range(K_MAX)]
    for K in (3, 2, 1, 1):
        def frame(k):
            trs[k % K].render_tiles_device(opts, 0, world, bufs[k % K].data_ptr(), streams[k % K].cuda_stream)
        for k in range(6): frame(k)
        torch.cuda.synchronize()
        n = 60
        t0 = time.perf_counter()
        for k in range(n): frame(k)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        print('world %d, %d in flight: %.2f ms per frame of rank 0 -> %.0f Mpx/s whole job' % (world, K, ms, trs[0].width * trs[0].height / ms / 1e3), flush=True)
