"""Dev helper (GPU box): ONE frame as two half-frames (even / odd tiles) on two contexts and streams of one device, the second
started `delay` GPU cycles after the first (torch.cuda._sleep), against the frame on one context.  Does one half's latency-bound
head and tail fill the other's?  usage: python tools/stagger_time.py hw14"""
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
trs = [pkg.Tracer(hs) for _ in range(4)]
strs = [torch.cuda.Stream(dev) for _ in range(5)][1:]   # (the pool's first stream shares a hardware queue with a side stream)
def run(parts, delay):
    per = tiles.tiles_per_rank(trs[0].width, trs[0].height, parts)
    bufs = [torch.zeros(per * 192, dtype=torch.float32, device=dev) for _ in range(parts)]
    def frame():
        for p in range(parts):
            with torch.cuda.stream(strs[p]):
                if p and delay: torch.cuda._sleep(int(delay * p))
                trs[p].render_tiles_device(opts, p, parts, bufs[p].data_ptr(), strs[p].cuda_stream)
        torch.cuda.synchronize()
    for _ in range(4): frame()
    n = 20; t0 = time.perf_counter()
    for _ in range(n): frame()
    return (time.perf_counter() - t0) / n * 1e3
print('1 context: %.2f ms' % run(1, 0), flush=True)
for parts in (2, 3, 4):
    for delay in (0, 500_000, 1_000_000, 2_000_000, 3_000_000, 4_000_000):
        print('%d parts, delay %d cycles each: %.2f ms per frame' % (parts, delay, run(parts, delay)), flush=True)
