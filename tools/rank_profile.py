"""Dev helper: renders rank 0's share of a world of N (argv[2]) a few times, for rocprofv3 --kernel-trace."""
import importlib, sys
sys.path.insert(0, '.')
import torch
pkg = importlib.import_module('course-assignment-danielhalachev_amd'); sc = pkg.scenes
tiles = importlib.import_module('course-assignment-danielhalachev_amd.tiles')
name, world = sys.argv[1], int(sys.argv[2])
s = sc.make(name); tr = pkg.Tracer(pkg.Scene(json_text=sc.to_json(s)))
opts = pkg.make_options(sc.CONFIGS[name][3])
per = tiles.tiles_per_rank(tr.width, tr.height, world)
buf = torch.zeros(per * 192, dtype=torch.float32, device='cuda:0')
allbuf = torch.zeros(world * per * 192, dtype=torch.float32, device='cuda:0')   # room for every rank's part
frame = torch.zeros(tr.width * tr.height * 3, dtype=torch.float32, device='cuda:0')
for _ in range(4):
    tr.render_tiles_device(opts, 0, world, buf.data_ptr())
    tr.unpack_tiles_device(allbuf.data_ptr(), world, per * 192, frame.data_ptr())   # marks the frame boundary for tools/timeline.py
torch.cuda.synchronize()
