"""GI frames at 480x270 on the full-detail scenes: filter kernels vs reference-order kernels vs (hw11) render_lanes, bit for bit."""
import importlib, sys
import numpy as np
sys.path.insert(0, '.')
pkg = importlib.import_module('course-assignment-danielhalachev_amd'); sc = pkg.scenes
for name, depth in (('hw14', 3), ('hw11', 4), ('hw12', 2)):
    scene = sc.make(name, width=480, height=270)
    import tempfile; folder = tempfile.mkdtemp() + '/'
    if scene.get('textures'): sc.write_bitmaps(scene, folder)
    hs = pkg.Scene(json_text=sc.to_json(scene), folder=folder)
    frames = {}
    for tag, t in (('filter', ''), ('reference-order', 'bvh=0'), ('filter, no side stream', 'side_blocks=0')):
        tr = pkg.Tracer(hs, tuning=pkg.tuning_from_string(t))
        for seed in (5, 6):
            frames[(tag, seed)] = tr.render(options=pkg.make_options(depth, use_gi=True, gi_sample_size=2, rays_per_pixel=2, gi_seed=seed)).copy()
        print(name, tag, 'fallback frames', tr.stats().fallback_frames, 'regrows', tr.stats().queue_regrows, flush=True)
    for seed in (5, 6):
        a = frames[('reference-order', seed)].view(np.uint32)
        for tag in ('filter', 'filter, no side stream'):
            print(name, 'seed', seed, tag, 'differing floats vs reference-order:', int((frames[(tag, seed)].view(np.uint32) != a).sum()), flush=True)
