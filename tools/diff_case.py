"""Dev helper: render a reduced BASELINE case under several crt_tuning settings ('quad=0 heavy_level=0' ...) and count floats that differ from the oracle."""
import importlib, os, subprocess, sys
if len(sys.argv) > 2 and sys.argv[2] == '--child':
    sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
    import numpy as np
    import torch  # noqa: F401  (HIP runtime load order, see tests/conftest.py)
    pkg = importlib.import_module('course-assignment-danielhalachev_amd'); sc = pkg.scenes
    from oracle import oracle_api as oa
    from helpers import small_case
    import tempfile
    scene, depth, folder = small_case(sc, sys.argv[1], tempfile.mkdtemp())
    tr = pkg.Tracer(pkg.Scene(json_text=sc.to_json(scene), folder=folder), tuning=pkg.tuning_from_string(__import__('os').environ.get('CRT_TUNING', '')))
    got = tr.render(max_depth=depth)
    want, _ = oa.OracleScene(sc.to_blob(scene)).render(depth)
    d = np.argwhere(got.view(np.uint32) != want.view(np.uint32))
    print(len(d), 'floats differ', d[:4].tolist())
else:
    for setting in sys.argv[2:]:
        env = dict(os.environ, CRT_TUNING=setting)
        out = subprocess.run([sys.executable, 'tools/diff_case.py', sys.argv[1], '--child'], env=env, capture_output=True, text=True, timeout=120)
        print(setting, '->', out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)
