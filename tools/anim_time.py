"""Dev helper (GPU box): frames per second of the camera-orbit driver on a full-size scene, 1 / 2 / 3 frames in flight, with
and without the PPM files.  usage: python tools/anim_time.py hw14 [frames]"""
import importlib, json, os, subprocess, sys, tempfile
sys.path.insert(0, '.')
pkg = importlib.import_module('course-assignment-danielhalachev_amd'); sc = pkg.scenes
name = sys.argv[1] if len(sys.argv) > 1 else 'hw14'
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 60
d = tempfile.mkdtemp(prefix='crt_anim_')
open(os.path.join(d, 'scene.crtscene'), 'w').write(sc.to_json(sc.make(name)))
exe = os.path.join('course-assignment-danielhalachev_amd', 'crt_animation')
for k, extra in ((1, ['--no-ppm']), (2, ['--no-ppm']), (3, ['--no-ppm']), (2, ['--writers', '8'])):
    r = subprocess.run([os.path.abspath(exe), 'scene.crtscene', os.path.join(d, 'f'), '--depth', '5', '--fps', str(frames), '--seconds', '1',
                        '--in-flight', str(k)] + extra, cwd=d, capture_output=True, text=True, timeout=600)
    print(k, extra, r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:], flush=True)
    for f in os.listdir(d):
        if f.endswith('.ppm'): os.remove(os.path.join(d, f))
