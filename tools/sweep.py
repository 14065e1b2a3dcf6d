"""Dev helper: run tools/stream_stats.py under several crt_tuning settings, one child process each.
usage: python tools/sweep.py hw14 "quad=1 quad_stack=16" "quad=0" "" ..."""
import os, subprocess, sys
scene = sys.argv[1]
for setting in sys.argv[2:]:
    env = dict(os.environ, CRT_TUNING=setting)
    try:
        out = subprocess.run([sys.executable, 'tools/stream_stats.py', scene], env=env, capture_output=True, text=True, timeout=90)
        lines = out.stdout.strip().splitlines()
        phase = [l for l in lines if l.startswith('phase ms')]
        last = eval(phase[0][len('phase ms'):])[-1] if phase else None
        ev = [l for l in lines if l.startswith('evicted per level') or l.startswith('shadow rays')]
        print(setting, '->', ['%.2f' % x for x in last] if last else out.stderr[-300:], '|', ' ; '.join(ev), flush=True)
    except subprocess.TimeoutExpired:
        print(setting, '-> TIMEOUT', flush=True)
        break
