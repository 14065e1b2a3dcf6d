"""Dev helper (GPU box): rocprofv3 kernel trace of tools/stream_stats.py, then per-kernel average duration per launch index within a frame.
usage: python tools/trace_levels.py hw14 TAG      (CRT_TUNING is passed through)"""
import csv, glob, os, subprocess, sys
scene, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = os.path.join(root, "gpurun_out", "trace_" + tag)
os.makedirs(d, exist_ok=True)
cmd = ["rocprofv3", "--kernel-trace", "-d", d, "--output-format", "csv", "--", "python3", os.path.join(root, "tools", "stream_stats.py"), scene]
with open(os.path.join(d, "run.log"), "w") as log:
    rc = subprocess.run(cmd, cwd=root, env=dict(os.environ, TMPDIR="/tmp"), stdout=log, stderr=subprocess.STDOUT, timeout=500).returncode
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
if not f:
    print("no trace, rc", rc); sys.exit(1)
rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Start_Timestamp"]))
def short(k):
    k = k[k.find("::") + 2:] if "::" in k else k
    return k.split("(")[0]
# frames: split at stream_resolve
frames, cur = [], []
for r in rows:
    cur.append(r)
    if short(r["Kernel_Name"]).startswith("stream_resolve"):
        frames.append(cur); cur = []
fr = frames[2] if len(frames) > 2 else frames[-1]   # the third frame: queues and launch sizes have settled
t0 = int(fr[0]["Start_Timestamp"])
for r in fr:
    print("%9.3f %9.3f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e6, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, short(r["Kernel_Name"])[:60]))
