"""Dev helper (GPU box): kernel + memory-copy trace of bench.py's timed loop; prints, for the last frames, when each frame's first and last
kernel ran and what sat between two frames.  usage: python tools/gap_trace.py [steps]"""
import csv, glob, os, subprocess, sys
steps = sys.argv[1] if len(sys.argv) > 1 else "60"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = os.path.join(root, "gpurun_out", "gap_trace")
os.makedirs(d, exist_ok=True)
cmd = ["rocprofv3", "--kernel-trace", "--memory-copy-trace", "-d", d, "--output-format", "csv", "--", "python3", os.path.join(root, "bench.py"),
       "--steps", steps, "--warmup", "3", "--no-cpu-baseline", "--in-flight", "0", "--no-alone"]
with open(os.path.join(d, "run.log"), "w") as log:
    subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=log, stderr=subprocess.STDOUT, timeout=500)
rows = []
for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]; k = k[k.find("::") + 2:] if "::" in k else k
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), k.split("(")[0][:40]))
for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
rows.sort()
# the last 400 events relative to the first of them
tail = rows[-260:]
t0 = tail[0][0]
for s, e, k in tail:
    print("%10.3f %8.3f  %s" % ((s - t0) / 1e6, (e - s) / 1e6, k))
