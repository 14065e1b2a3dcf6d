"""Dev helper (GPU box): per-kernel PMC sums of one counter set over tools/stream_stats.py (3 frames of a full-size scene).
usage: python tools/pmc_quick.py hw14 OUTTAG COUNTER [COUNTER ...]     (CRT_TUNING is passed through)
Counters in their own rocprofv3 run, no trace domains beside them."""
import csv, glob, os, subprocess, sys
scene, tag, counters = sys.argv[1], sys.argv[2], sys.argv[3:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = os.path.join(root, "gpurun_out", "pmcq_" + tag)
os.makedirs(d, exist_ok=True)
cmd = ["rocprofv3", "--pmc"] + counters + ["-d", d, "--output-format", "csv", "--", "python3", os.path.join(root, "tools", "stream_stats.py"), scene]
with open(os.path.join(d, "run.log"), "w") as log:
    rc = subprocess.run(cmd, cwd=root, env=dict(os.environ, TMPDIR="/tmp"), stdout=log, stderr=subprocess.STDOUT, timeout=150).returncode
f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
if not f:
    print("no counter file, rc", rc); sys.exit(1)
acc = {}
for row in csv.DictReader(open(f[0])):
    k = row["Kernel_Name"]; k = k[k.find("::") + 2:] if "::" in k else k; k = k.split("(")[0]
    a = acc.setdefault((k, row["Counter_Name"]), [0.0, set()]); a[0] += float(row["Counter_Value"]); a[1].add(row["Dispatch_Id"])
kern = sorted({k for k, _ in acc})
for k in kern:
    vals = {c: acc[(k, c)][0] / max(len(acc[(k, c)][1]), 1) for (kk, c) in acc if kk == k}
    n = max(len(acc[(k, c)][1]) for (kk, c) in acc if kk == k)
    if max(vals.values()) < 1e5: continue
    print("%-44s x%-4d %s" % (k[:44], n, "  ".join("%s=%.4g" % (c.replace("SQ_", ""), v) for c, v in sorted(vals.items()))), flush=True)
