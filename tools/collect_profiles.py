"""Collects the round's rocprofv3 evidence on the GPU box (run through gpurun) into gpurun_out/profiles_<tag>/:
kernel-trace stats of the default bench command, then separate --pmc passes for the HBM-side counters, and the
derived hbm_traffic entry for bench.py.  usage: python tools/collect_profiles.py r01b"""
import csv, glob, json, os, subprocess, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out", "profiles_" + tag)
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
DOM = "stream_trace_shadow_plan<0"


def run(args, name):
    d = os.path.join(out, name)
    cmd = ["rocprofv3"] + args + ["-d", d, "--output-format", "csv", "--", "python3", os.path.join(root, "bench.py"),
                                  "--steps", "10", "--warmup", "2", "--no-cpu-baseline", "--no-alone", "--in-flight", "0", "--settle", "0"]
    with open(os.path.join(out, name + ".log"), "w") as log:
        rc = subprocess.run(cmd, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT, timeout=400).returncode
    print(name, "rc", rc, flush=True)
    return d


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


stats_only = "--stats-only" in sys.argv
d = run(["--kernel-trace", "--stats"], "stats")
stats = find(d, "kernel_stats.csv")
if stats:
    os.replace(stats, os.path.join(out, tag + "_kernel_stats.csv"))
bench_line = None
for line in open(os.path.join(out, "stats.log")):
    if line.startswith('{"metric"'):
        bench_line = json.loads(line)

pmc = {}
for counters in () if stats_only else (["FETCH_SIZE"], ["WRITE_SIZE"], ["TCC_HIT_sum", "TCC_MISS_sum"], ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum"],
                 ["SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_WAVES", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU"],
                 ["SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES"]):
    name = "pmc_" + "_".join(c.replace("_sum", "") for c in counters)
    d = run(["--pmc"] + counters, name)
    f = find(d, "counter_collection.csv")
    if not f:
        continue
    acc = {}
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        k = k[k.find("::") + 2:] if "::" in k else k
        k = k.split("(")[0]
        key = (k, row["Counter_Name"])
        a = acc.setdefault(key, [0.0, set()])
        a[0] += float(row["Counter_Value"])
        a[1].add(row["Dispatch_Id"])
    for (k, c), (v, ids) in acc.items():
        pmc.setdefault(k, {})[c] = {"per_launch": v / max(len(ids), 1), "launches": len(ids)}

summary = {"bench": bench_line, "pmc_per_launch": pmc}
if stats_only:
    json.dump(bench_line, open(os.path.join(out, tag + "_stats_bench_line.json"), "w"), indent=1)
else:
    json.dump(summary, open(os.path.join(out, tag + "_summary.json"), "w"), indent=1)
dom = [k for k in pmc if k.startswith(DOM)]
if dom and "FETCH_SIZE" in pmc[dom[0]] and "WRITE_SIZE" in pmc[dom[0]]:
    f_kb, w_kb = pmc[dom[0]]["FETCH_SIZE"]["per_launch"], pmc[dom[0]]["WRITE_SIZE"]["per_launch"]
    json.dump({"kernel": dom[0], "hbm_bytes_per_launch": int((2 * f_kb + w_kb) * 1024), "FETCH_SIZE_KB_raw": f_kb, "WRITE_SIZE_KB_raw": w_kb,
               "launches": pmc[dom[0]]["FETCH_SIZE"]["launches"]}, open(os.path.join(out, "hbm_traffic_entry.json"), "w"), indent=1)
print("done", flush=True)
