"""Collects the round's rocprofv3 evidence on the GPU box (run through gpurun) into gpurun_out/profiles_<tag>/:

  <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the DEFAULT bench command (no development flags that change the frame)
  <tag>_kernel_medians.json per kernel: launches, median / mean / min / max duration from the same run's kernel trace
  <tag>_bench_line.json    the JSON line that very run printed (its roofline.kernel_ms = the dominant kernel's MEDIAN there)
  <tag>_pmc.json           per-launch counters of the dominant kernel, one --pmc pass per counter group (separate passes: the
                           blocks have few slots), + the clock from GRBM_GUI_ACTIVE / 8 / the launch's own duration
  hbm_traffic_entry.json   fabric-side bytes per launch: 2 x FETCH_SIZE + WRITE_SIZE (KB), the gfx950 correction of
                           /opt/skills/guides/MI355X_MICROARCH.md "HBM"

usage: python tools/collect_profiles.py r03 [--stats-only]        (copy what is to be judged into profiles/)"""
import csv, glob, json, os, subprocess, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out", "profiles_" + tag)
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
DOM = "bvh_trace_shadow<0u, 0>"     # the dominant kernel (most executed tests): the bulk shadow pass
LEVELS = "bvh_trace_queue<0"     # the frame's critical path below level 0
import importlib
sys.path.insert(0, root)
CSRC = importlib.import_module("course-assignment-danielhalachev_amd").csrc_sha256()
BENCH = ["--steps", "10", "--warmup", "3", "--no-cpu-baseline", "--in-flight", "0"]


def run(args, name, extra=()):
    d = os.path.join(out, name)
    cmd = ["rocprofv3"] + args + ["-d", d, "--output-format", "csv", "--", "python3", os.path.join(root, "bench.py")] + BENCH + list(extra)
    with open(os.path.join(out, name + ".log"), "w") as log:
        try:
            rc = subprocess.run(cmd, cwd="/tmp", env=env, stdout=log, stderr=subprocess.STDOUT, timeout=240).returncode
        except subprocess.TimeoutExpired:
            rc = -1
    print(name, "rc", rc, flush=True)
    return d


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def short(k):
    k = k[k.find("::") + 2:] if "::" in k else k
    return k.split("(")[0]


d = run(["--kernel-trace", "--stats"], "stats")
stats = find(d, "kernel_stats.csv")
if stats:
    os.replace(stats, os.path.join(out, tag + "_kernel_stats.csv"))
bench_line = None
for line in open(os.path.join(out, "stats.log")):
    if line.startswith('{"metric"'):
        bench_line = json.loads(line)
json.dump(bench_line, open(os.path.join(out, tag + "_bench_line.json"), "w"), indent=1)
trace = find(d, "kernel_trace.csv")
if trace:
    # the stats file averages every launch of a kernel in the run -- the timed frames, the one tallying launch (3x as long) and
    # the `alone` measurement's; the MEDIAN per kernel is the in-frame launch, which is what the bench line's kernel_ms reports
    import statistics
    per = {}
    for r in csv.DictReader(open(trace)):
        per.setdefault(short(r["Kernel_Name"]), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    json.dump({k: {"launches": len(v), "median_ms": round(statistics.median(v), 4), "mean_ms": round(statistics.mean(v), 4),
                   "min_ms": round(min(v), 4), "max_ms": round(max(v), 4)} for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))},
              open(os.path.join(out, tag + "_kernel_medians.json"), "w"), indent=1)
if "--stats-only" in sys.argv:
    print("done", flush=True)
    sys.exit(0)

# rocprofv3 serialises launches while it counts: the level queue's launch cannot run BESIDE level 0 there (it would sit out its patience
# and leave the work to its second launch: two launches per frame, one of them idle, averaged into one row).  The counting passes
# therefore put it behind level 0 (crt_tuning::level_queue bit 9): one launch per frame, the kernel with the chip to itself, as every
# counted kernel is.
PMC_EXTRA = ["--no-alone", "--tuning", "level_queue=513"]
GROUPS = [["FETCH_SIZE"], ["WRITE_SIZE"], ["TCC_HIT_sum", "TCC_MISS_sum"], ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_READ_REQ_sum"],
          ["TA_TA_BUSY_sum", "TA_FLAT_READ_WAVEFRONTS_sum"], ["TCP_PENDING_STALL_CYCLES_sum"],
          ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_WAVES", "SQ_THREAD_CYCLES_VALU", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_SMEM"],
          ["SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_ANY"]]
pmc = {}
for counters in GROUPS:
    name = "pmc_" + "_".join(c.replace("_sum", "") for c in counters)[:60]
    d = run(["--pmc"] + counters, name, PMC_EXTRA)
    f = find(d, "counter_collection.csv")
    if not f:
        continue
    acc = {}
    for row in csv.DictReader(open(f)):
        key = (short(row["Kernel_Name"]), row["Counter_Name"])
        a = acc.setdefault(key, [0.0, set()])
        a[0] += float(row["Counter_Value"])
        a[1].add(row["Dispatch_Id"])
    for (k, c), (v, ids) in acc.items():
        pmc.setdefault(k, {})[c] = {"per_launch": v / max(len(ids), 1), "launches": len(ids)}

# the clock the chip holds in this kernel: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / the launch's duration, both from ONE pass
clock = None
clock_levels, ms_levels = None, None
d = run(["--pmc", "GRBM_GUI_ACTIVE", "--kernel-trace"], "pmc_clock", PMC_EXTRA)
fc, ft = find(d, "counter_collection.csv"), find(d, "kernel_trace.csv")
if fc and ft:
    dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(ft)) if short(r["Kernel_Name"]).startswith(DOM)}
    cyc = {r["Dispatch_Id"]: float(r["Counter_Value"]) for r in csv.DictReader(open(fc)) if short(r["Kernel_Name"]).startswith(DOM) and r["Counter_Name"] == "GRBM_GUI_ACTIVE"}
    both = [k for k in dur if k in cyc and dur[k] > 0]
    if both:
        clock = sum(cyc[k] / 8.0 / dur[k] for k in both) / len(both)   # cycles per ns = GHz
        ms_in_pass = sum(dur[k] for k in both) / len(both) / 1e6
    dur2 = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(ft)) if short(r["Kernel_Name"]).startswith(LEVELS)}
    cyc2 = {r["Dispatch_Id"]: float(r["Counter_Value"]) for r in csv.DictReader(open(fc)) if short(r["Kernel_Name"]).startswith(LEVELS) and r["Counter_Name"] == "GRBM_GUI_ACTIVE"}
    both2 = [k for k in dur2 if k in cyc2 and dur2[k] > 0]
    if both2:
        clock_levels = sum(cyc2[k] / 8.0 / dur2[k] for k in both2) / len(both2)
        ms_levels = sum(dur2[k] for k in both2) / len(both2) / 1e6

dom = [k for k in pmc if k.startswith(DOM)]
json.dump({"bench": bench_line, "pmc_per_launch": pmc}, open(os.path.join(out, tag + "_summary.json"), "w"), indent=1)
if dom:
    per = {c: v["per_launch"] for c, v in pmc[dom[0]].items()}
    cfg = bench_line["config"] if bench_line else {"scene": "hw14", "width": 1920, "height": 1080, "max_depth": 8}
    key = "%s_%dx%d_d%d_n1" % (cfg["scene"], cfg["width"], cfg["height"], cfg["max_depth"])
    doc = {"csrc_sha256": CSRC,
           key: {"kernel": dom[0], "per_launch": per, "launches": max(v["launches"] for v in pmc[dom[0]].values()),
                 "clock_ghz": round(clock, 4) if clock else 2.0, "kernel_ms_in_the_clock_pass": round(ms_in_pass, 4) if clock else None,
                 "how": "rocprofv3 --pmc <group> -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --in-flight 0 --no-alone --tuning level_queue=513 (the level queue's launch behind level 0: launches are serialised while counting), one pass per group "
                        "(tools/collect_profiles.py); counts averaged over the launches of the kernel in the pass"}}
    lev = [k for k in pmc if k.startswith(LEVELS)]
    if lev and clock_levels:
        doc[key + "|levels"] = {"kernel": lev[0], "per_launch": {c: v["per_launch"] for c, v in pmc[lev[0]].items()},
                                "launches": max(v["launches"] for v in pmc[lev[0]].values()), "clock_ghz": round(clock_levels, 4),
                                "kernel_ms_in_the_clock_pass": round(ms_levels, 4)}
    json.dump(doc, open(os.path.join(out, tag + "_pmc.json"), "w"), indent=1)
    if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
        json.dump({key: {"csrc_sha256": CSRC, "kernel": dom[0], "hbm_bytes_per_launch": int((2 * per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024), "FETCH_SIZE_KB_raw": per["FETCH_SIZE"],
                         "WRITE_SIZE_KB_raw": per["WRITE_SIZE"], "correction": "2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE tallies 128-B requests at 64 B)"}},
                  open(os.path.join(out, "hbm_traffic_entry.json"), "w"), indent=1)
print("done", flush=True)
