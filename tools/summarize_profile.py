#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (tools/profile_bench.sh) into the committed summaries under profiles/:
<tag>_kernel_stats.csv, <tag>_rocprof_summary.md, <tag>_pmc_traffic.json, <tag>_pmc_sq.json."""
import csv, glob, json, os, shutil, sys, collections
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = f"gpurun_out/prof_{tag}"
KERNEL = "k_step<1, 8>"          # the dominant kernel of the headline run: fused step, FAST arithmetic, 8 bins per lane
os.makedirs("profiles", exist_ok=True)
lines = [f"# rocprofv3 summary {tag} (MI355X, one GPU)", ""]
kb = {}


def newest(pattern):
    """gpurun merges new files over old ones (run ids are process ids, not ordered): keep the files of the most recent run."""
    fs = glob.glob(pattern)
    if not fs:
        return []
    rid = lambda f: os.path.basename(f).split("_")[0]
    last = max(fs, key=os.path.getmtime)
    return [f for f in fs if rid(f) == rid(last)]


ks = newest(f"{src}/trace/*/*kernel_stats.csv")
if ks:
    lines += ["## `rocprofv3 --kernel-trace --stats -- python3 bench.py --headline-only`", "",
              "(set-up phase = 1500 burn-in + learning iterations, 1000 of them with adaptation -> `k_iterate`/`k_loglike` lockstep launches; "
              "warm-up 200 + timed 5000 acquire iterations -> two `k_step` launches each, one per chain group on its own stream, or one joint launch when the swap pair straddles the groups)", "",
              "| kernel | calls | total ms | avg us | % | min us | max us |", "|---|---|---|---|---|---|---|"]
    for r in csv.DictReader(open(ks[0])):
        lines.append(f"| `{r['Name'][:110]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.2f} | "
                     f"{float(r['Percentage']):.2f} | {float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} |")
    lines.append("")
    shutil.copy(ks[0], f"profiles/{tag}_kernel_stats.csv")
try:
    b = json.loads(open(f"{src}/bench.json").read())
    kb = b.get("roofline", {}).get("kernel", {})
    lines += ["## `python3 bench.py` line of the same build (un-profiled default run, all legs)", "", "```json", json.dumps(b, indent=1), "```", ""]
except Exception as e:
    lines += [f"(no bench.json: {e})", ""]
agg = collections.defaultdict(list)
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_lds"):
    for f in newest(f"{src}/{d}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
calib = collections.defaultdict(list)
for f in newest(f"{src}/calib_fetch/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        calib[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
if calib:
    lines += ["## FETCH_SIZE calibration (tools/pmc_calib.hip: each launch streams 1 GiB = 1 048 576 KB once)", "", "| kernel | FETCH_SIZE (KB) | bytes read / (FETCH_SIZE x 1024) |", "|---|---|---|"]
    for k, v in calib.items():
        m = sum(v) / len(v)
        lines.append(f"| `{k}` | {m:.0f} | {1048576.0 / m:.3f} |")
    lines += ["", "FETCH_SIZE counts half the bytes for 8-byte-per-lane coalesced loads (this path's x / y reads) exactly as for 16-byte ones: the x2 "
              "correction of MI355X_MICROARCH.md applies unchanged.", ""]
if agg:
    mean = {k: sum(v) / len(v) for k, v in agg.items()}
    lines += [f"## PMC, kernel `{KERNEL}` (fused step: one chain group's likelihood tiles, each deciding the previous iteration for its chain first, + commit workgroups + candidate roles + L z blocks), mean over {len(next(iter(agg.values())))} launches of "
              "`bench.py --headline-only --steps 300`, separate passes", "", "| counter | mean per launch |", "|---|---|"]
    for k in sorted(mean):
        lines.append(f"| {k} | {mean[k]:.4g} |")
    if "FETCH_SIZE" in mean and "WRITE_SIZE" in mean:
        hbm = (2.0 * mean["FETCH_SIZE"] + mean["WRITE_SIZE"]) * 1024.0
        lines += ["", f"HBM traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB = **{hbm/1e6:.2f} MB** against "
                      f"{(kb.get('algorithmic_bytes_per_launch') or 0)/1e6:.1f} MB algorithmic (16 B x 1e5 bins x {kb.get('evaluations_per_launch') or 0:.2f} evaluations "
                      "per launch on average; an iteration is two launches): the spectrum is served from the XCDs' L2s; what reaches the fabric is mostly the "
                      "candidates' tables and background series (written by the roles, read by the next launch's tiles) and the write-through partial sums."]
        json.dump({"hbm_bytes_per_launch": hbm, "fetch_size_kb": mean["FETCH_SIZE"], "write_size_kb": mean["WRITE_SIZE"],
                   "correction": "FETCH_SIZE x2 (gfx950; verified for 8-byte-per-lane loads with tools/pmc_calib.hip in the same session)",
                   "kernel": KERNEL, "evaluations_per_launch": kb.get("evaluations_per_launch"), "algorithmic_bytes_per_launch": kb.get("algorithmic_bytes_per_launch"),
                   "source": f"profiles/{tag}_pmc_traffic.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py --headline-only "
                             f"--steps 300, kernel {KERNEL}, FETCH_SIZE x2 (gfx950, calibrated on 8-byte-per-lane reads); not re-measured live"},
                  open(f"profiles/{tag}_pmc_traffic.json", "w"), indent=1)
    if "SQ_INSTS_VALU" in mean:
        json.dump({"kernel": KERNEL, "SQ_INSTS_VALU_per_launch": mean["SQ_INSTS_VALU"], "SQ_WAVES_per_launch": mean.get("SQ_WAVES"),
                   "SQ_ACTIVE_INST_VALU": mean.get("SQ_ACTIVE_INST_VALU"), "SQ_WAVE_CYCLES": mean.get("SQ_WAVE_CYCLES"), "SQ_WAIT_ANY": mean.get("SQ_WAIT_ANY"),
                   "SQ_WAIT_INST_ANY": mean.get("SQ_WAIT_INST_ANY"), "SQ_BUSY_CYCLES": mean.get("SQ_BUSY_CYCLES")},
                  open(f"profiles/{tag}_pmc_sq.json", "w"), indent=1)
        if mean.get("SQ_WAVE_CYCLES"):
            wc = mean["SQ_WAVE_CYCLES"]
            lines += ["", f"Wave-cycle split: VALU issuing {100*mean.get('SQ_ACTIVE_INST_VALU',0)/wc:.0f} %, waiting on memory/LDS/barrier (SQ_WAIT_ANY) "
                          f"{100*mean.get('SQ_WAIT_ANY',0)/wc:.0f} %, issue stalls (SQ_WAIT_INST_ANY) {100*mean.get('SQ_WAIT_INST_ANY',0)/wc:.0f} %; "
                          f"{mean['SQ_INSTS_VALU']/mean['SQ_WAVES']:.0f} VALU instructions per wave, {mean['SQ_WAVES']:.0f} waves per launch."]
open(f"profiles/{tag}_rocprof_summary.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:30]))
