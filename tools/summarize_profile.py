#!/usr/bin/env python3
"""Turns gpurun_out/prof_<tag>/ (tools/profile_bench.sh) into the committed summaries under profiles/."""
import csv, glob, json, os, sys, collections
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
lines = [f"# rocprofv3 summary {tag} (MI355X, one GPU)", ""]
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
    lines += ["## `rocprofv3 --kernel-trace --stats -- python3 bench.py --headline-only` (the default run without its extra legs: only the timed headline launches and their warm-up)", "",
              "| kernel | calls | total ms | avg us | % | min us | max us |", "|---|---|---|---|---|---|---|"]
    for r in csv.DictReader(open(ks[0])):
        lines.append(f"| `{r['Name'][:110]}` | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.2f} | {float(r['AverageNs'])/1e3:.2f} | "
                     f"{float(r['Percentage']):.2f} | {float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} |")
    lines.append("")
try:
    b = json.loads(open(f"{src}/bench.json").read())
    lines += ["## `python3 bench.py` line of the same build (un-profiled default run, all legs)", "", "```json", json.dumps(b, indent=1), "```", ""]
except Exception as e:
    lines += [f"(no bench.json: {e})", ""]
agg = collections.defaultdict(list)
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_lds"):
    for f in newest(f"{src}/{d}/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_loglike<1, 64, 8, false" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
if agg:
    lines += ["## PMC, kernel `k_loglike<FAST, workgroup 64, K=8>` (default geometry), 10-evaluation launch (C3: one chain group of the device sampler), mean over 10 launches, separate passes", "",
              "| counter | mean per launch |", "|---|---|"]
    for k in sorted(agg):
        lines.append(f"| {k} | {sum(agg[k])/len(agg[k]):.4g} |")
    fetch = sum(agg["FETCH_SIZE"]) / len(agg["FETCH_SIZE"]) if agg.get("FETCH_SIZE") else None
    write = sum(agg["WRITE_SIZE"]) / len(agg["WRITE_SIZE"]) if agg.get("WRITE_SIZE") else None
    if fetch is not None and write is not None:
        hbm = (2.0 * fetch + write) * 1024.0   # FETCH_SIZE/WRITE_SIZE are in KB; gfx950: FETCH_SIZE reads 1/2 of a wide stream
        lines += ["", f"HBM traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB = **{hbm/1e6:.2f} MB** "
                      f"(algorithmic bytes of the launch: 16 B x 1e5 bins x 10 evaluations = 16 MB; the spectrum is served from L2)."]
        json.dump({"hbm_bytes_per_launch": hbm, "fetch_size_kb": fetch, "write_size_kb": write, "correction": "FETCH_SIZE x2 (gfx950)",
                   "launch": "k_loglike FAST wg=64 K=8, B=10, Nx=1e5", "evaluations": 10}, open("profiles/r01_pmc_traffic.json", "w"), indent=1)
try:
    old = open(f"profiles/{tag}_rocprof_summary.md").read()
    keep = old[old.index("## Other legs"):] if "## Other legs" in old else ""
except Exception:
    keep = ""
open(f"profiles/{tag}_rocprof_summary.md", "w").write("\n".join(lines) + "\n" + ("\n" + keep if keep else ""))
for f in ks:
    import shutil
    shutil.copy(f, f"profiles/{tag}_kernel_stats.csv")
print("\n".join(lines[:40]))
