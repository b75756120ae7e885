"""Mean of every counter per kernel name over the counter_collection.csv files under a rocprofv3 output directory.
python tools/pmc_mean.py <dir> [kernel-name substring]"""
import collections, csv, glob, sys
d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(list)
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print(f"{k:60s} {c:24s} n={len(v):6d} mean={sum(v) / len(v):14.1f}")
