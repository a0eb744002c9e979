"""Sums rocprofv3 --pmc csv output per kernel: python scripts/pmc_summary.py <outdir> [kernel substring]"""
import csv, glob, os, re, sys, collections
out = sys.argv[1]; want = sys.argv[2] if len(sys.argv) > 2 else ""
tot = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(os.path.join(out, "pass*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(\w+)\([^()]*(\([^()]*\)[^()]*)*\)$", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"]
        if want and want not in k: continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[k][r["Counter_Name"]] += 1
for k in tot:
    print(k)
    for c in sorted(tot[k]):
        print(f"  {c:32s} total {tot[k][c]:.6g}  per-dispatch {tot[k][c]/calls[k][c]:.6g}  dispatches {calls[k][c]}")
