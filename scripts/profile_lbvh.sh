#!/bin/bash
# Per-kernel times and counters of the LBVH build (random soup, 1M and 10M primitives, both sort variants).
# usage: scripts/profile_lbvh.sh <tag>      (results under gpurun_out/<tag>/)
set -e -o pipefail
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 scripts/bench_lbvh.py 1000000 10000000 > "$out/stats.log" 2>&1
find "$out/stats" -name "*kernel_stats.csv" -exec cp {} "$out/lbvh_kernel_stats.csv" \;
find "$out/stats" -name "*kernel_trace.csv" -exec cp {} "$out/lbvh_kernel_trace.csv" \;
python3 - "$out" <<'PY'
import collections, csv, re, sys
out = sys.argv[1]
rows = collections.defaultdict(list)
for r in csv.DictReader(open(out + "/lbvh_kernel_trace.csv")):
    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("cpt::(anonymous namespace)::", "").replace("void ", ""))
    items = int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0))
    rows[(name, items)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1000.0)
with open(out + "/lbvh_kernel_times.txt", "w") as f:
    f.write("# per kernel and launch size: rocprofv3 --kernel-trace over `python3 scripts/bench_lbvh.py 1000000 10000000` (both sort variants), microseconds\n")
    f.write("%-36s %18s %6s %9s %9s\n" % ("kernel", "grid (work-items)", "calls", "avg", "min"))
    for (name, items), v in sorted(rows.items()):
        f.write("%-36s %18d %6d %9.2f %9.2f\n" % (name[:36], items, len(v), sum(v) / len(v), min(v)))
print(open(out + "/lbvh_kernel_times.txt").read())
PY
bash scripts/pmc_passes.sh "$out/pmc_lbvh" python3 scripts/bench_lbvh.py 10000000
for k in k_scatter_wide k_hist k_build_up k_climb k_morton_tile k_scan_bins; do python3 scripts/pmc_summary.py "$out/pmc_lbvh" $k; done > "$out/pmc_lbvh_soup10m.txt"
tail -5 "$out/pmc_lbvh_soup10m.txt"
