#!/bin/bash
# One round's measurement set on the GPU box: bench line, rocprofv3 kernel stats of the same
# command, counter passes for the bounce-loop kernel, HBM traffic per launch.
# usage: scripts/profile_round.sh <tag>      (results under gpurun_out/<tag>/)
set -e -o pipefail
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root"
timeout -k 10 400 python3 bench.py > "$out/bench.json" 2> "$out/bench.err"
tail -c 1500 "$out/bench.json"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py --steps 256 --warmup 128 --no-cpu-baseline --no-lbvh > "$out/stats.log" 2>&1
find "$out/stats" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \;
head -4 "$out/kernel_stats.csv" | cut -c1-200
bash scripts/pmc_passes.sh "$out/pmc" python3 scripts/prof_render.py 128 1024 8 3 0 128
python3 scripts/pmc_summary.py "$out/pmc" k_render_pool > "$out/pmc_k_render_pool.txt"
python3 - "$out" <<'PY'
import json, re, sys
out = sys.argv[1]
vals = {}
for line in open(out + "/pmc_k_render_pool.txt"):
    m = re.match(r"\s+(\w+)\s+total \S+\s+per-dispatch (\S+)", line)
    if m:
        vals[m.group(1)] = float(m.group(2))
# FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE tallies 128-byte read requests at 64 bytes
traffic = (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0
json.dump({"kernel": "k_render_pool", "hbm_bytes_per_launch": int(traffic), "fetch_size_kb_raw": vals["FETCH_SIZE"], "write_size_kb": vals["WRITE_SIZE"],
           "samples_per_launch": 128, "note": "per dispatch of k_render_pool (128 samples), cornell-box-glass 1024x1024 depth 8; 2 x FETCH_SIZE + WRITE_SIZE"}, open(out + "/traffic.json", "w"))
print(open(out + "/traffic.json").read())
PY
