#!/bin/bash
# One measurement set on the GPU box: the bench line, rocprofv3 kernel stats of the same command, counter passes for the
# bounce-loop kernels (Cornell box and the 10M-triangle scene, exact and fast traversal) and the batch query kernel, and
# (also on the 1M-triangle heightfield and soup), and the counter-derived values bench.py reads (profiles/r02_counters.json is a copy of <out>/counters.json).
# usage: scripts/profile_round2.sh <tag> [skip-bench]      (results under gpurun_out/<tag>/)
set -e -o pipefail
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root"
J=$out/counters.json
rm -f "$J"
# Cornell box, the kernel of the headline line: launches of 128 samples
bash scripts/pmc_passes.sh "$out/pmc_cornell" python3 scripts/prof_render.py 128 1024 8 3 0 128
python3 scripts/pmc_summary.py "$out/pmc_cornell" k_render_pool > "$out/pmc_cornell_k_render_pool.txt"
python3 scripts/pmc_to_json.py "$out/pmc_cornell" k_render_pool "k_render_pool/cornell-box-glass/exact" 128 "$J" "scripts/prof_render.py 128 1024 8 3 0 128"
echo "cornell counters done"
# 10M-triangle heightfield 3840x2160: camera rays (10 + 1 launches of k_first_hit) and 32 samples of k_render_flow per launch, exact then fast
for trav in 0 1; do
  name=$([ $trav = 0 ] && echo exact || echo fast)
  bash scripts/pmc_passes.sh "$out/pmc_hf10m_$name" python3 scripts/bench_synth.py heightfield 10000000 3840 2160 32 5 4 $trav 32
  python3 scripts/pmc_summary.py "$out/pmc_hf10m_$name" k_render_flow > "$out/pmc_hf10m_${name}_k_render_flow.txt"
  python3 scripts/pmc_summary.py "$out/pmc_hf10m_$name" k_first_hit > "$out/pmc_hf10m_${name}_k_first_hit.txt"
  python3 scripts/pmc_to_json.py "$out/pmc_hf10m_$name" k_render_flow "k_render_flow/heightfield-10m/$name" 32 "$J" "scripts/bench_synth.py heightfield 10000000 3840 2160 32 5 4 $trav 32 (two dispatches of 32 samples)"
  python3 scripts/pmc_to_json.py "$out/pmc_hf10m_$name" k_first_hit "k_first_hit/heightfield-10m/$name" 1 "$J" "same run: 8.3M camera rays per dispatch"
  echo "10M $name counters done"
done
# 1M-triangle scenes 1920x1080: camera rays only (the bounce loop runs two samples to keep the passes short)
for kind in heightfield soup; do
  for trav in 0 1; do
    name=$([ $trav = 0 ] && echo exact || echo fast)
    bash scripts/pmc_passes.sh "$out/pmc_${kind}1m_$name" python3 scripts/bench_synth.py $kind 1000000 1920 1080 2 5 4 $trav 2
    python3 scripts/pmc_summary.py "$out/pmc_${kind}1m_$name" k_first_hit > "$out/pmc_${kind}1m_${name}_k_first_hit.txt"
    python3 scripts/pmc_to_json.py "$out/pmc_${kind}1m_$name" k_first_hit "k_first_hit/${kind}-1m/$name" 1 "$J" "scripts/bench_synth.py $kind 1000000 1920 1080 2 5 4 $trav 2: 2.07M camera rays per dispatch"
  done
  echo "1M $kind counters done"
done
cp "$J" "$root/profiles/r02_counters.json"
if [ -z "$2" ]; then
  timeout -k 10 600 python3 bench.py > "$out/bench.json" 2> "$out/bench.err"
  tail -c 600 "$out/bench.json"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py --steps 256 --warmup 128 --no-cpu-baseline --no-lbvh --no-extras > "$out/stats.log" 2>&1
  find "$out/stats" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \;
  head -4 "$out/kernel_stats.csv" | cut -c1-200
fi
