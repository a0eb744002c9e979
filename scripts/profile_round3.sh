#!/bin/bash
# One measurement set on the GPU box: counter passes (one rocprofv3 run per pass) for the kernels the bench line reports
# on -- the 10M-triangle scene: k_render_flow at depth 5 (exact and ordered traversal) and at unbounded depth, the
# wavefront kernels at depth 5 and unbounded, k_first_hit; the Cornell box: k_render_pool; the 1M scenes: k_first_hit --
# the counter-derived values bench.py reads (profiles/r03_counters.json is a copy of <out>/counters.json), then the bench
# line and rocprofv3 kernel stats of the same command.
# usage: scripts/profile_round3.sh <tag> [parts]      (results under gpurun_out/<tag>/)
# parts: any of m (10M megakernel passes), w (10M wavefront passes), l (vector memory pipeline, 10M megakernel and Cornell box), c (Cornell box and 1M scenes), b (bench line + kernel stats);
# default mwcb.  The counter file is added to, part by part, so the parts can run in separate calls on the same tag... of the
# same tree: each call starts from the copy kept in profiles/r03_counters.json.
set -e -o pipefail
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd "$root"
parts=${2:-mwcb}
J=$out/counters.json
cp "$root/profiles/r03_counters.json" "$J" 2>/dev/null || rm -f "$J"
S="python3 scripts/bench_synth.py heightfield 10000000 3840 2160"
if [[ $parts == *m* ]]; then
# 10M-triangle heightfield 3840x2160: camera rays (10 + 1 launches of k_first_hit) and 32 samples of k_render_flow per launch
for trav in 0 1; do
  name=$([ $trav = 0 ] && echo exact || echo fast)
  bash scripts/pmc_passes.sh "$out/pmc_hf10m_$name" $S 32 5 4 $trav 32
  python3 scripts/pmc_summary.py "$out/pmc_hf10m_$name" k_render_flow > "$out/pmc_hf10m_${name}_k_render_flow.txt"
  python3 scripts/pmc_summary.py "$out/pmc_hf10m_$name" k_first_hit > "$out/pmc_hf10m_${name}_k_first_hit.txt"
  python3 scripts/pmc_to_json.py "$out/pmc_hf10m_$name" k_render_flow "k_render_flow/heightfield-10m/$name" 32 "$J" "scripts/bench_synth.py heightfield 10000000 3840 2160 32 5 4 $trav 32 (two dispatches of 32 samples)"
  python3 scripts/pmc_to_json.py "$out/pmc_hf10m_$name" k_first_hit "k_first_hit/heightfield-10m/$name" 1 "$J" "same run: 8.3M camera rays per dispatch"
  echo "10M $name megakernel counters done"
done
# unbounded depth (Russian roulette only), megakernel
bash scripts/pmc_passes.sh "$out/pmc_hf10m_unbounded" $S 32 -1 4 0 32
python3 scripts/pmc_summary.py "$out/pmc_hf10m_unbounded" k_render_flow > "$out/pmc_hf10m_unbounded_k_render_flow.txt"
python3 scripts/pmc_to_json.py "$out/pmc_hf10m_unbounded" k_render_flow "k_render_flow/heightfield-10m/unbounded/exact" 32 "$J" "scripts/bench_synth.py heightfield 10000000 3840 2160 32 -1 4 0 32 (two dispatches of 32 samples)"
echo "10M unbounded megakernel counters done"
fi
if [[ $parts == *w* ]]; then
# wavefront integrator (queues + persistent tail): 8 samples warm-up + 8 timed = 16 samples per run, every k_wf_* dispatch summed
for depth in 5 -1; do
  name=$([ $depth = 5 ] && echo exact || echo unbounded/exact)
  dir=$out/pmc_hf10m_wf_$([ $depth = 5 ] && echo depth5 || echo unbounded)
  bash scripts/pmc_passes.sh "$dir" $S 8 $depth 2 0 32
  for k in k_wf_generate k_wf_trace k_wf_shade k_wf_finish; do python3 scripts/pmc_summary.py "$dir" $k > "${dir}_$k.txt"; done
  python3 scripts/pmc_to_json.py "$dir" k_wf_ "k_wf/heightfield-10m/$name" run:16 "$J" "scripts/bench_synth.py heightfield 10000000 3840 2160 8 $depth 2 0 32: all k_wf_* dispatches of the run's 16 samples summed"
  python3 scripts/pmc_to_json.py "$dir" k_wf_trace "k_wf_trace/heightfield-10m/$name" run:16 "$J" "same run, the two traversal kernels (closest hit, any hit) alone"
  python3 scripts/pmc_to_json.py "$dir" k_wf_shade "k_wf_shade/heightfield-10m/$name" run:16 "$J" "same run, the shading kernel alone"
  python3 scripts/pmc_to_json.py "$dir" k_wf_finish "k_wf_finish/heightfield-10m/$name" run:16 "$J" "same run, the persistent finishing kernel alone"
  echo "10M wavefront depth $depth counters done"
done
fi
if [[ $parts == *l* ]]; then
# how busy the vector memory pipeline is: text summaries only, nothing bench.py reads
bash scripts/pmc_l1.sh "$out/l1_hf10m" python3 scripts/bench_synth.py heightfield 10000000 3840 2160 32 5 4 0 32
python3 scripts/pmc_summary.py "$out/l1_hf10m" k_render_flow > "$out/l1_hf10m_k_render_flow.txt"
python3 scripts/pmc_summary.py "$out/l1_hf10m" k_first_hit > "$out/l1_hf10m_k_first_hit.txt"
bash scripts/pmc_l1.sh "$out/l1_cornell" python3 scripts/prof_render.py 128 1024 8 3 0 128
python3 scripts/pmc_summary.py "$out/l1_cornell" k_render_pool > "$out/l1_cornell_k_render_pool.txt"
echo "vector memory pipeline counters done"
fi
if [[ $parts == *c* ]]; then
# Cornell box: launches of 128 samples
bash scripts/pmc_passes.sh "$out/pmc_cornell" python3 scripts/prof_render.py 128 1024 8 3 0 128
python3 scripts/pmc_summary.py "$out/pmc_cornell" k_render_pool > "$out/pmc_cornell_k_render_pool.txt"
python3 scripts/pmc_to_json.py "$out/pmc_cornell" k_render_pool "k_render_pool/cornell-box-glass/exact" 128 "$J" "scripts/prof_render.py 128 1024 8 3 0 128"
echo "cornell counters done"
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
fi
cp "$J" "$root/profiles/r03_counters.json"
cp "$J" "$out/../r03_counters.json"
if [[ $parts == *b* ]]; then
  timeout -k 10 600 python3 bench.py > "$out/bench.json" 2> "$out/bench.err"
  tail -c 600 "$out/bench.json"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py --steps 64 --warmup 32 --no-cpu-baseline --no-lbvh --no-extras > "$out/stats.log" 2>&1
  find "$out/stats" -name "*kernel_stats.csv" -exec cp {} "$out/kernel_stats.csv" \;
  head -4 "$out/kernel_stats.csv" | cut -c1-200
fi
