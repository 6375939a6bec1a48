#!/bin/bash
# Regenerates the committed evidence set for one round tag (run on the GPU box, e.g. through gpurun):
#   tools/profile_round.sh r01_k     ->  gpurun_out/<tag>_{bench.json,bench_under_rocprof.json,kernel_stats_default_bench_graph.txt,timeline_one_step.txt}
# (copy them into profiles/ afterwards).  PMC passes (FETCH_SIZE / WRITE_SIZE -> traffic_latest.json) are separate runs, see
# tools/make_traffic_json.py; they must not be combined with trace domains other than --kernel-trace.
set -u
tag=${1:-rXX}
out=gpurun_out
mkdir -p $out
export TMPDIR=/tmp
python bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
timeout 900 rocprofv3 --kernel-trace --stats -d $out/prof_$tag -o run -- python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extras --no-roofline \
    > $out/${tag}_bench_under_rocprof.json 2> $out/${tag}_prof.err
db=$(find $out/prof_$tag -name "*.db" | head -1)
python tools/rocpd_stats.py $db > $out/${tag}_kernel_stats_default_bench_graph.txt
python tools/rocpd_timeline.py $db > $out/${tag}_timeline_one_step.txt
rm -rf $out/prof_$tag
# HBM traffic per kernel: two separate PMC passes (counters only with --kernel-trace, never with other trace domains)
cmd="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-roofline --no-graph"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d $out/pmc_${tag}_$c -o run -- $cmd > /dev/null 2> $out/${tag}_pmc_$c.err
done
python tools/make_traffic_json.py $(find $out/pmc_${tag}_FETCH_SIZE -name "*.db" | head -1) $(find $out/pmc_${tag}_WRITE_SIZE -name "*.db" | head -1) \
    "$cmd" > $out/${tag}_traffic.json
rm -rf $out/pmc_${tag}_FETCH_SIZE $out/pmc_${tag}_WRITE_SIZE
tail -c 600 $out/${tag}_bench.json
