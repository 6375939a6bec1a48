#!/bin/bash
# bench_ab.sh "HARP_ENG settings A" "settings B" [rounds]: fresh bench.py processes alternating between two engine settings (the burst regime
# the driver's bench measures: one process, 200 replays right after the warm-up)
A=$1; B=$2; R=${3:-3}
for i in $(seq $R); do
  for v in "$A" "$B"; do
    ms=$(HARP_ENG="$v" python bench.py --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "[$v] $ms"
  done
done
