#!/bin/bash
# bench_env_ab.sh ROUNDS "ENV=.. ENV2=.." "ENV=.." ...: fresh bench.py processes, alternating between environment settings ("-" = none)
R=$1; shift
for i in $(seq $R); do
  for v in "$@"; do
    e=$v; [ "$v" = "-" ] && e=""
    ms=$(env $e python bench.py --no-cpu-baseline --no-extras --no-roofline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])")
    echo "[$v] $ms"
  done
done
