#!/bin/bash
# bench_stage_ab.sh ROUNDS "ENV=.." ...: fresh bench.py processes (with extras): headline + the other stages' / batch sizes' ms per step
R=$1; shift
for i in $(seq $R); do
  for v in "$@"; do
    e=$v; [ "$v" = "-" ] && e=""
    env $e python bench.py --no-cpu-baseline --no-profile --vgg-weights none 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); ex=d['extras']
print('[$v] headline %.4f  geom %.4f  app %.4f  app_lean %.4f  B18 %.4f  C5 %.4f' % (d['ms_per_step'], ex['C3_stage_geometry_only']['ms_per_step'], ex['C3_stage_appearance_only']['ms_per_step'], ex['C3_stage_appearance_only_lean']['ms_per_step'], ex['C2_reference_batch_18']['ms_per_step'], ex['C5_arm_1024_per_gpu_share']['ms_per_step']))"
  done
done
