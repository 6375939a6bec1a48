#!/bin/bash
# timeline_ab.sh TAG ["HARP_ENG settings"]: one graph-replayed step's kernel timeline (rocprofv3 --kernel-trace over 40 replays of bench.py)
# under the given engine switches -> gpurun_out/TAG_timeline.txt; and the sustained step time without the profiler (3 x 60 replays)
set -u
tag=$1; export HARP_ENG=${2:-}
out=gpurun_out; mkdir -p $out; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $out/tl_$tag -o run -- python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extras --no-roofline > /dev/null 2> $out/tl_$tag.err
python tools/rocpd_timeline.py $(find $out/tl_$tag -name "*.db" | head -1) > $out/${tag}_timeline.txt
rm -rf $out/tl_$tag
echo "== $tag [$HARP_ENG]"; cut -c1-110 $out/${tag}_timeline.txt | tail -22
python tools/dev/gpu_step_ms.py 2>/dev/null | tail -1
