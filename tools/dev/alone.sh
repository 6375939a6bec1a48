#!/bin/bash
# alone.sh TAG [env...]: stand-alone (single-stream, eager) per-kernel stats of one step under rocprofv3 -> gpurun_out/TAG_alone.txt
tag=$1; shift
export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace -d gpurun_out/prof_$tag -o run -- python tools/dev/gpu_alone_stats.py > /dev/null 2>&1
db=$(find gpurun_out/prof_$tag -name "*.db" | head -1)
python tools/rocpd_stats.py $db | grep -v "at::native\|Cijk\|rocclr" > gpurun_out/${tag}_alone.txt
rm -rf gpurun_out/prof_$tag
head -32 gpurun_out/${tag}_alone.txt | cut -c1-60,80-140
