#!/bin/bash
# timeline.sh TAG [env...]: one-step kernel timeline of graph replays (tools/dev/gpu_pipe_tl.py) -> gpurun_out/TAG_timeline.txt
tag=$1; shift
export TMPDIR=/tmp
env "$@" rocprofv3 --kernel-trace -d gpurun_out/prof_$tag -o run -- python tools/dev/gpu_pipe_tl.py > /dev/null 2>&1
db=$(find gpurun_out/prof_$tag -name "*.db" | head -1)
python tools/rocpd_timeline.py $db > gpurun_out/${tag}_timeline.txt
rm -rf gpurun_out/prof_$tag
cut -c1-110 gpurun_out/${tag}_timeline.txt
