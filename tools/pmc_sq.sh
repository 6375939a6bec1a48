#!/bin/bash
# SQ counter passes (counters only with --kernel-trace) over an eager 3-step bench run: tools/pmc_sq.sh TAG [kernel-substring] [name]
#   -> gpurun_out/TAG_pmc_sq_NAME.txt (both passes + the derived figures of tools/pmc_sq_derive.py)
set -u
tag=${1:-rXX}; filt=${2:-shade_kernel}; name=${3:-kernel}
out=gpurun_out; mkdir -p $out; export TMPDIR=/tmp
cmd="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-roofline --no-graph"
p1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM"
p2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
i=0
for p in "$p1" "$p2"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $p -d $out/pmcsq_${tag}_$i -o run -- $cmd > /dev/null 2> $out/${tag}_pmcsq_$i.err
  python tools/pmc_summary.py $(find $out/pmcsq_${tag}_$i -name "*.db" | head -1) "$filt" 4 > $out/${tag}_pmc_sq_$i.txt
  rm -rf $out/pmcsq_${tag}_$i
done
cat $out/${tag}_pmc_sq_1.txt $out/${tag}_pmc_sq_2.txt > $out/${tag}_pmc_sq_${name}.txt
python tools/pmc_sq_derive.py $out/${tag}_pmc_sq_${name}.txt > /dev/null
cat $out/${tag}_pmc_sq_${name}.txt
