#!/bin/bash
# SQ counter passes (counters only with --kernel-trace) over an arbitrary command: tools/pmc_cmd.sh TAG KERNEL-SUBSTRING NAME -- cmd...
#   -> gpurun_out/TAG_pmc_sq_NAME.txt (three passes: instruction mix, wait/LDS split, matrix-core busy + derived figures where available)
set -u
tag=$1; filt=$2; name=$3; shift 4
out=gpurun_out; mkdir -p $out; export TMPDIR=/tmp
p1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM"
p2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
p3="SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY SQ_BUSY_CU_CYCLES"
i=0
: > $out/${tag}_pmc_sq_${name}.txt
for p in "$p1" "$p2" "$p3"; do
  i=$((i+1))
  timeout 900 rocprofv3 --kernel-trace --pmc $p -d $out/pmccmd_${tag}_$i -o run -- "$@" > /dev/null 2> $out/${tag}_pmccmd_$i.err
  python tools/pmc_summary.py $(find $out/pmccmd_${tag}_$i -name "*.db" | head -1) "$filt" 3 >> $out/${tag}_pmc_sq_${name}.txt
  rm -rf $out/pmccmd_${tag}_$i
done
python tools/pmc_sq_derive.py $out/${tag}_pmc_sq_${name}.txt > /dev/null 2>&1
cat $out/${tag}_pmc_sq_${name}.txt
