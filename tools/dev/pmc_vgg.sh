set -u
out=gpurun_out; mkdir -p $out; export TMPDIR=/tmp
cmd="python tools/dev/gpu_vgg_profile.py 0"
p1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA"
p2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
i=0
: > $out/r06_f_pmc_sq_conv_sub8.txt
for p in "$p1" "$p2"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $p -d $out/pmcvgg_$i -o run -- $cmd > /dev/null 2> $out/pmcvgg_$i.err
  db=$(find $out/pmcvgg_$i -name "*.db" | head -1)
  for k in "conv3x3_kernel<0, 2, true>" "conv3x3_kernel<0, 2, false>"; do python tools/pmc_summary.py $db "$k" 4 >> $out/r06_f_pmc_sq_conv_sub8.txt; done
  rm -rf $out/pmcvgg_$i
done
cat $out/r06_f_pmc_sq_conv_sub8.txt
