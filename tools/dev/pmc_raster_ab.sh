set -u
out=gpurun_out; mkdir -p $out; export TMPDIR=/tmp
cmd="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-roofline --no-graph"
p1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM"
p2="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
for lib in dense blocked; do
  if [ $lib = blocked ]; then export HARP_LIB_PATH=$PWD/harp_amd/csrc/variants/libharp_blocked.so; fi
  i=0
  for p in "$p1" "$p2"; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $p -d $out/pmcab_${lib}_$i -o run -- $cmd > /dev/null 2> $out/pmcab_${lib}_$i.err
    db=$(find $out/pmcab_${lib}_$i -name "*.db" | head -1)
    for k in "raster_kernel<1" "raster_kernel<0"; do python tools/pmc_summary.py $db "$k" 4 >> $out/pmcab_${lib}.txt; done
    rm -rf $out/pmcab_${lib}_$i
  done
done
