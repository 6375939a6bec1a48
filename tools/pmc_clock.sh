#!/bin/bash
# effective clock and matrix-pipe occupancy of the kernels of a command: tools/pmc_clock.sh TAG KERNEL-SUBSTRING -- cmd...
#   one counter pass (GRBM_GUI_ACTIVE, SQ_BUSY_CYCLES, SQ_VALU_MFMA_BUSY_CYCLES, SQ_INSTS_MFMA) + the kernel trace's durations
set -u
tag=$1; filt=$2; shift 3
out=gpurun_out; mkdir -p $out; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $out/pmcclk_$tag -o run -- "$@" > /dev/null 2> $out/${tag}_pmcclk.err
db=$(find $out/pmcclk_$tag -name "*.db" | head -1)
python tools/pmc_summary.py $db "$filt" 3 > $out/${tag}_pmc_clock.txt
python - "$db" "$filt" >> $out/${tag}_pmc_clock.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table' or type='view'")]
t = [x for x in tabs if x.startswith("kernels") or x == "kernels"]
try:
    rows = cur.execute("select name, avg(end - start), count(*) from kernels where name like ? group by name", ("%" + sys.argv[2] + "%",)).fetchall()
    for n, d, c in rows:
        print(f"# kernel-trace: {n[:90]}  avg duration {d / 1e3:.1f} us over {c} launches")
except Exception as e:
    print("# (no kernels view:", e, tabs[:8], ")")
PY
rm -rf $out/pmcclk_$tag
cat $out/${tag}_pmc_clock.txt
