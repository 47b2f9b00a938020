#!/bin/bash
# attention backward: per-kernel durations (kernel trace) + PMC passes on tools/attn_one.py bwd
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-b2}
mkdir -p $OUT
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt -o a --output-format csv -- python $OLDPWD/tools/attn_one.py bwd > $OUT/kt.log 2>&1)
python - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/kt/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "attn" in r["Name"]: print(r["Name"][:60], r["Calls"], "avg_us", round(float(r["AverageNs"])/1e3, 1))
PY
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o a --output-format csv -- python $OLDPWD/tools/attn_one.py bwd > $OUT/p$i.log 2>&1)
done
python tools/pmc_summary.py $OUT/p1 $OUT/p2 2>&1 | grep -i "attn_bwd2"
