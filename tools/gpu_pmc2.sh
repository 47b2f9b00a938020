#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-pmcB}
mkdir -p $OUT
i=0
for set in "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_IFETCH SQ_IFETCH_LEVEL SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU2" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_SMEM"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/p$i -o a --output-format csv -- python $OLDPWD/tools/attn_one.py ${2:-} > $OUT/p$i.log 2>&1)
done
python tools/pmc_summary.py $OUT/p1 $OUT/p2 2>&1 | grep -i attn
