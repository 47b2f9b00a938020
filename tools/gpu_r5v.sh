#!/bin/bash
# PMC passes (separate, kernel-trace only) on the hd-64 attention backward at the Qwen2-0.5B geometry and at 16 heads
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5v; mkdir -p $OUT
SQ1="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
SQ2="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
for cfg in "14 2" "16 16"; do set -- $cfg; tag=nh$1_kv$2
  (cd /tmp && A1_B=16 A1_NH=$1 A1_NKV=$2 A1_HD=64 timeout 200 rocprofv3 --kernel-trace --pmc $SQ1 -d $OUT/${tag}_sq -o a --output-format csv -- python $OLDPWD/tools/attn_one.py bwd > $OUT/${tag}_sq.log 2>&1)
  (cd /tmp && A1_B=16 A1_NH=$1 A1_NKV=$2 A1_HD=64 timeout 200 rocprofv3 --kernel-trace --pmc $SQ2 -d $OUT/${tag}_inst -o a --output-format csv -- python $OLDPWD/tools/attn_one.py bwd > $OUT/${tag}_inst.log 2>&1)
done
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/*_sq")) + sorted(glob.glob("$OUT/*_inst")):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", d.split("/")[-1])
    for k, v in acc.items():
        if "attn" in k: print("  ", k, {c: f"{sum(x)/len(x):.4g}" for c, x in sorted(v.items())})
PY
find $OUT -name "*.csv" ! -name "*counter_collection.csv" -delete; find $OUT -name "*kernel_trace*" -delete 2>/dev/null
du -sh $OUT
