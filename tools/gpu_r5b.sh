#!/bin/bash
# round 5, session B: re-run of the tests fixed after session A, config-5 routing probe, per-XCD footprint sweep (GROUP_M 2/4/8/16:
# TF + socket watts in one process, FETCH_SIZE per arm), the config-4 line with its full-depth twin floors
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5b; mkdir -p $OUT
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_knobs_gpu.py -m gpu -q --maxfail=20 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -12 $OUT/pytest.log
timeout 300 python tools/probe/cfg5_routing.py > $OUT/cfg5.log 2>&1; echo "cfg5 rc=$?"; tail -45 $OUT/cfg5.log
timeout 300 python tools/gemm_variants_ab.py --shapes qkv,tdown --only gm2,gm8,gm16 --seconds 2.0 --rounds 2 > $OUT/xcd_ab.md 2> $OUT/xcd_ab.err; echo "xcd rc=$?"; cat $OUT/xcd_ab.md
for v in current gm2 gm8 gm16; do
  lib=$PWD/alt_libs/liblmod_$v.so; [ $v = current ] && lib=$PWD/llava-mod_amd/llavamod/_lib/liblmod_hip.so
  (cd /tmp && LMOD_HIP_LIB=$lib timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch_$v -o a --output-format csv -- python $OLDPWD/tools/gemm_one.py > $OUT/fetch_$v.log 2>&1)
  python - <<PY
import csv, glob
rows=[r for f in glob.glob("$OUT/fetch_$v/*counter_collection.csv") for r in csv.DictReader(open(f)) if "gemm4_kernel" in r["Kernel_Name"] and r["Counter_Name"]=="FETCH_SIZE"]
v=[float(r["Counter_Value"]) for r in rows]
print("FETCH_SIZE $v", len(v), "dispatches, mean", sum(v)/max(1,len(v))/1e6*2, "GB (x2 corrected)")
PY
done
timeout 1500 python bench.py --stage dpo --micro-batch 8 > $OUT/bench_dpo.json 2> $OUT/bench_dpo.err; echo "dpo rc=$?"
python - <<PY
import json
l=[x for x in open("$OUT/bench_dpo.json") if x.startswith("{")]
d=json.loads(l[-1]); print(d["value"], d["unit"], d["ms_per_step"]); print(json.dumps(d.get("cpu_baseline"), indent=1)[:5000])
PY
tail -3 $OUT/bench_dpo.err
