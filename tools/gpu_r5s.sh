#!/bin/bash
# round 5, after the hd-64 backward work: what the driver runs at round end + a kernel trace of the hd-64 attention bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=$PWD/gpurun_out/r5s; mkdir -p $OUT
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
(cd /tmp && export TMPDIR=/tmp && cd $OLDPWD/tools && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_hd64 -- python bench_attn.py --hd64 > $OUT/hd64_under_profiler.jsonl 2>/dev/null)
f=$(ls $OUT/prof_hd64/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -12 "$f" > $OUT/hd64_kernel_stats.csv; rm -rf $OUT/prof_hd64
t0=$(date +%s); timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench rc=$? $(( $(date +%s) - t0 )) s"
python - <<PY
import json
l=[x for x in open("$OUT/bench_driver.json") if x.startswith("{")]
d=json.loads(l[-1]); print(d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"])
print({k:(v.get("value"), v.get("ms_per_step")) for k,v in d.get("extra",{}).items()})
PY
