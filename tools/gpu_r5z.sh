#!/bin/bash
# round 5, last session: exactly what the driver runs at round end, on the final commit
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=$PWD/gpurun_out/r5z; mkdir -p $OUT
timeout 900 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
t0=$(date +%s); timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2> $OUT/bench_driver.err; echo "bench rc=$? $(( $(date +%s) - t0 )) s"
python - <<PY
import json
l=[x for x in open("$OUT/bench_driver.json") if x.startswith("{")]
d=json.loads(l[-1]); print(d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic_note"][-120:])
print({k:(v.get("value"), v.get("ms_per_step")) for k,v in d.get("extra",{}).items()})
cb=d.get("cpu_baseline",{}); print(cb.get("value"), {k:(v.get("rel"),v.get("floor")) for k,v in cb.get("loss_delta",{}).items() if isinstance(v,dict) and "rel" in v})
PY
