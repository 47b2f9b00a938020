#!/bin/bash
# last session of round 5: the whole GPU suite + smoke + a short bench line on the final binary
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=$PWD/gpurun_out/r5y; mkdir -p $OUT
timeout 600 python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest.log | tail -2
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_short.json 2> $OUT/bench_short.err; echo "bench rc=$?"
python - <<PY
import json
l=[x for x in open("$OUT/bench_short.json") if x.startswith("{")]
d=json.loads(l[-1]); print(d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"])
PY
