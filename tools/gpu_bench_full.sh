#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/bf
SECONDS=0; timeout 1500 python bench.py --steps 3 --warmup 1 > gpurun_out/bf/bench.json 2> gpurun_out/bf/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
r = json.loads(open("gpurun_out/bf/bench.json").read().strip().split("\n")[-1])
print(r["value"], r["ms_per_step"], json.dumps(r["cpu_baseline"], indent=1)[:2500])
PY
echo "wall ${SECONDS}s"; tail -3 gpurun_out/bf/bench.err
