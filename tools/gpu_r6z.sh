#!/bin/bash
# Round 6, last call: the driver's own sequence on the final binary — GPU suite, smoke, default bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=$PWD/gpurun_out/${1:-r6z}
mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest_gpu.txt | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -1
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_form.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
r=json.loads([l for l in open("$OUT/bench_driver_form.json") if l.startswith("{")][-1])
print(r["value"], r["ms_per_step"], r["roofline"]["kernel"][:50], r["roofline"]["achieved"], r["roofline"]["frac"], r["roofline"]["traffic"], r["grad_dtype"])
print(r["extra"]["config4_pairs_per_s"]["value"], r["extra"]["config5_samples_per_s"]["value"], r["cpu_baseline"]["value"], {k: v.get("rel") for k, v in r["cpu_baseline"]["loss_delta"].items() if isinstance(v, dict) and "rel" in v})
PY
