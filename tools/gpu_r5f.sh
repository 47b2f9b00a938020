#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=$PWD/gpurun_out/r5f; mkdir -p $OUT
timeout 900 python -m pytest tests/test_step_parity_gpu.py::test_single_rank_rccl_exchange_paths_are_exact tests/test_two_ranks_one_gpu.py tests/test_full_width_gpu.py::test_config5_full_depth_step_properties tests/test_moe_ep_gpu.py tests/test_comm_gpu.py -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest.log | cut -c1-400
grep -n "^E " $OUT/pytest.log | head -20 | cut -c1-300
timeout 600 python bench.py --no-cpu-baseline --no-extras > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - <<PY
import json
l=[x for x in open("$OUT/bench.json") if x.startswith("{")]
d=json.loads(l[-1]); print(d["value"], d["unit"], d["ms_per_step"], d["roofline"]["achieved"], d["roofline"]["in_step"]["achieved"], d["roofline"]["whole_step"]["frac_executed"])
PY
