#!/bin/bash
# round 5, session D: attention timing ablations, the whole GPU suite, the config-4 line with per-token log-prob statistics
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=$PWD/gpurun_out/r5d; mkdir -p $OUT
timeout 600 python tools/attn_ablation.py > $OUT/attn_ablation.txt 2> $OUT/attn_ablation.err; echo "abl rc=$?"; cat $OUT/attn_ablation.txt; tail -3 $OUT/attn_ablation.err
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest.log | cut -c1-300
timeout 1500 python bench.py --stage dpo --micro-batch 8 > $OUT/bench_dpo.json 2> $OUT/bench_dpo.err; echo "dpo rc=$?"
python - <<PY
import json
l=[x for x in open("$OUT/bench_dpo.json") if x.startswith("{")]
d=json.loads(l[-1]); print(d["value"], d["unit"], d["ms_per_step"]); cb=d.get("cpu_baseline",{})
print(json.dumps({k:{kk:vv for kk,vv in v.items() if kk in ("rel","floor","floor_forced")} for k,v in cb.get("loss_delta",{}).items()}))
print(json.dumps(cb.get("token_logp_vs_fp32"), indent=1)[:6000])
PY
tail -3 $OUT/bench_dpo.err
