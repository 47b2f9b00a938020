#!/bin/bash
# Round 6, call H: (1) the reference's own launch regime (micro-batch 1 x accum 8) — which launches lose, and the 256-tile kernel forced
# on every GEMM as an A/B; (2) config-4 parity evidence over 4 pairs; (3) the forced-picks oracle arm of the mimic comparison.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=$PWD/gpurun_out/${1:-r6h}
mkdir -p $OUT
for t in default 256 default 256; do
  ev="X=1"; [ $t != default ] && ev="LMOD_GEMM_TILE=$t"
  env $ev timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras --micro-batch 1 --grad-accum 8 2> $OUT/b1_$t.err | grep '^{' | tail -1 > $OUT/b1_${t}_$RANDOM.json
done
for f in $OUT/b1_*.json; do python - <<PY
import json
r=json.load(open("$f")); print("$f".split("/")[-1], r["value"], r["ms_per_step"])
PY
done
python - <<PY
import json,glob
f=sorted(glob.glob("$OUT/b1_default_*.json"))[0]
r=json.load(open(f))
for k,v in r["roofline"]["in_step"]["families"].items():
    print(f"{k:66s} launches {v['launches']:4d} ms {v['ms']:8.2f} TF {v['achieved']}")
    for s,x in v["by_shape"].items(): print("      ", s, x)
PY
timeout 2400 python bench.py --stage dpo --micro-batch 8 --dpo-pairs 4 --steps 3 --warmup 1 > $OUT/bench_dpo.json 2> $OUT/bench_dpo.err; echo "dpo rc=$?"; tail -2 $OUT/bench_dpo.err
timeout 1500 python bench.py --steps 4 --warmup 2 --no-extras --cpu-forced-arm > $OUT/bench_forced_arm.json 2> $OUT/bench_forced_arm.err; echo "forced rc=$?"; tail -2 $OUT/bench_forced_arm.err
python - <<PY
import json
try:
    r=json.loads([l for l in open("$OUT/bench_dpo.json") if l.startswith("{")][-1])
    cb=r["cpu_baseline"]; print("dpo", r["value"], cb.get("sample","")[:120])
    for k,v in cb.get("token_logp_pooled",{}).items():
        if k=="note": print(v[:300]); continue
        print(k, {a: (x["n"], x["mean"], x["stderr"], x["mean_in_stderr"], x["within_3_stderr"]) for a,x in v.items()})
except Exception as e: print("dpo parse", repr(e))
try:
    r=json.loads([l for l in open("$OUT/bench_forced_arm.json") if l.startswith("{")][-1])
    ld=r["cpu_baseline"]["loss_delta"]
    for k in ("grad_rel_frobenius_free_running","grad_rel_frobenius_gpu_picks_forced_into_fp32_oracle","grad_rel_frobenius_twin_forced","grad_gpu_picks_forced_within_2x_twin_forced_floor","loss_gpu_picks_forced","forced_arm_note"):
        print(k, json.dumps(ld.get(k))[:700])
except Exception as e: print("forced parse", repr(e))
PY
