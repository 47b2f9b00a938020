#!/bin/bash
# Round 6, call A: baseline of the round on this box + the micro-batch sweep VERDICT r05 next #5 asks for
# (B in {1, 2, 4, 8, 16} x accum {2, 8}, incl. the reference's own 1 x 8: dense2sparse_distillation.sh:70-72).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=$PWD/gpurun_out/${1:-r6a}
mkdir -p $OUT
timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_base.json 2> $OUT/bench_base.err; echo "base rc=$?"
for a in 2 8; do
  for b in ${SWEEP_B:-1 2 4 8 16}; do
    [ "$a" = 2 ] && [ "$b" = 16 ] && continue     # = the baseline line above
    steps=3; [ $((a * b)) -le 8 ] && steps=6
    timeout 400 python bench.py --steps $steps --warmup 1 --no-cpu-baseline --no-extras --micro-batch $b --grad-accum $a 2> $OUT/sweep_b${b}_a${a}.err | grep '^{' | tail -1 > $OUT/sweep_b${b}_a${a}.json
    python - <<EOF | tee -a $OUT/micro_batch_sweep.jsonl
import json
try:
    r = json.load(open("$OUT/sweep_b${b}_a${a}.json"))
    print(json.dumps({"micro_batch": $b, "grad_accum": $a, "samples_per_s": r["value"], "ms_per_step": r["ms_per_step"],
                      "whole_step_frac": r["roofline"]["whole_step"]["frac"], "peak_hbm_gb": r["config"].get("peak_hbm_gb")}))
except Exception as e:
    print(json.dumps({"micro_batch": $b, "grad_accum": $a, "error": repr(e)}))
EOF
  done
done
tail -c 400 $OUT/bench_base.json
