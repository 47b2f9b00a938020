#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=$PWD/gpurun_out/r5j; mkdir -p $OUT
export PYTHONFAULTHANDLER=1 LMOD_DIST_BACKEND=gloo
for c in 1 2; do
  t0=$(date +%s)
  LMOD_EP_CHUNKS=$c timeout -s ABRT 170 python bench.py --gpus 2 --micro-batch 1 --grad-accum 1 --experts 2 --ep 2 --steps 1 --warmup 1 --no-cpu-baseline --no-extras > $OUT/chunks$c.json 2> $OUT/chunks$c.err
  echo "chunks=$c rc=$? $(( $(date +%s) - t0 )) s"; grep -c '^{' $OUT/chunks$c.json
  grep -n "File \"/root/repo\|File \".*llavamod\|Thread\|Current thread" $OUT/chunks$c.err | head -40
done
