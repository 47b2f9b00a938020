#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=$PWD/gpurun_out/r5k; mkdir -p $OUT
export LMOD_DIST_BACKEND=gloo
t0=$(date +%s)
timeout 240 python bench.py --gpus 2 --micro-batch 4 --experts 8 --ep 2 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/two_ranks_ep2.json 2> $OUT/two_ranks_ep2.err
echo "ep2 rc=$? $(( $(date +%s) - t0 )) s"; grep -c '^{' $OUT/two_ranks_ep2.json; grep '^{' $OUT/two_ranks_ep2.json | tail -c 900
t0=$(date +%s)
LMOD_EP_CHUNKS=2 timeout 240 python bench.py --gpus 2 --micro-batch 4 --experts 2 --ep 2 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/two_ranks_chunks2.json 2> $OUT/two_ranks_chunks2.err
echo "chunks rc=$? $(( $(date +%s) - t0 )) s"; grep -c '^{' $OUT/two_ranks_chunks2.json; grep '^{' $OUT/two_ranks_chunks2.json | tail -c 900
