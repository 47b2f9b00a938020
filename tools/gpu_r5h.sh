#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=$PWD/gpurun_out/r5h; mkdir -p $OUT
ATTN_AB_ROUNDS=${ATTN_AB_ROUNDS:-1} timeout 900 python tools/attn_ab.py $ATTN_AB_ARGS > $OUT/attn_ab.jsonl 2> $OUT/attn_ab.err; echo "ab rc=$?"; cat $OUT/attn_ab.jsonl; tail -3 $OUT/attn_ab.err
