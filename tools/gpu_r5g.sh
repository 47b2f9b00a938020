#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=$PWD/gpurun_out/r5g; mkdir -p $OUT
LMOD_ATTN_FWD=3 timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "attn" > $OUT/pytest.log 2>&1; echo "pytest(fwd3) rc=$?"; tail -4 $OUT/pytest.log | cut -c1-300
grep -n "^E  " $OUT/pytest.log | head -12 | cut -c1-260
ATTN_AB_ROUNDS=${ATTN_AB_ROUNDS:-2} timeout 900 python tools/attn_ab.py $ATTN_AB_ARGS > $OUT/attn_ab.jsonl 2> $OUT/attn_ab.err; echo "ab rc=$?"; cat $OUT/attn_ab.jsonl; tail -3 $OUT/attn_ab.err
