#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=$PWD/gpurun_out/r5e; mkdir -p $OUT
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "attn" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log | cut -c1-300
timeout 600 python tools/attn_ab.py ${ATTN_AB_ARGS:-f2nodma} > $OUT/attn_ab.jsonl 2> $OUT/attn_ab.err; echo "ab rc=$?"; cat $OUT/attn_ab.jsonl; tail -3 $OUT/attn_ab.err
