#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=$PWD/gpurun_out/r5c; mkdir -p $OUT
timeout 600 python -m pytest tests/test_full_width_gpu.py -m gpu -q -k config5 > $OUT/alone.log 2>&1; echo "alone rc=$?"; tail -5 $OUT/alone.log | cut -c1-300
timeout 900 python -m pytest tests/test_full_width_gpu.py -m gpu -q > $OUT/file.log 2>&1; echo "file rc=$?"; tail -8 $OUT/file.log | cut -c1-300
grep -n "slots_used\|AssertionError" $OUT/file.log | head -10 | cut -c1-400
