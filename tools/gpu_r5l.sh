#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=$PWD/gpurun_out/r5l; mkdir -p $OUT
LMOD_HIP_LIB=$PWD/alt_libs/liblmod_g4tob.so timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -k "gemm_tn or wgrad" > $OUT/pytest.log 2>&1; echo "pytest(ob) rc=$?"; tail -3 $OUT/pytest.log | cut -c1-200
timeout 600 python tools/wgrad_ab.py g4tob > $OUT/wgrad_ab.jsonl 2> $OUT/wgrad_ab.err; echo "ab rc=$?"; cat $OUT/wgrad_ab.jsonl; tail -2 $OUT/wgrad_ab.err
