#!/bin/bash
# hd-128 attention backward (attn_bwd2.hip): parity, then timing against the generic kernels
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/b1
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "attn" > gpurun_out/b1/pytest_attn.log 2>&1; echo "pytest attn rc=$?" >> gpurun_out/b1/pytest_attn.log
tail -${T_TAIL:-15} gpurun_out/b1/pytest_attn.log
timeout 300 python tools/bench_attn.py --bwd-only > gpurun_out/b1/bench_new.jsonl 2>&1
LMOD_ATTN_BWD=1 timeout 300 python tools/bench_attn.py --bwd-only > gpurun_out/b1/bench_old.jsonl 2>&1
grep -h attn_bwd gpurun_out/b1/bench_new.jsonl gpurun_out/b1/bench_old.jsonl
