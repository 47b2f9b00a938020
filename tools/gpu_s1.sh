#!/bin/bash
# GPU session 1: new hd-128 attention forward — parity, then A/B timing against the round-1 kernel and build variants
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/s1
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attn" > gpurun_out/s1/pytest_attn.log 2>&1; echo "pytest attn rc=$?" >> gpurun_out/s1/pytest_attn.log
tail -5 gpurun_out/s1/pytest_attn.log
LMOD_ATTN_FWD=1 timeout 300 python tools/bench_attn.py --bwd > gpurun_out/s1/bench_old.jsonl 2>&1
timeout 300 python tools/bench_attn.py > gpurun_out/s1/bench_new.jsonl 2>&1
for v in depth2 depth4 nosched thr0; do
  LMOD_HIP_LIB=$PWD/alt_libs/liblmod_$v.so timeout 300 python tools/bench_attn.py > gpurun_out/s1/bench_$v.jsonl 2>&1
done
grep -h '"S": 2048, "nh": 16' gpurun_out/s1/bench_*.jsonl | grep '"B": 8\|"B": 16' | head -40
