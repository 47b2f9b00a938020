#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/s2
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "attn" > gpurun_out/s2/pytest_attn.log 2>&1; echo "pytest attn rc=$?" >> gpurun_out/s2/pytest_attn.log
tail -3 gpurun_out/s2/pytest_attn.log
timeout 300 python tools/bench_attn.py > gpurun_out/s2/bench_new.jsonl 2>&1
cat gpurun_out/s2/bench_new.jsonl
bash tools/gpu_pmc_attn.sh s2/pmc
