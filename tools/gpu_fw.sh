#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/fw
free -g | head -2; nproc
timeout 1200 python -m pytest tests/test_full_width_gpu.py tests/test_kernels_gpu.py tests/test_step_parity_gpu.py -x -q -s -k "config2 or deep_k or sumsq or clipped or two_gpu" > gpurun_out/fw/pytest.log 2>&1; echo "rc=$?"
tail -30 gpurun_out/fw/pytest.log
