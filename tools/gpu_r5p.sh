#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5p; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_step_parity_gpu.py -m gpu -q > $O/pytest.log 2>&1; echo "rc=$?"; tail -6 $O/pytest.log
