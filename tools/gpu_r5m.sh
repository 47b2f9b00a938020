#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/r5m
timeout 600 python -m pytest tests/test_step_parity_gpu.py -m gpu -q -s -k "per_token" > gpurun_out/r5m/pytest.log 2>&1; echo "rc=$?"; grep -E "labelled tokens|passed|failed|^E " gpurun_out/r5m/pytest.log | head
