#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/t
timeout ${T_TIMEOUT:-600} python -m pytest ${T_ARGS:-tests/test_step_parity_gpu.py} -m gpu -x -q -s > gpurun_out/t/pytest.log 2>&1; echo "rc=$?"
tail -${T_TAIL:-30} gpurun_out/t/pytest.log
