#!/bin/bash
# full GPU suite + default bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/full
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/full/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/full/pytest_gpu.log
tail -15 gpurun_out/full/pytest_gpu.log
timeout 900 python bench.py ${BENCH_ARGS:---steps 4 --warmup 2 --no-cpu-baseline} > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err; echo "bench rc=$?"
cat gpurun_out/full/bench.json | cut -c1-1500; tail -5 gpurun_out/full/bench.err
