#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5n; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attn_fwd_bwd" > $O/pytest64.log 2>&1; echo "rc=$?"; tail -5 $O/pytest64.log
LMOD_ATTN_BWD64=1 timeout 200 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attn_fwd_bwd" 2>&1 | tail -2
(cd tools && timeout 200 python bench_attn.py --hd64 > ../$O/hd64_bwd2.jsonl 2>/dev/null; LMOD_ATTN_BWD64=1 timeout 200 python bench_attn.py --hd64 > ../$O/hd64_generic.jsonl 2>/dev/null)
grep attn_bwd $O/hd64_bwd2.jsonl; echo ---; grep attn_bwd $O/hd64_generic.jsonl
