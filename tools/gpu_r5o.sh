#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
O=gpurun_out/r5o; mkdir -p $O
timeout 300 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "attn" > $O/pytest_attn.log 2>&1; echo "rc=$?"; tail -5 $O/pytest_attn.log
(cd tools && timeout 200 python bench_attn.py --hd64 > ../$O/hd64_split.jsonl 2>/dev/null; LMOD_ATTN_BWD_SPLIT=0 timeout 200 python bench_attn.py --hd64 > ../$O/hd64_nosplit.jsonl 2>/dev/null; timeout 200 python bench_attn.py --bwd-only > ../$O/hd128_bwd.jsonl 2>/dev/null)
grep attn_bwd $O/hd64_split.jsonl; echo ---; grep attn_bwd $O/hd64_nosplit.jsonl; echo ---; grep attn_bwd $O/hd128_bwd.jsonl
