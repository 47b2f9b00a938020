#!/bin/bash
# attention backward variants: alt_libs/liblmod_<v>.so for v in "$@" (plus the in-tree library first)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
mkdir -p gpurun_out/b3
fmt='import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    if r["kernel"] == "attn_bwd": print("  ", r["B"], r["S"], "C" if r["causal"] else "F", r["ms"], r["tflops_algo(2.5x fwd)"])'
echo base; timeout 200 python tools/bench_attn.py --bwd-one 2>/dev/null | python -c "$fmt"
for v in "$@"; do
  echo $v; LMOD_HIP_LIB=$PWD/alt_libs/liblmod_$v.so timeout 200 python tools/bench_attn.py --bwd-one 2>/dev/null | python -c "$fmt"
done
