#!/bin/bash
# attention forward variants: alt_libs/liblmod_<v>.so for v in "$@" (plus the in-tree library first)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
fmt='import sys, json
for l in sys.stdin:
    try: r = json.loads(l)
    except Exception: continue
    if r["kernel"] == "attn_fwd": print("  ", r["B"], r["S"], r["nh"], "C" if r["causal"] else "F", r["ms"], r["tflops"])'
echo base; timeout 200 python tools/bench_attn.py 2>/dev/null | python -c "$fmt"
for v in "$@"; do
  echo $v; LMOD_HIP_LIB=$PWD/alt_libs/liblmod_$v.so timeout 200 python tools/bench_attn.py 2>/dev/null | python -c "$fmt"
done
