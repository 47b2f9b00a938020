#!/bin/bash
# Round 6, call G: load-carrying epilogues — residual batches in flight (G4_RES_DEPTH 1 / 4 / 8) and de-phased persistent starts
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=$PWD/gpurun_out/${1:-r6g}
mkdir -p $OUT
run() { env "$@" timeout 300 python tools/bench_epilogue_r6.py 2>/dev/null | tee -a $OUT/epilogue_ab.jsonl; }
for rep in 1 2; do
  run LMOD_HIP_LIB=$PWD/alt_libs/liblmod_res1.so
  run X=1
  run LMOD_HIP_LIB=$PWD/alt_libs/liblmod_res8.so
  run LMOD_GEMM_STAGGER=$((4*256+3))
  run LMOD_GEMM_STAGGER=$((4*256+2))
  run LMOD_GEMM_STAGGER=$((2*256+4))
  run LMOD_GEMM_STAGGER=$((8*256+1))
done
python - <<PY
import json, collections
rows=[json.loads(l) for l in open("$OUT/epilogue_ab.jsonl")]
agg=collections.defaultdict(list)
for r in rows: agg[(r["op"], r["lib"], r["stagger"])].append(r["tflops"])
ops=sorted({r["op"] for r in rows})
for op in ops:
    print(op)
    for (o,l,s),v in sorted(agg.items()):
        if o==op: print("    lib %-20s stagger %-5s TF %s" % (l, s, v))
PY
