#!/bin/bash
# round 5, session I: functional lines for the multi-GPU rows on the one GPU there is, and the ragged variants on the final binary
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=$PWD/gpurun_out/r5i; mkdir -p $OUT
RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 LMOD_FORCE_DIST=1 LMOD_DP_FORCE=1 LMOD_DP_NATIVE=1 timeout 600 python bench.py --no-cpu-baseline --no-extras > $OUT/world1_native.json 2> $OUT/world1_native.err; echo "native rc=$?"; tail -c 1500 $OUT/world1_native.json
LMOD_DIST_BACKEND=gloo LMOD_EP_CHUNKS=2 timeout 900 python bench.py --gpus 2 --micro-batch 4 --experts 2 --ep 2 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/two_ranks_ep_chunks.json 2> $OUT/two_ranks_ep_chunks.err; echo "ep chunks rc=$?"; tail -c 1200 $OUT/two_ranks_ep_chunks.json; tail -3 $OUT/two_ranks_ep_chunks.err
timeout 600 python bench.py --ragged --no-cpu-baseline --no-extras > $OUT/ragged_padded.json 2>/dev/null; echo "ragged rc=$?"
timeout 600 python bench.py --ragged --unpad --no-cpu-baseline --no-extras > $OUT/ragged_unpad.json 2>/dev/null; echo "unpad rc=$?"
python - <<PY
import json
for f in ("ragged_padded", "ragged_unpad"):
    l=[x for x in open("$OUT/%s.json" % f) if x.startswith("{")]
    d=json.loads(l[-1]); print(f, d["value"], d["ms_per_step"])
PY
