#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=$PWD/gpurun_out/${1:-r6d}
mkdir -p $OUT
for a in "1 256 1 1 0" "1 256 1 1 1" "1 512 1 1 0"; do echo "== $a"; timeout 200 python tools/dbg_ds.py $a 2>&1 | grep -v amdgpu.ids | tail -60; done > $OUT/dbg_ds.txt 2>&1
cat $OUT/dbg_ds.txt | cut -c1-600
