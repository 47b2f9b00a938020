#!/bin/bash
# Round 6, call J: regression legs on the final binary — smoke(), the ragged variants of SURVEY §8(d) (padded with key masks: dS-spill form with
# seqlens; unpadded: cu_seqlens, two-kernel form), and a 2-rank functional line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=$PWD/gpurun_out/${1:-r6j}
mkdir -p $OUT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
timeout 400 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --ragged 2>/dev/null | grep '^{' | tail -1 > $OUT/bench_ragged_padded.json
timeout 400 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --ragged --unpad 2>/dev/null | grep '^{' | tail -1 > $OUT/bench_ragged_unpad.json
LMOD_ATTN_DS=0 timeout 400 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extras --ragged 2>/dev/null | grep '^{' | tail -1 > $OUT/bench_ragged_padded_ds0.json
for f in ragged_padded ragged_unpad ragged_padded_ds0; do python -c "
import json
r=json.load(open('$OUT/bench_$f.json')); print('$f', r['value'], r['ms_per_step'], r['config']['final_loss'], r['config']['batch_shape'][:60])"; done
