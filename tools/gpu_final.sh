#!/bin/bash
# End-of-round measurement set (everything profiles/rNN_final_* is built from):
#   default bench line; rocprofv3 kernel trace of the bench workload; PMC passes (separate, kernel-trace only) on the dominant
#   GEMM (SQ counters, FETCH_SIZE, WRITE_SIZE) and on the attention kernels; kernel micro-benchmarks.
# usage: gpu_final.sh <tag>      -> gpurun_out/<tag>/ ; then  python tools/profile_md.py gpurun_out/<tag> profiles/rNN_final
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${1:-final}
mkdir -p $OUT
timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/bench -o a --output-format csv -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-teacher-prefetch > $OUT/bench_trace.log 2>&1)
grep '^{' $OUT/bench_trace.log | tail -1 > $OUT/bench_line.json
SQ1="GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
# (round 6: tools/gemm_one.py launches the kernel the bench line's roofline describes — G1_KERNEL, default the fused SwiGLU forward — and says which in gemm_one_meta.json)
(cd /tmp && G1_META=$OUT/gemm_one_meta.json timeout 300 rocprofv3 --kernel-trace --pmc $SQ1 -d $OUT/gemm_sq -o a --output-format csv -- python $OLDPWD/tools/gemm_one.py > $OUT/gemm_sq.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/gemm_fetch -o a --output-format csv -- python $OLDPWD/tools/gemm_one.py > $OUT/gemm_fetch.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/gemm_write -o a --output-format csv -- python $OLDPWD/tools/gemm_one.py > $OUT/gemm_write.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $SQ1 -d $OUT/wgrad_sq -o a --output-format csv -- python $OLDPWD/tools/wgrad_one.py > $OUT/wgrad_sq.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/wgrad_fetch -o a --output-format csv -- python $OLDPWD/tools/wgrad_one.py > $OUT/wgrad_fetch.log 2>&1)
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/wgrad_write -o a --output-format csv -- python $OLDPWD/tools/wgrad_one.py > $OUT/wgrad_write.log 2>&1)
(cd /tmp && A1_B=16 timeout 300 rocprofv3 --kernel-trace --pmc $SQ1 -d $OUT/attn_sq -o a --output-format csv -- python $OLDPWD/tools/attn_one.py bwd > $OUT/attn_sq.log 2>&1)
(cd /tmp && A1_B=16 timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/attn_inst -o a --output-format csv -- python $OLDPWD/tools/attn_one.py bwd > $OUT/attn_inst.log 2>&1)
timeout 600 python tools/bench_kernels.py > $OUT/kernel_microbench.jsonl 2> $OUT/kernel_microbench.err
timeout 300 python tools/bench_gemm.py > $OUT/gemm_shapes.json 2> $OUT/gemm_shapes.err
timeout 300 python tools/bench_attn.py --bwd-only > $OUT/attn_bench.jsonl 2>/dev/null
timeout 300 python tools/bench_attn.py >> $OUT/attn_bench.jsonl 2>/dev/null
if [ -n "$FINAL_DPO" ]; then timeout 1500 python bench.py --stage dpo --micro-batch 8 > $OUT/bench_dpo.json 2> $OUT/bench_dpo.err; echo "dpo rc=$?"; fi
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_driver_form.json 2>/dev/null
timeout 300 python tools/bench_r5_routing.py > $OUT/routing.jsonl 2>/dev/null
for ds in 1 0; do LMOD_ATTN_DS=$ds timeout 300 python tools/bench_attn.py --bwd-only 2>/dev/null | grep attn_bwd >> $OUT/attn_bwd_forms.jsonl; done
timeout 1800 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.txt 2>&1; tail -3 $OUT/pytest_gpu.txt
(cd tools && timeout 300 python bench_attn.py --hd64 > $OUT/attn_hd64.jsonl 2>/dev/null; LMOD_ATTN_FWD=1 timeout 300 python bench_attn.py --hd64 > $OUT/attn_hd64_generic.jsonl 2>/dev/null)
ls $OUT; tail -c 600 $OUT/bench_default.json
