#!/bin/bash
# round 5, session A: the whole GPU suite (new knob / tolerance / grouped-persistent tests included), the routing A/B, then the
# default bench line with the config-4 / config-5 legs and the full-depth twin floors
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
OUT=$PWD/gpurun_out/r5a; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"
tail -15 $OUT/pytest.log
timeout 300 python tools/bench_r5_routing.py > $OUT/routing.jsonl 2> $OUT/routing.err; echo "routing rc=$?"; cat $OUT/routing.jsonl; tail -3 $OUT/routing.err
timeout 1200 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?"
tail -c 6000 $OUT/bench_default.json; tail -5 $OUT/bench_default.err
