"""`python bench.py --gpus N` must launch itself (VERDICT r02 missing #1; the reference's launcher is one command,
shells/train/qwen/dense2sparse_distillation.sh:48).  Runs here on CPU: two ranks rendezvous over gloo, prove the world
with a collective, lay out the REAL config-2 gradient buffer / ZeRO-2 shard plan / optimizer state on `meta` tensors and
print the `exchange` object.  No kernel runs (the HIP path has no CPU fallback)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

TRAINABLE = 2035388416        # config 2: 12 dense + 12x4-expert FFNs, 12 routers, projector (dense2sparse_distillation.sh:27-42)


def _run(*extra, gpus=2):
    env = dict(os.environ, LMOD_DIST_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)                                       # plain start: no launcher environment
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--launch-check", *extra],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                           # rank 0 alone prints, one JSON line
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks_and_plans_config2_exchange():
    out = _run()
    ex = out["exchange"]
    assert out["launch_check"] == "ok" and out["n_gpus"] == 2
    assert ex["backend"] == "gloo" and ex["world_seen_by_backend"] == 2 and ex["zero2"] is True
    assert out["trainable_params"] == TRAINABLE and out["grad_buffer_elems"] >= TRAINABLE
    plan = ex["plan"]
    c = plan["collectives_per_step"]
    # ZeRO-2: every big span is reduce-scattered — in bf16 by default since round 5, the reference's bf16 engine's exchange (SURVEY
    # config 3: 4.07 GB per optimizer step) — and its bf16 weights all-gathered; routers / biases are all-reduced
    assert ex["grad_dtype"] == "bf16"
    assert plan["sharded_params"] + plan["replicated_params"] == TRAINABLE and plan["rank_local_params"] == 0
    assert c["reduce_scatter"]["bytes"] == 2 * plan["sharded_params"] and 4.06e9 < c["reduce_scatter"]["bytes"] < 4.08e9
    assert c["all_gather"]["bytes"] == 2 * plan["sharded_params"]
    assert c["all_reduce"]["bytes"] == 2 * plan["replicated_params"]
    assert c["reduce_scatter"]["calls"] == c["all_gather"]["calls"] == 12 * 2 + 12 * 2 + 2     # FFN pairs + projector
    # optimizer state: half of every sharded span + all replicated ones
    assert out["optimizer_state_elems_rank0"] == plan["sharded_params"] // 2 + plan["replicated_params"]


@pytest.mark.parametrize("extra,check", [
    (("--ep", "2", "--experts", "8"), "ep"),
    (("--stage", "dpo", "--no-zero2", "--grad-dtype", "bf16"), "dpo"),
])
def test_bench_self_launch_variants(extra, check):
    out = _run(*extra)
    plan = out["exchange"]["plan"]
    if check == "ep":
        # config 5 on 2 ranks: each rank holds 4 of the 8 experts; expert gradients have no peer to reduce with (the
        # expert-data-parallel group has one member), dense spans are still exchanged over the world
        assert out["exchange"]["ep_size"] == 2
        assert plan["rank_local_params"] == 12 * 4 * 3 * 2048 * 5504
        assert plan["sharded_params"] + plan["replicated_params"] + plan["rank_local_params"] == out["trainable_params"]
    else:
        assert out["stage"] == "dpo" and out["exchange"]["zero2"] is False and out["exchange"]["grad_dtype"] == "bf16"
        c = plan["collectives_per_step"]
        assert c["reduce_scatter"]["calls"] == 0 and c["all_reduce"]["bytes"] == 2 * TRAINABLE
        assert out["optimizer_state_elems_rank0"] == out["grad_buffer_elems"]


def test_bench_self_launch_world4_expert_parallel_pairs_times_expert_data_parallel_pairs():
    """World 4, ep_size 2, 4 experts: ranks {0,1} and {2,3} are expert-parallel pairs (2 experts per rank), ranks {0,2} and {1,3} hold
    the SAME experts — their gradients are reduced over those pairs (groups of 2), everything else over all four.  The plan of
    this launch equals what the same command issued on hardware (profiles/r03_ranks_share_one_gpu_gloo.jsonl, 4th line)."""
    out = _run("--ep", "2", "--experts", "4", gpus=4)
    ex = out["exchange"]
    plan = ex["plan"]
    c = plan["collectives_per_step"]
    assert out["n_gpus"] == 4 and ex["world_seen_by_backend"] == 4 and ex["ep_size"] == 2 and plan["rank_local_params"] == 0
    expert_local = 12 * 2 * 3 * 2048 * 5504                      # 12 MoE layers x 2 local experts x (gate + up + down)
    dense = out["trainable_params"] - expert_local - plan["replicated_params"]
    assert plan["sharded_params"] == dense + expert_local
    assert c["reduce_scatter"]["calls"] == c["all_gather"]["calls"] == 12 * 2 + 12 * 2 + 2
    # a reduce-scatter hands the whole local span to the collective: 2 bytes (bf16 exchange) per element of dense and of local expert spans alike
    assert c["reduce_scatter"]["bytes"] == 2 * plan["sharded_params"] == 2447376384
    # optimizer state of rank 0: 1/4 of the dense spans, 1/2 of its experts' spans (the expert-data-parallel group has 2 members)
    assert out["optimizer_state_elems_rank0"] == dense // 4 + expert_local // 2 + plan["replicated_params"] == 508923904


def test_bench_self_launch_world8_config3_and_config5():
    """The two 8-GPU configurations of BASELINE.json as the driver starts them (`python bench.py --gpus 8 [...]`): config 3 (data
    parallel, ZeRO-2) and config 5 (8 experts, ep_size 8: one expert of every MoE layer per rank, nothing to reduce them with)."""
    out = _run(gpus=8)
    plan = out["exchange"]["plan"]
    assert out["n_gpus"] == 8 and out["exchange"]["world_seen_by_backend"] == 8 and plan["rank_local_params"] == 0
    assert plan["sharded_params"] + plan["replicated_params"] == TRAINABLE
    assert out["optimizer_state_elems_rank0"] == plan["sharded_params"] // 8 + plan["replicated_params"]
    out = _run("--ep", "8", "--experts", "8", gpus=8)
    plan = out["exchange"]["plan"]
    one_expert_per_layer = 12 * 3 * 2048 * 5504
    assert out["exchange"]["ep_size"] == 8 and plan["rank_local_params"] == one_expert_per_layer
    assert plan["sharded_params"] + plan["replicated_params"] + plan["rank_local_params"] == out["trainable_params"]
    assert out["optimizer_state_elems_rank0"] == plan["sharded_params"] // 8 + plan["replicated_params"] + one_expert_per_layer
    assert plan["collectives_per_step"]["reduce_scatter"]["calls"] == 12 * 2 + 2        # dense FFN pairs + projector; experts stay local


def test_every_rank_runs_what_contains_collectives():
    """Round 4's bench ran its in-step aggregate (one more optimizer step) and the optimizer timing (sharded AdamW + all-gathers) inside
    `if rank == 0:` — at N > 1 rank 0 then waits for peers that never come.  Everything that launches work must sit in front of that block."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    gate = main.index("    if rank == 0:\n        sps = ")
    for call in ("in_step_gemm_aggregate(step", "time_optimizer(gb, opt", "time_dominant_kernel(dom_key"):
        assert 0 < main.index(call) < gate, call
    tail = main[gate:]
    for call in ("in_step_gemm_aggregate(", "time_optimizer(", "timed(step,"):
        assert call not in tail.split("want_extras")[0], call          # (the world-1-only extra legs come after `want_extras`)
