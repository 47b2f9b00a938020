"""The roofline record must be reproducible from `profiles/`: every committed
`*_gemm_traffic.json` states the algorithmic bytes of ITS shape (A + B + C once, bf16), bench.py computes the same
figure from the shape it launches, and `tools/profile_md.py` counts the optimizer steps of a trace from the trace."""
import glob
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_traffic_records_state_the_algorithmic_bytes_of_their_shape():
    import bench
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_gemm_traffic.json")))
    assert files
    for f in files:
        d = json.load(open(f))
        M, N, K = d["shape"]
        out_cols = d.get("output_cols", N)            # fused SwiGLU forward (round 6): the GEMM is N = 2I wide, the stored result I
        assert d["algorithmic_bytes"] == (M * K + N * K + M * out_cols) * 2, f
        if out_cols == N:
            assert d["algorithmic_bytes"] == bench.gemm_algorithmic_bytes(M, N, K), f
        assert d["fetch_bytes_corrected"] > 0 and d["write_bytes"] >= M * out_cols * 2 * 0.9, f


def test_profile_md_counts_steps_from_the_trace_and_shares_the_byte_formula():
    import bench
    pm = _load(os.path.join(ROOT, "tools", "profile_md.py"), "profile_md")
    assert pm.gemm_algorithmic_bytes(32768, 12288, 4096) == bench.gemm_algorithmic_bytes(32768, 12288, 4096) == 1174405120
    # the round-4 trace: 560 fused QKV + RoPE launches = (24 + 32) layers x 2 micro-batches x 5 optimizer steps
    assert pm.steps_in_trace(560, 2) == 5
    assert pm.steps_in_trace(0, 2) == 0


def test_one_workload_name_per_bench_line():
    import bench
    names = [bench.workload_name("mimic", 4, 1, 1), bench.workload_name("mimic", 4, 8, 1), bench.workload_name("dpo", 4, 1, 1),
             bench.workload_name("mimic", 8, 8, 8), bench.workload_name("mimic", 8, 1, 1)]
    for n, c in zip(names, ("config 2", "config 3", "config 4", "config 5", "config 5")):
        assert n.startswith(c + ":") and sum(n.count(f"config {i}") for i in range(1, 6)) == 1, n


def test_roofline_label_follows_the_committed_kernel_summary():
    """the `roofline` object's dominant kernel is the TOP ROW of the newest committed rocprofv3 summary of the
    bench workload, not a constant in bench.py."""
    import csv
    import re
    import bench
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_final_bench_kernel_stats.csv")))
    assert files
    rows = list(csv.DictReader(open(files[-1])))
    top = max(rows, key=lambda r: float(r["TotalDurationNs"]))
    dom, key, label = bench.roofline_recipe()
    assert dom["file"] == os.path.basename(files[-1]) and dom["name"] == top["Name"] and dom["pct"] == float(top["Percentage"])
    # the live recipe times the SAME instantiation: `gemm4_kernel<N, ...>` in the profile <-> `gemm4_kernel<N>` in the label
    inst = re.match(r"void (\w+)<(\d+)", top["Name"])
    assert inst and label.startswith(f"{inst.group(1)}<{inst.group(2)}>"), (top["Name"], label)
    assert key in ("swiglu", "nt")
    assert 50.0 < dom["family_pct"] <= 100.0


def test_in_step_rows_cover_the_gemm4_family_and_attention():
    """The in-step aggregate classifies every traced entry point (argument positions of include/lmod_hip.h)."""
    import bench
    nt = (1, 1, 1, 0, 32768, 12288, 4096, 4096, 4096, 12288, 1, 0, 0, 0, 0, 0, 0, 0, 0)
    fam, key, fl = bench._in_step_row("lmod_gemm_bf16_nt", nt)
    assert "<7>" in fam and key == "32768x12288x4096" and fl == 2.0 * 32768 * 12288 * 4096
    sw = (1, 1, 1, 0, 32768, 11008, 4096, 4096, 4096, 11008, 0, 1, 0, 0, 0, 0, 0)
    fam, key, fl = bench._in_step_row("lmod_gemm_swiglu_bf16", sw)
    assert "<1>" in fam and key == "32768x22016x4096" and fl == 2.0 * 32768 * 22016 * 4096
    grouped = sw[:11] + (4,) + sw[12:16] + (1,)
    assert bench._in_step_row("lmod_gemm_swiglu_bf16", grouped)[2] is None          # live rows are device-side: timed, no flop count
    af = (1, 1, 1, 1, 1, 0, 0, 16, 2048, 16, 16, 128, 2048, 2048, 2048, 2048, 1, 1)
    fam, key, fl = bench._in_step_row("lmod_attn_fwd", af)
    assert fam == "attn_fwd2_kernel" and "causal" in key and fl == 4.0 * 16 * 16 * 2048 * 2048 * 128 * 0.5
    assert set(bench._IN_STEP) >= {"lmod_gemm_qkv_rope_bf16", "lmod_gemm_bf16_nt_res", "lmod_gemm_bf16_tn", "lmod_attn_bwd_split"}
