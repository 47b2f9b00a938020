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
        assert d["algorithmic_bytes"] == (M * K + N * K + M * N) * 2 == bench.gemm_algorithmic_bytes(M, N, K), f
        assert d["fetch_bytes_corrected"] > 0 and d["write_bytes"] >= M * N * 2 * 0.9, f


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
