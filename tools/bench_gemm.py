"""GEMM-only micro-benchmark over the step's dominant shapes (interleaved rounds, median)."""
import json, os, sys, statistics, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
from llavamod import kernels as K
SHAPES = [(16384, 22016, 4096, "t gate+up"), (16384, 4096, 11008, "t down"), (16384, 12288, 4096, "t qkv"),
          (16384, 4096, 4096, "t o"), (16384, 11008, 2048, "s gate+up"), (16384, 2048, 5504, "s down"),
          (16384, 6144, 2048, "s qkv"), (16384, 2048, 2048, "s o"), (8192, 8192, 8192, "sq8k"), (4096, 4096, 4096, "sq4k")]
bufs = []
for M, N, Kd, tag in SHAPES:
    bufs.append((torch.randn(M, Kd, device="cuda").to(torch.bfloat16), torch.randn(N, Kd, device="cuda").to(torch.bfloat16),
                 torch.empty(M, N, device="cuda", dtype=torch.bfloat16)))
res = {s[3]: [] for s in SHAPES}
for rnd in range(5):
    for (M, N, Kd, tag), (a, b, o) in zip(SHAPES, bufs):
        K.gemm_nt(a, b, out=o); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            K.gemm_nt(a, b, out=o)
        e1.record(); torch.cuda.synchronize()
        res[tag].append(2.0 * M * N * Kd / (e0.elapsed_time(e1) / 5 * 1e-3) / 1e12)
out = {t: round(statistics.median(v)) for t, v in res.items()}
tot_f = sum(2.0 * M * N * Kd for M, N, Kd, _ in SHAPES[:8])
tot_t = sum(2.0 * M * N * Kd / (out[t] * 1e12) for M, N, Kd, t in SHAPES[:8])
out["weighted8"] = round(tot_f / tot_t / 1e12)
print(os.environ.get("LMOD_HIP_LIB", "default").split("/")[-1], json.dumps(out))
