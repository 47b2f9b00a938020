"""Persistent vs one-tile-per-workgroup form of the plain GEMM at few rounds of the CUs and short K (the student's o_proj / dgrad shapes)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
from llavamod import kernels as K
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best
for (M, N, Kd) in [(32768, 2048, 2048), (32768, 2048, 6144), (32768, 4096, 4096), (32768, 2048, 5504), (32768, 5504, 2048)]:
    a = torch.randn(M, Kd, device="cuda").to(torch.bfloat16); b = torch.randn(N, Kd, device="cuda").to(torch.bfloat16)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    res = {}
    for r in ("10", "2"):
        os.environ["LMOD_GEMM_PERSIST_ROUNDS"] = r
        ms = t(lambda: K.gemm_nt(a, b, out=out))
        res["persist" if r == "2" else "one-tile"] = f"{ms:.4f} ms {2.0 * M * N * Kd / ms / 1e9:.0f} TF"
    print((M, N, Kd), "rounds", (M // 256) * ((N + 255) // 256) / 256, res, flush=True)
