"""Dense weight-gradient GEMMs under a forced split-K factor (LMOD_WGRAD_SPLIT, read once per process)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
from llavamod import kernels as K
T = 32768
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
out = []
for tag, M, N in [("gate+up", 11008, 2048), ("down", 2048, 5504), ("proj1", 2048, 1024), ("qkv", 6144, 2048), ("o", 2048, 2048)]:
    dyt = torch.randn(M, T, device="cuda").to(torch.bfloat16); xt = torch.randn(N, T, device="cuda").to(torch.bfloat16)
    g = torch.zeros(M, N, device="cuda", dtype=torch.float32)
    ms = t(lambda: K.gemm_wgrad(dyt, xt, g))
    out.append(f"{tag} {ms:.3f}ms {2.0 * T * M * N / ms / 1e9:.0f}TF")
print("split", os.environ.get("LMOD_WGRAD_SPLIT", "auto"), "|", " | ".join(out), flush=True)
