"""4-wave kernel (LMOD_GEMM_WAVES=4) fused SwiGLU forward timing for ablation builds (results are wrong by design)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
from llavamod import kernels as K
g = torch.Generator(device="cuda"); g.manual_seed(0)
rnd = lambda *s: (torch.rand(*s, device="cuda", generator=g) * 2 - 1).to(torch.bfloat16)
for tag, M, I, Kd in [("teacher", 32768, 11008, 4096), ("student", 32768, 5504, 2048)]:
    x, w = rnd(M, Kd), rnd(2 * I, Kd)
    for _ in range(3): K.gemm_swiglu(x, w)
    best = 1e9
    for r in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(20): K.gemm_swiglu(x, w)
        b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) / 20)
    print(os.path.basename(os.environ.get("LMOD_HIP_LIB", "current")), tag, round(best, 4), "ms", round(4.0 * M * I * Kd / best / 1e9, 1), "TF", flush=True)
