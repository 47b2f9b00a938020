"""wgrad forms at the student's shapes: NT (both operands transposed copies) vs X read reduction-major (MODE 3) vs TN."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
from llavamod import kernels as K
T = 32768
def t(fn, n=6):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for tag, M, N in [("gate+up", 11008, 2048), ("qkv", 6144, 2048), ("down", 2048, 5504), ("o", 2048, 2048)]:
    dy = torch.randn(T, M, device="cuda").to(torch.bfloat16); x = torch.randn(T, N, device="cuda").to(torch.bfloat16)
    g = torch.zeros(M, N, device="cuda", dtype=torch.float32)
    fl = 2.0 * T * M * N
    dyt, xt = K.transpose(dy), K.transpose(x)
    res = {"T(dy)": t(lambda: K.transpose(dy)), "T(x)": t(lambda: K.transpose(x)),
           "nt": t(lambda: K.gemm_wgrad(dyt, xt, g)),
           "x_kmajor": t(lambda: K.gemm_wgrad(dyt, x, g, b_kmajor=True)),
           "tn": t(lambda: K.gemm_tn(dy, x, out=g, accumulate=True))}
    print(tag, {k: f"{v:.3f}ms" + (f" {fl / v / 1e9:.0f}TF" if not k.startswith("T(") else "") for k, v in res.items()}, flush=True)
