"""gate/up GEMM + SwiGLU kernel vs the fused-epilogue GEMM (teacher shape)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
from llavamod import kernels as K
M, I, Kd = 32768, 11008, 4096
x = torch.randn(M, Kd, device="cuda").to(torch.bfloat16)
w = (torch.randn(2 * I, Kd, device="cuda") * 0.02).to(torch.bfloat16)
w_il = w.view(2, I // 8, 8, Kd).transpose(0, 1).reshape(2 * I, Kd).contiguous()
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
gu = torch.empty(M, 2 * I, device="cuda", dtype=torch.bfloat16)
act = torch.empty(M, I, device="cuda", dtype=torch.bfloat16)
def two():
    K.gemm_nt(x, w, out=gu); K.swiglu_fwd(gu[:, :I], gu[:, I:])
print("gemm only ms", t(lambda: K.gemm_nt(x, w, out=gu)))
print("gemm + swiglu ms", t(two))
print("fused ms", t(lambda: K.gemm_nt(x, w_il, act=3, out=act)))
