import os, sys, torch
sys.path.insert(0, "llava-mod_amd")
from llavamod import kernels as K
BF = torch.bfloat16
def t(fn, it=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
T = 32768
gu = torch.randn(T, 11008, device="cuda").to(BF); dy = torch.randn(T, 2048, device="cuda").to(BF); wdt = torch.randn(5504, 2048, device="cuda").to(BF)
print("waves", os.environ.get("LMOD_GEMM_WAVES", "default"), "dense swiglu bwd ms", round(t(lambda: K.gemm_swiglu_bwd(dy, wdt, gu, K=2048)), 3))
E, C, H, I = 4, 24576, 2048, 5504
mv = torch.tensor([16384] * 4, dtype=torch.int32, device="cuda")
dye = torch.randn(E, C, H, device="cuda").to(BF); wte = torch.randn(E, I, H, device="cuda").to(BF); gue = torch.randn(E, C, 2 * I, device="cuda").to(BF)
print("waves", os.environ.get("LMOD_GEMM_WAVES", "default"), "moe swiglu bwd ms", round(t(lambda: K.gemm_swiglu_bwd(dye, wte, gue, m_valid=mv, K=H)), 3))
