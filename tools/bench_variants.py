"""Every GEMM variant the step launches, at step shapes (micro-batch 16): run under LMOD_GEMM_WAVES=4 / 8."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "llava-mod_amd"))
from llavamod import kernels as K
BF = torch.bfloat16
def t(fn, it=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3
R = {}
def rec(name, fl, fn):
    dt = t(fn); R[name] = (round(dt * 1e3, 3), round(fl / dt / 1e12))
T = 32768
x4 = torch.randn(T, 4096, device="cuda").to(BF); x2 = torch.randn(T, 2048, device="cuda").to(BF)
# teacher
w = (torch.randn(22016, 4096, device="cuda") * 0.02).to(BF)
rec("t swiglu fwd [32768x11008x4096]", 4.0 * T * 11008 * 4096, lambda: K.gemm_swiglu(x4, w))
act = torch.randn(T, 11008, device="cuda").to(BF); wd = torch.randn(4096, 11008, device="cuda").to(BF)
rec("t down [32768x4096x11008]", 2.0 * T * 4096 * 11008, lambda: K.gemm_nt(act, wd))
wq = torch.randn(12288, 4096, device="cuda").to(BF); bq = torch.randn(12288, device="cuda").to(BF)
rec("t qkv+bias [32768x12288x4096]", 2.0 * T * 12288 * 4096, lambda: K.gemm_nt(x4, wq, bias=bq))
# student dense
ws = (torch.randn(11008, 2048, device="cuda") * 0.02).to(BF)
rec("s swiglu fwd+gu [32768x5504x2048]", 4.0 * T * 5504 * 2048, lambda: K.gemm_swiglu(x2, ws, want_gu=True))
gu = torch.randn(T, 11008, device="cuda").to(BF); dy = torch.randn(T, 2048, device="cuda").to(BF)
wdt = torch.randn(5504, 2048, device="cuda").to(BF)
rec("s swiglu bwd [32768x5504x2048]", 2.0 * T * 5504 * 2048, lambda: K.gemm_swiglu_bwd(dy, wdt, gu, K=2048))
dgu = torch.randn(T, 11008, device="cuda").to(BF); wgt = torch.randn(2048, 11008, device="cuda").to(BF)
rec("s gu dgrad [32768x2048x11008]", 2.0 * T * 2048 * 11008, lambda: K.gemm_nt(dgu, wgt))
dgut, xt = K.transpose(dgu), K.transpose(x2)
g = torch.zeros(11008, 2048, device="cuda")
rec("s gu wgrad (split-K) [11008x2048x32768]", 2.0 * T * 11008 * 2048, lambda: K.gemm_wgrad(dgut, xt, g))
dqt = K.transpose(torch.randn(T, 6144, device="cuda").to(BF)); g2 = torch.zeros(6144, 2048, device="cuda")
rec("s qkv wgrad [6144x2048x32768]", 2.0 * T * 6144 * 2048, lambda: K.gemm_wgrad(dqt, xt, g2))
# MoE grouped
E, C, H, I = 4, 24576, 2048, 5504
mv = torch.tensor([16384] * 4, dtype=torch.int32, device="cuda")
xe = torch.randn(E, C, H, device="cuda").to(BF); we = (torch.randn(E, 2 * I, H, device="cuda") * 0.02).to(BF)
rec("moe swiglu fwd+gu grouped [4x16384x5504x2048]", 4.0 * 4 * 16384 * I * H, lambda: K.gemm_swiglu(xe, we, want_gu=True, m_valid=mv))
ae = torch.randn(E, C, I, device="cuda").to(BF); wde = torch.randn(E, H, I, device="cuda").to(BF)
rec("moe down grouped [4x16384x2048x5504]", 2.0 * 4 * 16384 * H * I, lambda: K.gemm_nt(ae, wde, m_valid=mv))
dyt = torch.randn(E, 2 * I, C, device="cuda").to(BF); xte = torch.randn(E, H, C, device="cuda").to(BF); ge = torch.zeros(E, 2 * I, H, device="cuda")
rec("moe gu wgrad grouped k_valid [4x11008x2048x16384]", 2.0 * 4 * 16384 * 2 * I * H, lambda: K.gemm_nt(dyt, xte, out=ge, out_f32=True, accumulate=True, k_valid=mv))
# heads
hr = torch.randn(8208, 2048, device="cuda").to(BF); wl = torch.randn(151936, 2048, device="cuda").to(BF)
rec("s lm_head [8208x151936x2048]", 2.0 * 8208 * 151936 * 2048, lambda: K.gemm_nt(hr, wl))
print("waves", os.environ.get("LMOD_GEMM_WAVES", "4"))
for k, v in R.items(): print(f"  {k:52s} {v[0]:8.3f} ms {v[1]:6d} TF")
