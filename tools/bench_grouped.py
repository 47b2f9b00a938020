import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
from llavamod import kernels as K
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e-3
E, C, H, I = 4, 12288, 2048, 5504
x = torch.randn(E, C, H, device="cuda").to(torch.bfloat16)
w = torch.randn(E, 2 * I, H, device="cuda").to(torch.bfloat16)
o = torch.empty(E, C, 2 * I, device="cuda", dtype=torch.bfloat16)
for name, mv in (("balanced 8192", [8192] * 4), ("full 12288", [12288] * 4), ("skewed", [12288, 9000, 7000, 4480]), ("none", None)):
    m = torch.tensor(mv, dtype=torch.int32, device="cuda") if mv else None
    rows = sum(mv) if mv else E * C
    dt = t(lambda: K.gemm_nt(x, w, out=o, m_valid=m))
    print(f"grouped gu fwd  m_valid={name:14s} {dt*1e6:8.1f} us  {2.0*rows*2*I*H/dt/1e12:7.1f} TF")
xd = torch.randn(32768, H, device="cuda").to(torch.bfloat16); od = torch.empty(32768, 2 * I, device="cuda", dtype=torch.bfloat16)
dt = t(lambda: K.gemm_nt(xd, w[0], out=od)); print(f"dense 32768x11008x2048          {dt*1e6:8.1f} us  {2.0*32768*2*I*H/dt/1e12:7.1f} TF")
# wgrad form: [E, 2I, Cpad] x [E, H, Cpad] -> [E, 2I, H] fp32 accumulate, k_valid
dyt = torch.randn(E, 2 * I, C, device="cuda").to(torch.bfloat16); xt = torch.randn(E, H, C, device="cuda").to(torch.bfloat16)
g = torch.zeros(E, 2 * I, H, device="cuda")
kv = torch.tensor([8192] * 4, dtype=torch.int32, device="cuda")
dt = t(lambda: K.gemm_nt(dyt, xt, out=g, out_f32=True, accumulate=True, k_valid=kv)); print(f"grouped wgrad gu k_valid=8192   {dt*1e6:8.1f} us  {2.0*E*8192*2*I*H/dt/1e12:7.1f} TF")
gd = torch.zeros(2 * I, H, device="cuda"); dd = torch.randn(2 * I, 16384, device="cuda").to(torch.bfloat16); xx = torch.randn(H, 16384, device="cuda").to(torch.bfloat16)
dt = t(lambda: K.gemm_nt(dd, xx, out=gd, out_f32=True, accumulate=True)); print(f"dense wgrad gu K=16384          {dt*1e6:8.1f} us  {2.0*16384*2*I*H/dt/1e12:7.1f} TF")
dt = t(lambda: K.transpose(xd[:16384])); print(f"transpose 16384x2048 {dt*1e6:.1f} us")
