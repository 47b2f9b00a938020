import os, sys, torch
sys.path.insert(0, "/root/repo/llava-mod_amd")
from llavamod import kernels as K
def t(fn, n=6):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (Kd, M, N) in [(8192, 8192, 8192), (4096, 16384, 4096), (16384, 11008, 2048 * 4)]:
    a = torch.randn(Kd, M, device="cuda").to(torch.bfloat16); b = torch.randn(Kd, N, device="cuda").to(torch.bfloat16)
    at, bt = a.t().contiguous(), b.t().contiguous()
    g = torch.zeros(M, N, device="cuda", dtype=torch.float32)
    fl = 2.0 * Kd * M * N
    r = {"nt": t(lambda: K.gemm_nt(at, bt, out=g, out_f32=True, accumulate=True)),
         "tn": t(lambda: K.gemm_tn(a, b, out=g, accumulate=True)),
         "x_kmajor": t(lambda: K.gemm_wgrad(at, b, g, b_kmajor=True))}
    print((Kd, M, N), {k: f"{v:.3f}ms {fl / v / 1e9:.0f}TF" for k, v in r.items()}, flush=True)
