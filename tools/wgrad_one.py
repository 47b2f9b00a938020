import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "llava-mod_amd"))
from llavamod import kernels as K
T, M, N = 32768, 6144, 2048
dy = torch.randn(T, M, device="cuda").to(torch.bfloat16); x = torch.randn(T, N, device="cuda").to(torch.bfloat16)
g = torch.zeros(M, N, device="cuda", dtype=torch.float32)
dyt, xt = K.transpose(dy), K.transpose(x)
for _ in range(3):
    K.gemm_wgrad(dyt, xt, g)
    K.gemm_wgrad(dyt, x, g, b_kmajor=True)
    K.gemm_tn(dy, x, out=g, accumulate=True)
torch.cuda.synchronize()
